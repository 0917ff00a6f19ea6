cd $GRAFT_REPO_ROOT
for d in 33 40 48 64 80 100 112; do
timeout 120 python tools/quick_engine_bench.py $d 65536 128 $((2*d)) 2>&1 | tail -1
done
