cd $GRAFT_REPO_ROOT
for d in 2 4 7 8 12 16 20 24 27 30 32; do
timeout 120 python tools/quick_engine_bench.py $d 65536 256 $((40*d)) 2>&1 | tail -1
done
for d in 33 48 64 80 100 112; do
timeout 120 python tools/quick_engine_bench.py $d 65536 128 $((2*d)) 2>&1 | tail -1
done
