cd $GRAFT_REPO_ROOT
for W in 4096 16384 32768 65536 131072 262144; do
timeout 120 python tools/quick_engine_bench.py 30 $W 256 1200 2>&1 | tail -1
done
for W in 16384 65536 131072; do
timeout 120 python tools/quick_engine_bench.py 100 $W 128 200 2>&1 | tail -1
done
