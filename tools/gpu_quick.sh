cd $GRAFT_REPO_ROOT
for K in 1 2 3; do
QB_MODES=$K timeout 120 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1 | sed 's/; basis.*//'
done
