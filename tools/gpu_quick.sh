cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/quick_engine_bench.py 100 65536 128 200 2>&1 | tail -1
