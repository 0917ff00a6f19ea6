cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -12
timeout 300 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
