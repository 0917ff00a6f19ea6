cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6
QB_NORM=21 timeout 300 python tools/quick_engine_bench.py 27 65536 256 1080 2>&1 | tail -1
timeout 300 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
