cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
timeout 120 python tools/quick_engine_bench.py 12 65536 256 480 2>&1 | tail -1 | sed 's/; basis.*//'
