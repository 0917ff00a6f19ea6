cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "big" 2>&1 | tail -4
timeout 120 python tools/quick_engine_bench.py 128 65536 128 256 2>&1 | tail -1
