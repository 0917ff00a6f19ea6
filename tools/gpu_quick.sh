cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "big" 2>&1 | tail -6
timeout 120 python tools/quick_engine_bench.py 100 65536 128 200 2>&1 | tail -1 | sed 's/; basis.*//'
QB_NORM=70 timeout 120 python tools/quick_engine_bench.py 100 65536 128 200 2>&1 | tail -1 | sed 's/; basis.*//'
