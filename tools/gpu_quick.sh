cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "block or drag or random_shapes" 2>&1 | tail -3
QB_BLOCKS=10,3 timeout 120 python tools/quick_engine_bench.py 30 65536 256 1400 2>&1 | tail -1
QB_NORM=21 QB_BLOCKS=6,2 timeout 120 python tools/quick_engine_bench.py 27 65536 256 540 2>&1 | tail -1
