cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "drag" 2>&1 | tail -3
QB_NORM=21 QB_BLOCKS=6,2,7 timeout 120 python tools/quick_engine_bench.py 27 65536 256 60 2>&1 | tail -1
