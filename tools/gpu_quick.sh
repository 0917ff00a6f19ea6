cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug_drag.py 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "drag or block" 2>&1 | tail -5
