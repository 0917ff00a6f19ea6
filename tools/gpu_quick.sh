cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q -k "big" 2>&1 | tail -4
