cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -k big 2>&1 | tail -3
