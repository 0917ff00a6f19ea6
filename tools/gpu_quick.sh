cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_sampler.py -x -q -k manual_blocking --durations=2 2>&1 | tail -8
