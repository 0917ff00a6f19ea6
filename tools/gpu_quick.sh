cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_sampler.py -x -q -k config2 --durations=3 2>&1 | tail -25
