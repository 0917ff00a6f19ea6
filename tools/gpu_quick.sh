cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_sampler.py -x -q -k "two_mode_mixture_at_d40" 2>&1 | tail -15
