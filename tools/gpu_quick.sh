cd $GRAFT_REPO_ROOT
MCMC_FUZZ_CASES=3000 timeout 280 python -m pytest tests/test_gpu_fuzz.py -q -x -k random_shapes 2>&1 | tail -6
