cd $GRAFT_REPO_ROOT
MCMC_FUZZ_CASES=400 timeout 500 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -12
