cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "big_dimension or two_wave" 2>&1 | tail -8
MCMC_FUZZ_CASES=5 MCMC_FUZZ_BIG_CASES=200 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8
