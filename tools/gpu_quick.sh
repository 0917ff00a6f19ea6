cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "big_dimension or two_wave" 2>&1 | tail -2
MCMC_FUZZ_CASES=5 MCMC_FUZZ_BIG_CASES=100 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1 | cut -c1-130; }
for d in 80 100 112 120 128; do run $d 65536 256 $((d*8)); done
QB_NORM=50 run 100 65536 256 800
