cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "two_wave or big_dimension" 2>&1 | tail -2
MCMC_FUZZ_CASES=100 timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1; }
for d in 35 37 38 39 41 43 45 47; do run $d 65536 256 $((d*25)); done
