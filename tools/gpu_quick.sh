cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "big_dimension or two_wave" 2>&1 | tail -2
MCMC_FUZZ_CASES=5 MCMC_FUZZ_BIG_CASES=200 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1 | cut -c1-120; }
for d in 53 55 56; do run $d 65536 256 $((d*10)); done
for d in 50 52 54 56; do echo "norm d=$d pair/mfma"; QB_NORM=$((d/2)) run $d 65536 256 $((d*10)); QB_NORM=$((d/2)) MCMC_HIP_NO_PAIR_BIG=1 run $d 65536 256 $((d*10)); done
