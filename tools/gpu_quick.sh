cd $GRAFT_REPO_ROOT
MCMC_FUZZ_CASES=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1; }
for d in 32 33 34 35 36 37 38 39 40 41 42 43 44 45 46 47 48 49; do run $d 65536 256 $((d*25)); done > gpurun_out/sweep_33_48.log
echo "-- matrix-core kernel at the same d (MCMC_HIP_NO_PAIR_BIG=1)" >> gpurun_out/sweep_33_48.log
for d in 33 40 48; do MCMC_HIP_NO_PAIR_BIG=1 run $d 65536 256 $((d*25)); done >> gpurun_out/sweep_33_48.log
