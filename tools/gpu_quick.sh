cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
MCMC_FUZZ_CASES=150 timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1; }
for d in 32 33 34 36 38 40 42 44 46 48 49; do run $d 65536 256 $((d*25)); done
