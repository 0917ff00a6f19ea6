cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "big_dimension or two_wave or moments" 2>&1 | tail -2
MCMC_FUZZ_CASES=5 MCMC_FUZZ_BIG_CASES=150 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1; }
for d in 88 100 112 128; do run $d 65536 256 $((d*8)); done
timeout 300 python bench.py --dim 100 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
