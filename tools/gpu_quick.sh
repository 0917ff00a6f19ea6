cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
MCMC_FUZZ_CASES=300 MCMC_FUZZ_BIG_CASES=60 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
for v in "" 1; do
  if [ -n "$v" ]; then export MCMC_HIP_NO_PREFETCH=1; else unset MCMC_HIP_NO_PREFETCH; fi
  echo "NO_PREFETCH=$v"
  timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['basis_kernel_ms_per_launch'])"
  timeout 200 python bench.py --dim 100 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['basis_kernel_ms_per_launch'])"
done
