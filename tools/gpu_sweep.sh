cd $GRAFT_REPO_ROOT
for FL in "-DMCMC_CH=16" "-DMCMC_CH=16 -DMCMC_FAKE_HALF"; do
  MCMC_HIP_DIMS=30 MCMC_HIP_EXTRA_FLAGS="$FL" python -m cobaya_amd.build > /dev/null 2>&1
  echo "$FL: $(python tools/quick_engine_bench.py 30 65536 256 300 2>&1 | tail -1)"
done
