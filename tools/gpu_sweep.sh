cd $GRAFT_REPO_ROOT
for FL in "-DMCMC_CH=12" "-DMCMC_CH=20" "-DMCMC_CH=24" "-DMCMC_CH=16 -DMCMC_RNGPIPE=false"; do
  MCMC_HIP_DIMS=30 MCMC_HIP_EXTRA_FLAGS="$FL" python -m cobaya_amd.build > /dev/null 2>&1
  echo "$FL: $(python tools/quick_engine_bench.py 30 65536 64 300 2>&1 | tail -1)"
done
