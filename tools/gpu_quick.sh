cd $GRAFT_REPO_ROOT
for v in base prio1 prio3 base prio1 prio3; do
echo -n "$v: "; MCMC_HIP_LIB=cobaya_amd/csrc/_exp/lib_$v.so timeout 120 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
done
