cd $GRAFT_REPO_ROOT
for v in prev cur prev cur; do
echo -n "$v: "; MCMC_HIP_LIB=cobaya_amd/csrc/_exp/lib_$v.so timeout 300 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
done
