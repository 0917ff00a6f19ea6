cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k big 2>&1 | tail -3
for v in rng4 rng1 rng4 rng1; do
echo -n "$v: "; MCMC_HIP_LIB=cobaya_amd/csrc/_exp/libbig_$v.so timeout 300 python tools/quick_engine_bench.py 100 65536 128 200 2>&1 | tail -1
done
