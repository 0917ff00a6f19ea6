cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in base s20 s24 base s20; do
echo -n "$v: "; MCMC_HIP_LIB=cobaya_amd/csrc/_exp/lib_$v.so timeout 300 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
done
