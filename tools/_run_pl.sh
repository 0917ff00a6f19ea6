cd /root/repo
export MCMC_HIP_LIB_COMPAT=1
for v in $TESTV; do
  echo "== tests $v"; MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 600 python -m pytest tests/test_gpu_pliklite.py -m gpu -q -x 2>&1 | tail -2
done
for rep in 1 2; do
for v in $TIMEV; do
  L=$PWD/cobaya_amd/csrc/_exp/lib_$v.so; [ $v = cur ] && L=$PWD/cobaya_amd/csrc/libmcmc_hip.so
  echo "== $v $(MCMC_HIP_LIB=$L timeout 300 python tools/pliklite_bench.py 26 65536 24 2>&1 | grep 'per step')"
done; done
for v in $CLKV; do
  echo "== clocks $v"; MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 200 python tools/pl_clocks.py 2>&1 | tail -42
done
