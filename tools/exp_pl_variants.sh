#!/bin/bash
# Developer experiments on pl_chi2_kernel: builds cobaya_amd/csrc/_exp/lib_<name>.so with extra -D
# flags for pliklite_kernels.hip (e.g. -DMCMC_PL_PREFETCH=4).
#   tools/exp_pl_variants.sh pf4 "-DMCMC_PL_PREFETCH=4" pf2 "-DMCMC_PL_PREFETCH=2"
# Run on the GPU with MCMC_HIP_LIB=<that .so> python tools/pliklite_bench.py
set -e
cd "$(dirname "$0")/.."
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -mllvm -pragma-unroll-threshold=1000000"
OBJS=$(ls $CS/_obj/*.o | grep -v pliklite.o)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $FL $flags -c $CS/pliklite_kernels.hip -o $CS/_exp/pl_$name.o -save-temps=obj 2>/dev/null &&
    hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/pl_$name.o $OBJS -o $CS/_exp/lib_$name.so &&
    echo "built $name: $(grep -A12 'pl_chi2_kernelILi5' $CS/_exp/pl_$name-hip-amdgcn-amd-amdhsa-gfx950.s | grep -m2 'vgpr_count\|vgpr_spill' | tr -d '\n')" ) &
done
wait
rm -f $CS/_exp/*.bc $CS/_exp/*.hipi $CS/_exp/*.hipfb $CS/_exp/*.out $CS/_exp/*.cui 2>/dev/null || true
