#!/bin/bash
# Developer experiments on the incremental step kernel of dq 1..8: builds
# cobaya_amd/csrc/_exp/lib_<name>.so with extra -D flags (MCMC_INC_WAVES_OVERRIDE,
# MCMC_INC_PIPE_OVERRIDE).   tools/exp_inc_variants.sh w3p4 "-DMCMC_INC_WAVES_OVERRIDE=3 -DMCMC_INC_PIPE_OVERRIDE=4" ...
# Run on the GPU with MCMC_HIP_LIB=<that .so> python bench.py --no-cpu-baseline --no-variants
set -e
cd "$(dirname "$0")/.."
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -pragma-unroll-threshold=1000000"
LO=${DQ_LO:-1}; HI=${DQ_HI:-8}     # the translation unit to rebuild: incremental_<LO>.o
OBJS=$(ls $CS/_obj/*.o | grep -v incremental_$LO.o)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $FL $flags -DMCMC_DQ_LO=$LO -DMCMC_DQ_HI=$HI -c $CS/incremental_kernels.hip -o $CS/_exp/inc_$name.o 2>/dev/null &&
    hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/inc_$name.o $OBJS -o $CS/_exp/lib_$name.so &&
    echo "built $name" ) &
done
wait
