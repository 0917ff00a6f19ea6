#!/bin/bash
# Developer experiments on the incremental step kernel of dq 1..8: builds
# cobaya_amd/csrc/_exp/lib_<name>.so with the experiment hooks of cobaya_amd/csrc/_exp/inc_experiment.h
# force-included (-DEXP_STEP_WAVES, -DEXP_PIPE, -DEXP_BLOCK_TIMES ...: see that header).
#   tools/exp_inc_variants.sh w3p4 "-DEXP_STEP_WAVES=3 -DEXP_PIPE=4" ...
# Run on the GPU with MCMC_HIP_LIB=<that .so> python bench.py --no-cpu-baseline --no-variants
set -e
cd "$(dirname "$0")/.."
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -mllvm -pragma-unroll-threshold=1000000 -include $CS/_exp/inc_experiment.h"
LO=${DQ_LO:-1}; HI=${DQ_HI:-8}     # the translation unit to rebuild: incremental_<LO>.o
OBJS=$(ls $CS/_obj/*.o | grep -v incremental_$LO.o)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $FL $flags -DMCMC_DQ_LO=$LO -DMCMC_DQ_HI=$HI -c $CS/incremental_kernels.hip -o $CS/_exp/inc_$name.o 2>/dev/null &&
    hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/inc_$name.o $OBJS -o $CS/_exp/lib_$name.so &&
    echo "built $name" ) &
done
wait
