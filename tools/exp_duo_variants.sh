#!/bin/bash
# Developer experiments on incremental_duo.hip (two lanes per walker): builds
# cobaya_amd/csrc/_exp/lib_<name>.so with _exp/inc_experiment.h force-included.
#   tools/exp_duo_variants.sh dep0 "-DEXP_DUO_DEPK=0" keepv "-DEXP_DUO_KEEPV=1" ...
# Run on the GPU with MCMC_HIP_LIB=<that .so> python tools/mix_bench.py 30:2
set -e
cd "$(dirname "$0")/.."
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -mllvm -pragma-unroll-threshold=1000000 -include $CS/_exp/inc_experiment.h"
LO=${DQ_LO:-1}; HI=${DQ_HI:-8}
OBJS=$(ls $CS/_obj/*.o | grep -v incremental_duo_$LO.o)
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $FL $flags -DMCMC_DUO_DQ_LO=$LO -DMCMC_DUO_DQ_HI=$HI -c $CS/incremental_duo.hip -o $CS/_exp/duo_$name.o 2>/dev/null &&
    hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/duo_$name.o $OBJS -ldl -o $CS/_exp/lib_$name.so &&
    echo "built $name" ) &
done
wait
