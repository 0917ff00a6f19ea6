#!/bin/bash
# Developer experiments: builds cobaya_amd/csrc/_exp/lib_<name>.so for ONE dimension with extra
# -D flags (timing-only variants of the step kernels; results of such variants are wrong).
#   tools/exp_variants.sh 30 base "" nobar "-DMCMC_EXP=1" ...
# Run on the GPU with MCMC_HIP_LIB=<that .so> python tools/quick_engine_bench.py 30 65536 256 1200
set -e
cd "$(dirname "$0")/.."
D=$1; shift
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -pragma-unroll-threshold=1000000"
[ -f $CS/_obj/capi.o ] || python -m cobaya_amd.build
BIG=""; [ "$D" -gt 32 ] && BIG=$CS/_obj/walker_big48.o; [ "$D" -gt 48 ] && BIG=$CS/_obj/walker_big56.o   # 32 < d <= 56: the other kernels
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $FL $flags -DMCMC_D=$D -c $CS/walker_kernels.hip -o $CS/_exp/w_$name.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/w_$name.o $BIG $CS/_obj/capi.o $CS/_obj/blocked.o $CS/_obj/general.o -o $CS/_exp/lib_$name.so &&
    echo "built $name" ) &
done
wait
