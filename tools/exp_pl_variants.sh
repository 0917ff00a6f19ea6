#!/bin/bash
# Developer experiments on pl_fused_kernel / pl_chi2_kernel: builds cobaya_amd/csrc/_exp/lib_<name>.so
# with extra -D flags for pliklite_kernels.hip (e.g. -DPL_DEEP_FROM=9).
#   tools/exp_pl_variants.sh nodeep "-DPL_DEEP_FROM=9" clk "-DPL_DEBUG_CLOCKS"
# Run on the GPU with MCMC_HIP_LIB=<that .so> python bench.py --workload pliklite ...
set -e
cd "$(dirname "$0")/.."
CS=cobaya_amd/csrc; mkdir -p $CS/_exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -mllvm -pragma-unroll-threshold=1000000"
OBJS=$(ls $CS/_obj/*.o | grep -v pliklite.o)
build_one() {
  name=$1; flags=$2
  hipcc $FL $flags -c $CS/pliklite_kernels.hip -o $CS/_exp/pl_$name.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 $CS/_exp/pl_$name.o $OBJS -ldl -o $CS/_exp/lib_$name.so
  hipcc $FL $flags -S --cuda-device-only -o /tmp/pl_$name.s $CS/pliklite_kernels.hip 2>/dev/null
  echo "built $name: $(grep -E '\.set .*pl_fused_kernelILi5ELi4.*(num_vgpr|private_seg_size)' /tmp/pl_$name.s | sed 's/.*\.\(num_vgpr\|private_seg_size\), /\1 /' | tr '\n' ' ')"
}
while [ $# -gt 0 ]; do
  build_one "$1" "$2" &
  shift 2
done
wait
