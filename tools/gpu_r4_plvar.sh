#!/bin/bash
# round 4: variants of pl_fused_kernel (tools/exp_pl_variants.sh): kernel time each (debug variants compute garbage)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4plvar
for lib in cobaya_amd/csrc/_exp/lib_*.so; do
  name=$(basename $lib .so)
  export MCMC_HIP_LIB=$PWD/$lib
  timeout 300 python tools/pliklite_bench.py 26 65536 24 > gpurun_out/r4plvar/$name.log 2>&1
  echo "$name | $(tail -2 gpurun_out/r4plvar/$name.log | tr '\n' ' ')"
done
