#!/bin/bash
# round 4: d = 128 / 124 (dq 32 / 31) at two waves per SIMD again, now that the trial forms no residual: same-box A/B
cd "$(dirname "$0")/.."
for d in 128 124; do
  for v in head w2k0 w2k1; do
    MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so INC_ONLY=1 timeout 300 python tools/inc_bench.py $d 2>&1 | grep -o "step kernel [0-9.]* ms per [0-9]* steps" | sed "s/^/d=$d $v: /"
  done
done
