#!/bin/bash
# round 4: pairs fetched four ahead (PIPE = 4) in the kernels at three / four waves per SIMD too: dimension sweep old vs new, same box
cd "$(dirname "$0")/.."
O=gpurun_out/r4k; rm -rf $O; mkdir -p $O
DIMS="4 8 12 16 20 24 27 30 32 36 40 44 48 52"
for v in new pipe4; do
  MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so INC_ONLY=1 timeout 900 python tools/inc_bench.py $DIMS > $O/sweep_$v.log 2>&1
done
paste -d'|' $O/sweep_new.log $O/sweep_pipe4.log | cut -c1-240
# MODE 1 (bounds that differ) and MODE 2 (normal priors: the config-5 shape of bench.py)
for v in new pipe4; do
  MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 600 python tools/cliff_bench.py 30:1:-1 48:1:-1 52:1:-1 > $O/mode1_$v.log 2>&1
done
paste -d'|' $O/mode1_new.log $O/mode1_pipe4.log | cut -c1-240
