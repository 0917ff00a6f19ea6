#!/bin/bash
# round 4: after StagedVariates in every incremental kernel -- the whole GPU suite, then the model sweep old vs new
cd "$(dirname "$0")/.."
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_tests.log
cat $O/gpu_tests.log
CASES="30:1:0 30:1:1 30:1:8 30:2:0 30:4:0 30:5:0 30:8:0 30:16:0 30:2:1 64:4:0 100:1:0 100:1:1 100:1:8 100:2:0 100:4:0 128:1:0"
for v in old new; do
  MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 900 python tools/cliff_bench.py $CASES > $O/sweep_$v.log 2>&1
done
paste -d'|' $O/sweep_old.log $O/sweep_new.log | cut -c1-260
