#!/bin/bash
# round 3: the periodic incremental kernel -- parity, then timing (shipped and experiment builds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3_per
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "periodic" 2>&1 | tail -15 > gpurun_out/r3_per/tests.log
cat gpurun_out/r3_per/tests.log
for lib in "" $(ls cobaya_amd/csrc/_exp/lib_per*.so 2>/dev/null); do
  echo "== ${lib:-shipped}"
  MCMC_HIP_LIB=$lib timeout 600 python tools/cliff_bench.py "$@" 2>&1
done > gpurun_out/r3_per/bench.log
cat gpurun_out/r3_per/bench.log
