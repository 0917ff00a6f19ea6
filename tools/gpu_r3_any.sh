#!/bin/bash
# round 3: the general incremental kernel -- parity cases, then the configuration sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3_any
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "general_kernel or refuses or periodic_steps or mixture_steps" 2>&1 | tail -25 > gpurun_out/r3_any/tests.log
cat gpurun_out/r3_any/tests.log
timeout 900 python tools/cliff_bench.py "$@" > gpurun_out/r3_any/cliff.log 2>&1
cat gpurun_out/r3_any/cliff.log
