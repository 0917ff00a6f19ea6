#!/bin/bash
# round 4: dragging rows, d > 32 blocks / dragging from scratch, then the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dragging or blocked or blocking" 2>&1 | tail -15 > gpurun_out/r4c/drag.log
cat gpurun_out/r4c/drag.log
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4c/gpu_tests.log
cat gpurun_out/r4c/gpu_tests.log
