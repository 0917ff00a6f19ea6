#!/bin/bash
# round 4: bounds kernels + communicator + the sampler tests that touch them
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_gpu_bounds.py tests/test_gpu_multirank.py tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r4b/tests.log
cat gpurun_out/r4b/tests.log
