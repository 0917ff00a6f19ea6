#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4fuzz
MCMC_FUZZ_GENERAL_CASES=150 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "above_d32" 2>&1 | tail -25 > gpurun_out/r4fuzz/fuzz.log
cat gpurun_out/r4fuzz/fuzz.log
