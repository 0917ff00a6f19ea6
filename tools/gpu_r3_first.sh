#!/bin/bash
# round 3, first GPU run: the binned (plik-lite) target's parity tests, then the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pliklite.py -x -q 2>&1 | tail -40 > gpurun_out/r3_pl_tests.log
cat gpurun_out/r3_pl_tests.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r3_gpu_tests.log
cat gpurun_out/r3_gpu_tests.log
