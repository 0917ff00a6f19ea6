#!/bin/bash
# round 4: the periodic incremental kernel with the box support test + kept pairs: parity, then rates
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4per
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "periodic or incremental" 2>&1 | tail -6 > gpurun_out/r4per/tests.log
cat gpurun_out/r4per/tests.log
timeout 600 python tools/periodic_bench.py 30:1 30:4 64:2 100:1 100:8 128:2 2>&1 | grep incremental > gpurun_out/r4per/periodic_bench.log
cat gpurun_out/r4per/periodic_bench.log
