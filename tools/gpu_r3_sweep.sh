#!/bin/bash
# round 3: whole GPU suite, then the configuration sweep of tools/cliff_bench.py (-> profiles/r03_model_sweep.log)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3_sweep
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r3_sweep/gpu_tests.log
cat gpurun_out/r3_sweep/gpu_tests.log
timeout 900 python tools/cliff_bench.py 30:1:0 30:1:-1 30:1:1 30:1:4 30:1:8 30:1:12 30:2:0 30:4:0 30:5:0 30:8:0 30:16:0 30:2:1 30:5:3 \
   64:1:2 64:4:0 64:8:0 80:2:0 100:1:0 100:1:1 100:1:8 100:1:12 100:2:0 100:4:0 100:3:3 128:1:2 128:2:0 128:4:0 > gpurun_out/r3_sweep/sweep.log 2>&1
cat gpurun_out/r3_sweep/sweep.log
