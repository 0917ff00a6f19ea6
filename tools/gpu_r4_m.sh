#!/bin/bash
# round 4: the carried log-likelihood (step_inc_kernel): the GPU suite, then same-box A/B at config 2 and d = 100
cd "$(dirname "$0")/.."
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/gpu_tests.log
VARIANTS="pipe carry" bash tools/gpu_r4_ab.sh
VARIANTS="pipe carry" BENCH_ARGS="--dim 100 --steps 10 --warmup 2" bash tools/gpu_r4_ab.sh | sed 's/^/d100 /'
