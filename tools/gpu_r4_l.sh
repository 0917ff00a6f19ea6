#!/bin/bash
# round 4: PIPE = 4 for dq 5..12 at three / four waves: GPU suite, then same-box A/B at config 2 and the config-5 shape
cd "$(dirname "$0")/.."
O=gpurun_out/r4l; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/gpu_tests.log
VARIANTS="new pipe" bash tools/gpu_r4_ab.sh
VARIANTS="new pipe" BENCH_ARGS="--dim 48 --steps 20 --warmup 4" bash tools/gpu_r4_ab.sh | sed 's/^/d48 /'
