#!/bin/bash
# round 4: the kept pairs fetched in batches (KB) at d = 100: parity of the big-d kernels, then same-box A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
VARIANTS="new kb4 kb6 kb9 kb13" BENCH_ARGS="--dim 100 --steps 10 --warmup 2" bash tools/gpu_r4_ab.sh 2>&1 | sed 's/^/d100 /'
