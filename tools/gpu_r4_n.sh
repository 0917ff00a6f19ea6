#!/bin/bash
# round 4: |u|^2 by scalar load: parity of the incremental kernels, same-box A/B
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
VARIANTS="carry sload" bash tools/gpu_r4_ab.sh
VARIANTS="carry sload" BENCH_ARGS="--dim 100 --steps 10 --warmup 2" bash tools/gpu_r4_ab.sh | sed 's/^/d100 /'
