#!/bin/bash
# round 4: the variates staged through LDS (no quad-broadcast switch) -- parity, same-box A/B, counters
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -6 > $O/gpu_tests.log
cat $O/gpu_tests.log
VARIANTS="old lds" bash tools/gpu_r4_ab.sh
VARIANTS="old lds" BENCH_ARGS="--dim 100 --steps 10 --warmup 2" bash tools/gpu_r4_ab.sh 2>&1 | sed 's/^/d100 /'
VARIANTS="old lds" bash tools/gpu_r4_abpmc.sh
