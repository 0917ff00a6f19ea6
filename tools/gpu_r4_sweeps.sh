#!/bin/bash
# round 4: the dimension sweep (incremental evaluation, 65 536 walkers) and the model-shape sweep at the final sources
cd "$(dirname "$0")/.."
O=gpurun_out/r4sweeps; rm -rf $O; mkdir -p $O
INC_ONLY=1 timeout 900 python tools/inc_bench.py 2 4 8 12 16 20 24 27 30 32 36 40 44 48 52 56 60 64 72 80 96 100 112 128 > $O/dimension_sweep.log 2>&1
timeout 900 python tools/cliff_bench.py > $O/model_sweep.log 2>&1
tail -3 $O/dimension_sweep.log; tail -3 $O/model_sweep.log
