#!/bin/bash
# round 4: the four per-configuration bench lines with the installed profiles/traffic.json (see tools/gpu_r4_final.sh, requote)
cd "$(dirname "$0")/.."
O=gpurun_out/requote; rm -rf $O; mkdir -p $O
rq() { TAG=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0.25 $* > $O/${TAG}_bench.json 2>/dev/null; tail -c 400 $O/${TAG}_bench.json | head -c 200; echo; }
rq r04
rq r04_full --evaluation full
rq r04_d100 --dim 100 --steps 10 --warmup 2
rq r04_pl --workload pliklite --steps 8 --warmup 2
