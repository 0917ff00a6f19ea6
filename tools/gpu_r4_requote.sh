#!/bin/bash
# round 4: the bench lines again on the installed profiles/traffic.json (after a change of bench.py alone): the four per-configuration lines and the full line
cd "$(dirname "$0")/.."
O=gpurun_out/requote; rm -rf $O; mkdir -p $O
rq() { TAG=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0.25 $* > $O/${TAG}_bench.json 2>/dev/null; }
rq r04
rq r04_full --evaluation full
rq r04_d100 --dim 100 --steps 10 --warmup 2
rq r04_pl --workload pliklite --steps 8 --warmup 2
timeout 900 python bench.py > $O/r04_bench_full_line.json 2> $O/full.err
tail -c 300 $O/r04_bench_full_line.json
