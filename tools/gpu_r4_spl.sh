#!/bin/bash
# round 4: launch length against the per-launch gap (config 2)
cd "$(dirname "$0")/.."
O=gpurun_out/r4spl; rm -rf $O; mkdir -p $O
for spl in 1200 2400 4800; do
  steps=$((120000 / spl))
  timeout 600 python bench.py --no-variants --no-cpu-baseline --cross-check-seconds 0 --steps-per-launch $spl --steps $steps --warmup 6 > $O/bench_$spl.json 2> $O/bench_$spl.err
  python - "$O/bench_$spl.json" $spl <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(sys.argv[2], "value %.4g ms/step %.4f kernel %.4f ckpts %s frac %.3f"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"], b["config"]["learn_checkpoints_in_timed_region"], b["roofline"]["frac"]))
PY
done
