#!/bin/bash
# round 4: side kernels of the gap (moments, Haar basis): GPU suite, bench, timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/gpu_tests.log
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-variants --cross-check-seconds 0 > $O/b_$rep.json 2>> $O/err.log
  python - $O/b_$rep.json $rep <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(sys.argv[2], "value %.4g ms/step %.4f kernel %.4f gap %.4f"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"], b["ms_per_step"]-b["roofline"]["kernel_ms_per_launch"]))
PY
done
bash tools/gpu_r4_timeline.sh > $O/timeline.txt 2>&1; sed -n '/step_inc/,$p' $O/timeline.txt | head -34
