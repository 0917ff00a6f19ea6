#!/bin/bash
# round 4: checkpoint_lag 1 vs 2 on one GPU now that a launch (0.91 ms) is shorter than the host's checkpoint pass
cd "$(dirname "$0")/.."
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for lag in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-variants --checkpoint-lag $lag --cross-check-seconds 1.0 > $O/b_${lag}_$rep.json 2>> $O/err.log
  python - $O/b_${lag}_$rep.json $lag $rep <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print("lag", sys.argv[2], sys.argv[3], "value %.4g ms/step %.4f kernel %.4f | cross-check %.4g (%.4f ms/step over %d steps)"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"], b["cross_check"]["value"], b["cross_check"]["ms_per_step"], b["cross_check"]["steps"]))
PY
done; done
