#!/bin/bash
# round 4: directions of a call's first launch formed AT that call (lazy): GPU suite, A/B against the eager order, timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/gpu_tests.log
for rep in 1 2 3; do for v in eager lazy; do
  E=0; [ $v = eager ] && E=1
  MCMC_HIP_EAGER_DIRECTIONS=$E timeout 300 python bench.py --no-cpu-baseline --no-variants --cross-check-seconds 0 > $O/b_${v}_$rep.json 2>> $O/err.log
  python - $O/b_${v}_$rep.json $v $rep <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], "value %.4g ms/step %.4f kernel %.4f"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"]))
PY
done; done
bash tools/gpu_r4_timeline.sh > $O/timeline_lazy.txt 2>&1; sed -n '/step_inc/,$p' $O/timeline_lazy.txt | head -48
