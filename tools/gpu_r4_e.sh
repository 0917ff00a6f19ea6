#!/bin/bash
# round 4: after the instruction diet of step_inc_kernel -- parity of everything incremental, then the bench with variants
cd "$(dirname "$0")/.."
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -12 > $O/gpu_tests.log
cat $O/gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - "$O/bench.json" <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
r=b["roofline"]
print("headline %.4g ms/step %.4f kernel %.4f frac %.3f cross %s"%(b["value"], b["ms_per_step"], r["kernel_ms_per_launch"], r["frac"], b.get("cross_check",{}).get("value")))
for v in b["variants"]:
    print("  %.4g  %.3f ms  %s | %s"%(v["value"], v["ms_per_step"], v.get("kernel_ms_per_launch"), v["variant"][:70]))
PY
