#!/bin/bash
# round 3: whole GPU suite + smoke + the full bench line (all variants)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3_check
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r3_check/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_check/smoke.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r3_check/bench.json 2> gpurun_out/r3_check/bench.err
cat gpurun_out/r3_check/gpu_tests.log; tail -3 gpurun_out/r3_check/smoke.log; tail -5 gpurun_out/r3_check/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3_check/bench.json") if x.startswith("{")]
b=json.loads(l[-1])
print("headline", b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"], b["roofline"]["frac"])
for v in b["variants"]:
    r=v.get("roofline",{})
    print(" -", v["variant"][:60], "%.3e"%v["value"], r.get("kernel"), r.get("kernel_ms_per_launch"), r.get("bound"), r.get("frac"))
print("cpu", b["cpu_baseline"]["value"], b["variants"][-1].get("cpu_baseline"))
PY
