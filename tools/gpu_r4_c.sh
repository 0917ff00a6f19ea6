#!/bin/bash
# round 4: the whole GPU suite + smoke + the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4c/gpu_tests.log
cat gpurun_out/r4c/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c/smoke.log 2>&1; tail -2 gpurun_out/r4c/smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
tail -4 gpurun_out/r4c/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4c/bench.json") if x.startswith("{")]
b=json.loads(l[-1])
print("headline", b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"], b["roofline"]["frac"], "cross", b.get("cross_check"))
for v in b["variants"]:
    r=v.get("roofline",{})
    print(" -", v["variant"][:70], "%.3e"%v["value"], v.get("kernel") or r.get("kernel"), r.get("kernel_ms_per_launch"), r.get("frac"))
print("cpu", b["cpu_baseline"]["value"])
PY
