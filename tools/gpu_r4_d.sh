#!/bin/bash
# round 4: after the tail redraw -- parity of everything that draws paired variates, new Tier-C tests, bench headline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4d
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4d/gpu_tests.log
cat gpurun_out/r4d/gpu_tests.log
timeout 600 python bench.py --no-variants --no-cpu-baseline > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err
python - <<'PY'
import json
b=json.loads([x for x in open("gpurun_out/r4d/bench.json") if x.startswith("{")][-1])
print("headline", b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"], b["cross_check"])
PY
