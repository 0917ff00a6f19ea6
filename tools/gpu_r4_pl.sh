#!/bin/bash
# round 4: plik-lite parity + its bench line (kernel times of the three step kernels)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4pl
timeout 900 python -m pytest tests/test_gpu_pliklite.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4pl/tests.log
cat gpurun_out/r4pl/tests.log
timeout 600 python bench.py --workload pliklite --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4pl/bench.json 2> gpurun_out/r4pl/bench.err
tail -3 gpurun_out/r4pl/bench.err
python - <<'PY'
import json
b=json.loads([x for x in open("gpurun_out/r4pl/bench.json") if x.startswith("{")][-1])
r=b["roofline"]
print("value %.4e  step_ms %.4f  chi2_ms %.4f frac %.3f  other %s" % (b["value"], r["metropolis_step_ms"], r["kernel_ms_per_launch"], r["frac"], r["other_kernels_ms_per_metropolis_step"]))
PY
