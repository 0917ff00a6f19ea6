#!/bin/bash
# round 4: plik-lite parity + its bench line (fused vs unfused)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4pl
timeout 900 python -m pytest tests/test_gpu_pliklite.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4pl/tests.log
cat gpurun_out/r4pl/tests.log
for mode in fused unfused; do
  if [ $mode = unfused ]; then export MCMC_HIP_PL_UNFUSED=1; else unset MCMC_HIP_PL_UNFUSED; fi
  timeout 600 python bench.py --workload pliklite --steps 8 --warmup 2 --no-cpu-baseline --cross-check-seconds 0 > gpurun_out/r4pl/bench_$mode.json 2> gpurun_out/r4pl/bench_$mode.err
  tail -2 gpurun_out/r4pl/bench_$mode.err
  python - <<PY
import json
b=json.loads([x for x in open("gpurun_out/r4pl/bench_$mode.json") if x.startswith("{")][-1])
r=b["roofline"]
print("$mode value %.4e  step_ms %.4f  kernel %s %.4f ms frac %.3f  other %s" % (b["value"], r["metropolis_step_ms"], r["kernel"], r["kernel_ms_per_launch"], r["frac"], r["other_kernels_ms_per_metropolis_step"]))
PY
done
unset MCMC_HIP_PL_UNFUSED
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dragging or blocked or blocking or refuses" 2>&1 | tail -5
