#!/bin/bash
# round 4, first check: the communicator tests + the launcher + a short bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_sampler.py -m gpu -x -q -k "rccl or multirank or ranks or launcher or bench or device_checkpoint or shards" 2>&1 | tail -25 > gpurun_out/r4a/tests.log
cat gpurun_out/r4a/tests.log
( time timeout 600 python bench.py --no-variants --cpu-seconds 3 ) > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -3 gpurun_out/r4a/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4a/bench.json") if x.startswith("{")]
b=json.loads(l[-1])
print("headline", b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms_per_launch"], b["roofline"]["frac"])
PY
