#!/bin/bash
# round 4: emit: chains with the rows retained on the host -- the drain ring as the store vs drain_copy
cd "$(dirname "$0")/.."
O=gpurun_out/r4rows; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sampler.py -m gpu -q -k "chains or drain or rows" 2>&1 | tail -5 | tee $O/tests.log
timeout 1500 python bench.py --no-cpu-baseline --cross-check-seconds 0 --steps 20 --warmup 4 > $O/bench.json 2> $O/bench.err
python - "$O/bench.json" <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print("headline %.4g"%b["value"])
for v in b["variants"]:
    if "chains" in v["variant"]:
        print("%.4g evals/s  %.1f ms/step  rows/s %.3g  B/s %.3g"%(v["value"], v["ms_per_step"], v["accepted_rows_per_s"], v["row_bytes_per_s"]), {k:v[k] for k in v if k.startswith("rows_") or k in ("drain_slots","stored_rows_copied_on_host")}, "|", v["variant"][:110])
PY
tail -3 $O/bench.err
