#!/bin/bash
# round 4: same-box A/B of builds of the library (MCMC_HIP_LIB), headline bench alternated
cd "$(dirname "$0")/.."
O=gpurun_out/r4ab; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for v in ${VARIANTS:-old new}; do
    MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_$v.so timeout 300 python bench.py --no-cpu-baseline --no-variants --cross-check-seconds 0 ${BENCH_ARGS} > $O/b_${v}_$rep.json 2>> $O/err.log
    python - $O/b_${v}_$rep.json $v $rep <<'PY'
import json,sys
b=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], "value %.4g ms/step %.4f kernel %.4f"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"]))
PY
  done
done
