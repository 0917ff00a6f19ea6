# The committed bench lines (after profiles/traffic.json has the PMC figures of the current build):
# headline with variants and CPU baseline, evaluation: full, d = 100.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lines
timeout 600 python bench.py > gpurun_out/lines/bench_full_line.json 2> gpurun_out/lines/err.log
timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 > gpurun_out/lines/bench.json 2>> gpurun_out/lines/err.log
timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --evaluation full > gpurun_out/lines/full_bench.json 2>> gpurun_out/lines/err.log
timeout 300 python bench.py --no-cpu-baseline --no-variants --dim 100 --steps 10 --warmup 2 > gpurun_out/lines/d100_bench.json 2>> gpurun_out/lines/err.log
for f in bench_full_line bench full_bench d100_bench; do python - gpurun_out/lines/$f.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], "%.4g evals/s" % j["value"], "ms/step %.4f" % j["ms_per_step"], r["bound"], "frac %.3f" % r["frac"], r["kernel"], "kernel ms %.4f" % r["kernel_ms_per_launch"])
PY
done
