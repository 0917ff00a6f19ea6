# kernel timeline of a few bench steps: start/end of every kernel, gaps on the critical path
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline --no-variants --steps 8 --warmup 3 $BENCH_ARGS > $OUT/bench.json 2> $OUT/trace.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/timeline/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# last 3 step kernels and everything between
idx = [i for i, r in enumerate(rows) if "step_inc_kernel" in r["Kernel_Name"]]
lo = idx[-4]
prev_end = None
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mcmc::", "").split("(")[0][:28]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-28s q%-3s start %10.1f us  dur %8.1f us  gap-from-prev-end %7.1f us" % (name, r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3, gap))
    prev_end = max(prev_end or 0, e)
PY
