#!/bin/bash
# round 4: the checkpoint modes (host / reduce / device), tests + bench + kernel trace of each
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4ckpt; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py tests/test_gpu_multirank.py -m gpu -q -k "checkpoint or rccl or bench" 2>&1 | tail -8 | tee $O/tests.log
for mode in host reduce device; do
  extra=""; [ $mode != host ] && extra="--attach-comm"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o t -- python bench.py --no-variants --no-cpu-baseline --checkpoint-on $mode $extra --cross-check-seconds 0 > $O/bench_$mode.json 2> $O/bench_$mode.err
  timeout 600 python bench.py --no-variants --no-cpu-baseline --checkpoint-on $mode $extra --cross-check-seconds 0 > $O/bench_plain_$mode.json 2>> $O/bench_$mode.err
  python - "$O" "$mode" <<'PY'
import json,sys,sqlite3,glob
O,mode=sys.argv[1:3]
for tag in ("bench","bench_plain"):
    b=json.loads([x for x in open(f"{O}/{tag}_{mode}.json") if x.startswith("{")][-1])
    print(mode, tag, "value %.4g ms/step %.4f kernel %.4f"%(b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"]), b["config"]["checkpoint_on"], b["config"]["learn_checkpoints_in_timed_region"], b["collective"].get("backend"))
db=sqlite3.connect(glob.glob(f"{O}/prof_{mode}/*.db")[0])
rows=db.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc").fetchall()
with open(f"{O}/kernels_{mode}.txt","w") as f:
    for r in rows[:14]:
        line=f"{r[0][:90]:90s} n={r[1]:5d} avg={r[2]:9.2f} us total={r[3]:8.2f} ms"
        f.write(line+"\n")
        if "ckpt" in r[0] or "ccl" in r[0].lower() or "step_inc" in r[0]: print("   ",line)
PY
  rm -rf $O/prof_$mode
done
