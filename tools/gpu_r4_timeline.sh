#!/bin/bash
# round 4: timeline of the kernels between two step launches (config 2, host checkpoint)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4tl; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o t -- python bench.py --no-variants --no-cpu-baseline --cross-check-seconds 0 ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err
python - "$O" <<'PY'
import sys,sqlite3,glob
O=sys.argv[1]
db=sqlite3.connect(glob.glob(f"{O}/prof/*.db")[0])
cols=[r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
rows=db.execute("select start, end, name, queue_id, stream_id from kernels order by start").fetchall() if "stream_id" in cols else db.execute("select start, end, name, queue_id, 0 from kernels order by start").fetchall()
try:
    mc=db.execute("select start, end, name, 0, 0 from memory_copies order by start").fetchall()
except Exception as e:
    print("no memory_copies view:", e); mc=[]
ev=sorted(rows+mc)
# find the 100th step kernel and print 3 launches from there
idx=[i for i,r in enumerate(ev) if "step_inc" in r[2]]
i0=idx[len(idx)//2]; i1=idx[len(idx)//2+5]
t0=ev[i0][0]
with open(f"{O}/timeline.txt","w") as f:
    for r in ev[i0:i1+1]:
        line=f"{(r[0]-t0)/1e3:10.2f} {(r[1]-t0)/1e3:10.2f} dur {(r[1]-r[0])/1e3:8.2f} us q{r[3]} s{r[4]} {r[2][:80]}"
        f.write(line+"\n"); print(line)
PY
rm -rf $O/prof
