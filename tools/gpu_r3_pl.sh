#!/bin/bash
# round 3: plik-lite kernel evidence -- tests, timing tool, bench line, rocprofv3 kernel trace and PMC passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3_pl; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pliklite.py -q 2>&1 | tail -5 > $OUT/tests.log
python tools/pliklite_bench.py 26 65536 40 > $OUT/tool_26.log 2>&1
python tools/pliklite_bench.py 6 65536 40 > $OUT/tool_6.log 2>&1
CMD="python bench.py --workload pliklite --no-cpu-baseline --steps 8 --warmup 2"
echo "$CMD" > $OUT/cmd.txt
timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_l2 -o p -- $CMD > $OUT/pmc_l2.log 2>&1
rocprofv3 --list-avail > $OUT/counters_avail.txt 2>&1
# keep the databases small: only the summaries travel back
python - <<'PY'
import sqlite3, os, glob
out = "gpurun_out/r3_pl"
with open(os.path.join(out, "summary.txt"), "w") as f:
    db = glob.glob(os.path.join(out, "trace", "**", "*results.db"), recursive=True)
    if db:
        c = sqlite3.connect(db[0])
        rows = c.execute("select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3 from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        f.write("# kernel-trace: name, calls, total_us, avg_us, pct\n")
        for n, k, t, a in rows:
            f.write(f"{n}, {k}, {t:.3f}, {a:.3f}, {100 * t / tot:.3f}\n")
    for p in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_l2"):
        db = glob.glob(os.path.join(out, p, "**", "*results.db"), recursive=True)
        if not db:
            f.write(f"# {p}: no database\n")
            continue
        c = sqlite3.connect(db[0])
        f.write(f"# {p}: kernel, counter, avg value, dispatches, avg dispatch ns\n")
        for kn, n, v, k, dur in c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%pl_%' group by kernel_name, counter_name"):
            f.write(f"{kn.split('(')[0][-40:]}, {n}, {v:.6g}, {k}, {dur:.0f}\n")
PY
find $OUT -name "*.db" -size +20M -delete
cat $OUT/tests.log $OUT/tool_26.log $OUT/tool_6.log $OUT/summary.txt; tail -3 $OUT/bench.err; head -c 1500 $OUT/bench.json
