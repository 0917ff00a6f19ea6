#!/bin/bash
# round 4: PMC passes of the plik-lite step kernels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4plpmc; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --workload pliklite --no-cpu-baseline --steps 8 --warmup 2 --cross-check-seconds 0"
echo "$CMD" > $OUT/cmd.txt
timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $OUT/pmc_lds -o p -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_l2 -o p -- $CMD > $OUT/pmc_l2.log 2>&1
python - <<'PY'
import sqlite3, os, glob
out = "gpurun_out/r4plpmc"
with open(os.path.join(out, "summary.txt"), "w") as f:
    db = glob.glob(os.path.join(out, "trace", "**", "*results.db"), recursive=True)
    if db:
        c = sqlite3.connect(db[0])
        rows = c.execute("select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3 from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        f.write("# kernel-trace: name, calls, total_us, avg_us, pct\n")
        for n, k, t, a in rows:
            f.write(f"{n}, {k}, {t:.3f}, {a:.3f}, {100 * t / tot:.3f}\n")
    for p in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_l2"):
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
cat $OUT/summary.txt; tail -2 $OUT/bench.err
