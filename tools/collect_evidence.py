#!/usr/bin/env python3
"""Turns gpurun_out/final (written by `tools/gpu.sh evidence` on the GPU box) into the committed
evidence under profiles/: <rnd>_bench.json, <rnd>_gpu_tests.log, <rnd>_kernel_stats.txt,
<rnd>_pmc.txt and an entry of traffic.json (per-launch HBM bytes and VALU instructions of the
dominant kernel, which bench.py quotes in its roofline block).

    python tools/collect_evidence.py [round-prefix, default r01] [source dir under gpurun_out,
                                      default final; e.g. `r01_d100 d100` after a run into that directory]
"""
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
# (EVIDENCE_DST: on the GPU box the summaries are written under gpurun_out/ -- the result databases
# are too big to travel back -- and copied into profiles/ afterwards)
DST = os.environ.get("EVIDENCE_DST") or os.path.join(ROOT, "profiles")
CMD = "python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4"


def main():
    global SRC
    global CMD
    os.makedirs(DST, exist_ok=True)
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        SRC = os.path.join(ROOT, "gpurun_out", sys.argv[2])
    if os.path.exists(os.path.join(SRC, "cmd.txt")):
        with open(os.path.join(SRC, "cmd.txt")) as f:
            CMD = f.read().strip()
    shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{rnd}_bench.json"))
    if os.path.exists(os.path.join(SRC, "gpu_tests.log")):
        shutil.copy(os.path.join(SRC, "gpu_tests.log"), os.path.join(DST, f"{rnd}_gpu_tests.log"))
    with open(os.path.join(SRC, "bench.json")) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    dom = bench["roofline"]["kernel"].split("(")[0].strip().split("::")[-1].split("<")[0]

    c = sqlite3.connect(os.path.join(SRC, "trace", "t_results.db"))
    rows = c.execute("select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3 "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(DST, f"{rnd}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {CMD}   (MI355X)\n")
        if bench["roofline"].get("basis_on_second_stream"):
            f.write("# (basis_* and whiten_directions_kernel run on a second stream, beside the "
                    "moment snapshot and the y refresh of the main stream: pct is of the sum of "
                    "durations, not of wall time)\n")
        f.write("# name, calls, total_us, avg_us, pct\n")
        for n, k, t, a in rows:
            f.write(f"{n}, {k}, {t:.3f}, {a:.3f}, {100 * t / tot:.3f}\n")
        # the dominant kernel over the TIMED launches only (the table above also averages the
        # spin-up and warm-up launches, which run on colder clocks): the line's
        # roofline.kernel_ms_per_launch must agree with the first figure to ~1 %
        durs = [r[0] for r in c.execute(
            "select (end - start) / 1e3 from kernels where name like ? order by start",
            (f"%{dom}%",)).fetchall()]
        lps = max(1, int(round(bench["roofline"].get("kernel_launches_per_step", 1))))
        n_timed = bench["steps"] * lps
        n_cross = (bench.get("cross_check") or {}).get("steps", 0) * lps
        if len(durs) >= n_timed + n_cross:
            end = len(durs) - n_cross
            timed = durs[end - n_timed:end]
            f.write(f"# {dom}: the {n_timed} dispatches of the timed region: avg "
                    f"{sum(timed) / len(timed):.3f} us (bench line: "
                    f"{1e3 * bench['roofline']['kernel_ms_per_launch']:.3f} us by HIP events)\n")
            if n_cross:
                cross = durs[end:]
                f.write(f"# {dom}: the {n_cross} dispatches of the cross-check region: avg "
                        f"{sum(cross) / len(cross):.3f} us\n")

    vals = {}
    lines = []
    for p in ("pmc_fetch", "pmc_lds", "pmc_sq", "pmc_write", "pmc_mfma", "pmc_l2"):
        db = os.path.join(SRC, p, "p_results.db")
        if not os.path.exists(db):
            continue
        c = sqlite3.connect(db)
        for n, v, k, dur in c.execute(
                "select counter_name, avg(value), count(*), avg(duration) from counters_collection "
                "where kernel_name like ? group by counter_name", (f"%{dom}%",)):
            lines.append(f"{p}, {n}, {v:.6g}, {k}, {dur:.0f}")
            vals[n] = v
    wl = bench["config"]
    if not lines:   # a kernel-trace-only run: the counters of the last PMC run stay
        print(open(os.path.join(DST, f"{rnd}_kernel_stats.txt")).read())
        return
    with open(os.path.join(DST, f"{rnd}_pmc.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc <counters> (one pass per line group) -- {CMD}\n")
        f.write(f"# per-dispatch averages for {bench['roofline']['kernel']}, "
                f"{wl['walkers_per_gpu']} walkers, {wl['metropolis_steps_per_launch']} steps "
                "per call of the engine\n# pass, counter, avg value, dispatches, avg dispatch ns\n")
        f.write("\n".join(lines) + "\n")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        traffic = {
            "d": wl["d"], "walkers": wl["walkers_per_gpu"],
            # (steps per KERNEL launch: a call of the engine may hold several launches)
            "steps_per_launch": int(round(wl["metropolis_steps_per_launch"]
                                          / max(1.0, bench["roofline"].get("kernel_launches_per_step", 1)))),
            "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
            "sq_insts_valu": vals.get("SQ_INSTS_VALU"),
            "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; KiB; "
                    "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, "
                    "MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated"}
        import subprocess
        sys.path.insert(0, ROOT)
        from bench import csrc_sha16   # hash of the kernel sources the measurement was taken on
        traffic.update(csrc_sha16=csrc_sha16())
        traffic.update(kernel=bench["roofline"]["kernel"].split("(")[0].strip(),
                       pmc_file=f"profiles/{rnd}_pmc.txt",
                       commit=os.environ.get("EVIDENCE_COMMIT") or subprocess.run(
                           ["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True,
                           text=True).stdout.strip())
        tj = os.path.join(DST, "traffic.json")   # keyed table; bench.py looks its workload up
        table = {}
        if os.path.exists(tj):
            with open(tj) as f:
                table = json.load(f)
        table[f"{rnd}_d{wl['d']}_w{wl['walkers_per_gpu']}_spl{traffic['steps_per_launch']}"] = traffic
        with open(tj, "w") as f:
            json.dump(table, f, indent=1)
    print(open(os.path.join(DST, f"{rnd}_kernel_stats.txt")).read())
    print(open(os.path.join(DST, f"{rnd}_pmc.txt")).read())


if __name__ == "__main__":
    main()
