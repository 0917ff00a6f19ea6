#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 result databases.

    python tools/pmc_summary.py <dir> [kernel-name-substring]

Walks <dir> for *_results.db (rocprofv3 --pmc / --kernel-trace output, sqlite), and prints for
every kernel whose name contains the substring: dispatches, average duration, and the average
per-dispatch value of every counter collected (SQ_* counters are in quad-cycles, summed over
the chip; FETCH_SIZE/WRITE_SIZE in KiB).
"""
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for dp, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if not f.endswith("_results.db"):
                continue
            c = sqlite3.connect(os.path.join(dp, f))
            try:
                rows = c.execute(
                    "select kernel_name, counter_name, avg(value), count(*), avg(duration) "
                    "from counters_collection where kernel_name like ? "
                    "group by kernel_name, counter_name", (f"%{pat}%",)).fetchall()
            except sqlite3.Error:
                rows = []
            for k, n, v, cnt, dur in rows:
                print(f"{os.path.relpath(dp, root)}, {k[:60]}, {n}, {v:.6g}, {cnt}, {dur:.0f}")
            if not rows:
                try:
                    for k, cnt, dur in c.execute(
                            "select name, count(*), avg(end - start) from kernels "
                            "where name like ? group by name", (f"%{pat}%",)):
                        print(f"{os.path.relpath(dp, root)}, {k[:60]}, duration_ns, {dur:.0f}, {cnt}")
                except sqlite3.Error as e:
                    print(f"{dp}/{f}: {e}")


if __name__ == "__main__":
    main()
