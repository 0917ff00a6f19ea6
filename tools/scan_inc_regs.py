#!/usr/bin/env python3
"""Register allocation of every incremental kernel (step / drag / mix, every DQ and MODE), from
the compiler's assembly: VGPRs, scratch bytes, and the waves per SIMD the launch bounds ask for.

    python tools/scan_inc_regs.py [--jobs 4]

A kernel with more than a few dozen scratch bytes spills inside its step loop: the table
`inc_min_waves` in incremental_kernels.hip has to be lowered for it."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.build import FLAGS, INC_DQ_RANGES, hipcc  # noqa: E402


def scan(rng):
    lo, hi = rng
    out = os.path.join(tempfile.gettempdir(), f"inc_{lo}.s")
    subprocess.run([hipcc()] + FLAGS + ["-S", "--cuda-device-only", f"-DMCMC_DQ_LO={lo}",
                                        f"-DMCMC_DQ_HI={hi}", "-o", out,
                                        os.path.join(ROOT, "cobaya_amd/csrc/incremental_kernels.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    rows = []
    text = open(out).read()
    # (ROCm 7.2 writes these as expressions over the callees: `max(128, .L…pair_tail….num_vgpr)`,
    # `96+max(…)`: the leading number is the kernel's own)
    for m in re.finditer(r"\.set (\S+)\.num_vgpr, (?:max\()?(\d+)", text):
        name = m.group(1)
        if name.startswith(".L"):
            continue
        seg = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text)
        dem = subprocess.run(["c++filt", name], capture_output=True,
                             text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::|mcmc::|void |\(.*", "", dem)
        rows.append((short, int(m.group(2)), int(seg.group(1)) if seg else -1))
    return rows


if __name__ == "__main__":
    with ThreadPoolExecutor(4) as ex:
        for rows in ex.map(scan, INC_DQ_RANGES):
            for name, vgpr, scratch in rows:
                if "step_inc" in name or "drag_inc" in name:
                    flag = "  <-- spills" if scratch > 64 else ""
                    print(f"{name:44s} vgpr {vgpr:4d} scratch {scratch:5d}{flag}")
