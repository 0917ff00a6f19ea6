"""What the register allocator did to step_duo_mix_kernel (incremental_duo.hip): the kernel runs at 256
VGPRs per lane (two waves per SIMD) with 2 dq K .. 2 dq (K + 1) doubles of state in them, and whether it
spills INSIDE the step loop was decided by details (the addresses of the epilogue's stores, the order of
the plane reads, where x lives: round 6 -- three modes with x in registers ran 2.2 x slower than the
four-lane kernel, with x in LDS 1.27 x faster).  Compiles the kernels of the top of the range (dq = 6 .. 8)
to assembly with the build's flags and reports, per instantiation, VGPRs, spilled registers, and the
scratch instructions inside the step loop (the blocks the compiler marks `Depth=2`).
    python tools/check_duo_spills.py [dq_lo dq_hi]        exit code 1 if a step loop stores to scratch or reloads more than once"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_to_asm(dq_lo=6, dq_hi=8, extra=()):
    from cobaya_amd import build as B
    src = os.path.join(ROOT, "cobaya_amd", "csrc", "incremental_duo.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "duo.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"]
        subprocess.run([B.hipcc(), *flags, *extra, f"-DMCMC_DUO_DQ_LO={dq_lo}", f"-DMCMC_DUO_DQ_HI={dq_hi}",
                        "-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
        with open(out) as f:
            return f.read()


def report(asm_text):
    """[{dq, modes, unit_t, box, vgprs, spilled, scratch_in_loop}] for every step_duo_mix_kernel in the text"""
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S*step_duo_mix_kernel\S+)\n(?:.*\n){0,10}?\s+\.vgpr_count:\s+(\d+)\n"
                         r"\s+\.vgpr_spill_count:\s+(\d+)", asm_text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    out = []
    for m in re.finditer(r"^(_ZN4mcmc\S*step_duo_mix_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])E\S*):\s.*?^\s*s_endpgm",
                         asm_text, re.S | re.M):
        in_loop, n, n_st = False, 0, 0
        for line in m.group(0).split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", line):
                in_loop = "Depth=2" in line
            elif in_loop and line.strip().startswith("scratch_"):
                n += 1
                n_st += line.strip().startswith("scratch_store")
        vg, sp = meta.get(m.group(1), (None, None))
        out.append({"dq": int(m.group(2)), "modes": int(m.group(3)), "unit_t": m.group(4) == "1",
                    "box": m.group(5) == "1", "vgprs": vg, "spilled": sp, "scratch_in_loop": n,
                    "scratch_stores_in_loop": n_st})
    return out


if __name__ == "__main__":
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 8)
    rows = report(compile_to_asm(lo, hi))
    for r in rows:
        print(r)
    # (one reload in the burn-in branch of the general-bounds kernels is tolerated: a constant of the
    # bookkeeping that only runs while some walker is still burning in)
    sys.exit(1 if any(r["scratch_stores_in_loop"] or r["scratch_in_loop"] > 1 for r in rows) or not rows else 0)
