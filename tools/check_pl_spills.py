"""What the register allocator did to pl_fused_kernel<5, 4> (the kernel runs at 256 VGPRs, 160 of them
accumulators, and a few registers more or less in a producer decide whether the compiler spills
INSIDE an MFMA loop, or spills a producer's requested operands one by one behind a full s_waitcnt
-- both seen in round 5, both worth 5-30 % of the kernel).  Compiles pliklite_kernels.hip to
assembly with the build's flags and reports, per innermost loop with MFMAs, the scratch
instructions in it, and per producer block the scratch instructions and the
`s_waitcnt vmcnt` -> `scratch_store` pairs (a requested operand spilled as it arrives).
    python tools/check_pl_spills.py [extra -D flags]      exit code 1 if a loop spills"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_lines(asm_text, name=r"pl_fused_kernelILi5ELi4"):
    m = re.search(r"^(_ZN4mcmc\S*" + name + r"\S*):\s.*?^\s*s_endpgm", asm_text, re.S | re.M)
    if not m:
        raise RuntimeError("kernel not found in the assembly")
    return m.group(0).split("\n")


def report(lines):
    label = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            label[m.group(1)] = i
    loops = []
    for i, line in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in label and label[m.group(1)] < i:
            body = lines[label[m.group(1)]:i + 1]
            n_mfma = sum("v_mfma" in x for x in body)
            if 0 < n_mfma <= 128:          # (the loops over chunks and sets hold hundreds)
                loops.append({"line": label[m.group(1)], "mfma": n_mfma,
                              "scratch": sum("scratch_" in x for x in body)})
    producers = []
    for i in [k for k, x in enumerate(lines) if "producer lane" in x]:
        j = i
        while j < len(lines) and "s_barrier" not in lines[j] and "chunk lane" not in lines[j] and j - i < 900:
            j += 1
        block = lines[i:j]
        producers.append({"line": i, "scratch": sum("scratch_" in x for x in block),
                          "spilled_on_arrival": sum(1 for x, y in zip(block, block[1:])
                                                    if "s_waitcnt vmcnt" in x and "scratch_store" in y)})
    return loops, producers


def compile_to_asm(extra=()):
    from cobaya_amd import build as B
    src = os.path.join(ROOT, "cobaya_amd", "csrc", "pliklite_kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "pl.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"]
        subprocess.run([B.hipcc(), *flags, *extra, "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True)
        with open(out) as f:
            return f.read()


if __name__ == "__main__":
    loops, producers = report(kernel_lines(compile_to_asm(sys.argv[1:])))
    for lp in loops:
        print("loop at line %(line)5d: %(mfma)3d MFMAs, %(scratch)d scratch instructions" % lp)
    for pr in producers:
        print("producer at line %(line)5d: %(scratch)d scratch instructions, %(spilled_on_arrival)d spilled on arrival" % pr)
    sys.exit(1 if any(lp["scratch"] for lp in loops) or any(p["spilled_on_arrival"] for p in producers) else 0)
