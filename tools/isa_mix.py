#!/usr/bin/env python3
"""Instruction mix of the hot step kernels, from the compiler's own assembly.

    python tools/isa_mix.py > profiles/r02_isa_mix.txt

Compiles incremental_kernels.hip (DQ 1..8) and walker_kernels.hip (-DMCMC_D=30) to gfx950
assembly (hipcc -S --cuda-device-only, the flags of cobaya_amd/build.py), cuts out
`step_inc_kernel<8, 0, true>` and `step_pair_kernel<true, false>` and counts instruction
classes over each whole kernel and over its innermost loops (the blocks between a loop label
and the backward branch to it).  Static counts: what the loop bodies contain, not how often
each executes -- the executed totals are SQ_INSTS_VALU in profiles/r02*_pmc.txt."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.build import FLAGS, hipcc  # noqa: E402

CLASSES = [
    ("fp64 fma", r"^v_(fma|fmac)_f64"),
    ("fp64 add/mul", r"^v_(add|mul|ldexp)_f64"),
    ("fp64 compare", r"^v_cmpx?_\w+_f64|^v_cmp_class_f64"),
    ("fp64 div/sqrt/rcp/rsq + helpers", r"^v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|trig_preop|frexp\w*|fract|floor|rndne)_f64"),
    ("cvt", r"^v_cvt_"),
    ("int mul (Philox)", r"^v_mul_(hi|lo)_u32|^v_mad_u64_u32"),
    ("int / logic", r"^v_(xor|and|or|not|lshlrev|lshrrev|ashrrev|lshl|lshr|ashr|add|sub|subrev|addc|subb|subbrev|bfe|bfi|bfrev|perm|alignbit|mad|add3|lshl_add|lshl_or|and_or|or3|xad|bitop3|min|max|med3|mbcnt_lo|mbcnt_hi|sad)\w*_(u32|i32|b32|u64|b64|u16|i64_i32|u32_b32)"),
    ("select (v_cndmask)", r"^v_cndmask"),
    ("move", r"^v_mov_b(32|64)(?!.*dpp)|^v_accvgpr|^v_readlane|^v_readfirstlane|^v_writelane|^v_swap"),
    ("DPP move", r"dpp|quad_perm|row_"),
    ("int compare", r"^v_cmpx?_\w+_(u32|i32|u64|i64|u16)"),
    ("LDS", r"^ds_"),
    ("global / scratch / buffer memory", r"^(global|scratch|buffer|flat)_"),
    ("scalar memory", r"^s_load|^s_buffer_load"),
    ("scalar ALU / branch / waitcnt", r"^s_"),
    ("MFMA", r"^v_mfma"),
]


def classify(op, line):
    if "dpp" in line or "quad_perm" in line:
        return "DPP move"
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other vector" if op.startswith("v_") else "other"


def assembly(src, defines):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [hipcc(), *[f for f in FLAGS if f not in ("-fPIC",)], *defines, "-S",
               "--cuda-device-only", src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        with open(out) as f:
            return f.read()


def kernel_body(asm, mangled_re):
    m = re.search(r"^(" + mangled_re + r"):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel {mangled_re} not found")
    return m.group(1), m.group(2).splitlines()


def count(lines):
    tally = {}
    n = 0
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op, t)
        tally[c] = tally.get(c, 0) + 1
        n += 1
    return n, tally


def loops(lines):
    """(label, first line, last line) of every innermost loop: a label with a later backward
    branch to it and no other loop wholly inside."""
    labels = {ln.split(":")[0].strip(): i for i, ln in enumerate(lines)
              if re.match(r"^\.LBB\d+_\d+:", ln.strip())}
    found = []
    for i, ln in enumerate(lines):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            found.append((m.group(1), labels[m.group(1)], i))
    inner = [a for a in found if not any(b is not a and a[1] <= b[1] and b[2] <= a[2] and (b[1], b[2]) != (a[1], a[2])
                                         for b in found)]
    return sorted(set(inner), key=lambda t: t[1])


def report(title, name, lines):
    print(f"== {title}\n   {name}")
    n, tally = count(lines)
    order = [c for c, _ in CLASSES] + ["other vector", "other"]
    print(f"   whole kernel: {n} instructions")
    for c in order:
        if tally.get(c):
            print(f"     {c:38s} {tally[c]:5d}")
    for lab, a, b in loops(lines):
        m, t = count(lines[a:b + 1])
        if m < 24:
            continue
        valu = sum(v for k, v in t.items() if k not in ("LDS", "global / scratch / buffer memory",
                                                        "scalar memory", "scalar ALU / branch / waitcnt",
                                                        "other"))
        print(f"   innermost loop {lab} (lines {a}-{b}): {m} instructions, {valu} VALU")
        for c in order:
            if t.get(c):
                print(f"     {c:38s} {t[c]:5d}")
    print()


def main():
    csrc = os.path.join(ROOT, "cobaya_amd", "csrc")
    inc = assembly(os.path.join(csrc, "incremental_kernels.hip"), ["-DMCMC_DQ_LO=1", "-DMCMC_DQ_HI=8"])
    name, body = kernel_body(inc, r"_ZN4mcmc12_GLOBAL__N_115step_inc_kernelILi8ELi0ELb1ELb0ELb0ELb0EEEvNS_11IncStepArgsE")
    report("step_inc_kernel<8, 0, true>  (d = 30, incremental evaluation; 16 walkers per wave: the "
           "step loop is the loop holding the ds_read_b128 / v_fma_f64 body; PairRng is the block "
           "with the v_mul_hi_u32 Philox rounds, entered every eighth step)", name, body)
    wk = assembly(os.path.join(csrc, "walker_kernels.hip"), ["-DMCMC_D=30"])
    name, body = kernel_body(wk, r"_ZN4mcmc12_GLOBAL__N_116step_pair_kernelILb1ELb0EEEvNS_8StepArgsE")
    report("step_pair_kernel<true, false>  (d = 30, evaluation: full; two waves per 64 walkers, "
           "both roles in one kernel body)", name, body)


if __name__ == "__main__":
    main()
