"""Builds libmcmc_hip.so (hipcc, gfx950 only) in-tree under cobaya_amd/csrc/.

One object per compiled dimension (walker_kernels.hip with -DMCMC_D=<d>), compiled in
parallel, plus capi.hip; linked into cobaya_amd/csrc/libmcmc_hip.so.  hipcc cross-compiles
without a GPU, so this runs in the CPU-only build container.

    python -m cobaya_amd.build            # all dimensions 1..32
    MCMC_HIP_DIMS=2,3,30 python -m cobaya_amd.build   # quick developer build
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libmcmc_hip.so")
ARCH = "gfx950"
ALL_DIMS = list(range(1, 33))
BIG_DPS = [48, 56, 64, 72, 80, 88, 96, 100, 112, 120, 128]  # padded sizes of the d > 32 kernels
INC_DQ_RANGES = [(1, 8), (9, 16), (17, 24), (25, 32)]  # incremental_kernels.hip: ceil(d / 4)
DUO_DQ_RANGES = [(1, 8), (9, 12)]  # incremental_duo.hip: mixtures, two lanes per walker (two modes: d <= 48)
PAIR_DIMS = list(range(33, 57))  # 32 < d <= 48: walker_kernels.hip's two-wave step kernel alone

# -ffp-contract=off: the kernels' arithmetic order is part of the specification (fused
# operations are written as fma()); see DESIGN.md "Ensemble specification".
# -pragma-unroll-threshold: the operand-stream loops must be unrolled completely (their
# register arrays are only addressable with compile-time indices).
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-fno-fast-math", "-fvisibility=hidden", "-mllvm", "-pragma-unroll-threshold=1000000", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
         "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the mcmc_hip engine can only be built with ROCm")
    return exe


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _compile(src, obj, defines, stamp):
    stamp_file = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(stamp_file):
        with open(stamp_file) as f:
            if f.read() == stamp:
                return False
    cmd = [hipcc(), *FLAGS, *defines, "-c", src, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return True


def selected_dims():
    env = os.environ.get("MCMC_HIP_DIMS")
    if env:
        return sorted({int(s) for s in env.split(",") if s.strip()})
    return ALL_DIMS


def build(dims=None, jobs=None, verbose=True):
    """Compile every selected dimension and link the shared library. Returns its path."""
    dims = list(dims) if dims is not None else selected_dims()
    extra = os.environ.get("MCMC_HIP_EXTRA_FLAGS", "").split()  # developer experiments
    if extra:
        FLAGS.extend(f for f in extra if f not in FLAGS)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in ("det_math.h", "kernels.h", "short_log_table.h")]
    root_hdr = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "mcmc_hip.h")
    wk = os.path.join(CSRC, "walker_kernels.hip")
    capi = os.path.join(CSRC, "capi.hip")
    tasks = []
    for d in dims:
        stamp = _digest([wk] + hdrs, extra=f"{d}|{' '.join(FLAGS)}")
        tasks.append((wk, os.path.join(OBJ, f"walker_d{d}.o"), [f"-DMCMC_D={d}"], stamp))
    big = os.path.join(CSRC, "walker_kernels_big.hip")
    big_dps = [] if os.environ.get("MCMC_HIP_NO_BIG") else BIG_DPS
    for d in ([] if os.environ.get("MCMC_HIP_NO_BIG") else PAIR_DIMS):
        stamp = _digest([wk] + hdrs, extra=f"{d}|{' '.join(FLAGS)}")
        tasks.append((wk, os.path.join(OBJ, f"walker_d{d}.o"), [f"-DMCMC_D={d}"], stamp))
    for dp in big_dps:
        stamp = _digest([big] + hdrs, extra=f"big{dp}|{' '.join(FLAGS)}")
        tasks.append((big, os.path.join(OBJ, f"walker_big{dp}.o"), [f"-DMCMC_DP={dp}"], stamp))
    blocked = os.path.join(CSRC, "blocked_kernels.hip")
    tasks.append((blocked, os.path.join(OBJ, "blocked.o"), [],
                  _digest([blocked] + hdrs, extra=" ".join(FLAGS))))
    general = os.path.join(CSRC, "general_kernels.hip")
    tasks.append((general, os.path.join(OBJ, "general.o"), [],
                  _digest([general] + hdrs, extra=" ".join(FLAGS))))
    pl = os.path.join(CSRC, "pliklite_kernels.hip")
    pl_hdr = os.path.join(CSRC, "pliklite_args.h")
    tasks.append((pl, os.path.join(OBJ, "pliklite.o"), [],
                  _digest([pl, pl_hdr] + hdrs, extra=" ".join(FLAGS))))
    ck = os.path.join(CSRC, "checkpoint_kernels.hip")
    ck_hdr = os.path.join(CSRC, "checkpoint_args.h")
    tasks.append((ck, os.path.join(OBJ, "checkpoint.o"), [],
                  _digest([ck, ck_hdr], extra=" ".join(FLAGS))))
    comm = os.path.join(CSRC, "comm.hip")   # the RCCL communicator (bound at run time)
    comm_hdr = os.path.join(CSRC, "comm.h")
    tasks.append((comm, os.path.join(OBJ, "comm.o"), [],
                  _digest([comm, comm_hdr, root_hdr], extra=" ".join(FLAGS))))
    inc = os.path.join(CSRC, "incremental_kernels.hip")
    inc_hdr = os.path.join(CSRC, "incremental_common.h")
    for lo_, hi_ in INC_DQ_RANGES:
        tasks.append((inc, os.path.join(OBJ, f"incremental_{lo_}.o"),
                      [f"-DMCMC_DQ_LO={lo_}", f"-DMCMC_DQ_HI={hi_}"],
                      _digest([inc, inc_hdr] + hdrs, extra=f"inc{lo_}-{hi_}|{' '.join(FLAGS)}")))
    for lo_, hi_ in INC_DQ_RANGES:   # the EMIT instantiations of step_inc_kernel (emit: chains)
        tasks.append((inc, os.path.join(OBJ, f"incremental_emit_{lo_}.o"),
                      ["-DMCMC_INC_EMIT_TU", f"-DMCMC_DQ_LO={lo_}", f"-DMCMC_DQ_HI={hi_}"],
                      _digest([inc, inc_hdr] + hdrs, extra=f"incemit{lo_}-{hi_}|{' '.join(FLAGS)}")))
    anyk = os.path.join(CSRC, "incremental_any.hip")   # the general incremental kernel
    for part in (0, 1, 2):   # the LDS kernel + KM = 4 | KM = 8 | KM = 16 register planes
        tasks.append((anyk, os.path.join(OBJ, f"incremental_any_{part}.o"), [f"-DANY_PART={part}"],
                      _digest([anyk, inc_hdr] + hdrs, extra=f"any{part}|{' '.join(FLAGS)}")))
    duo = os.path.join(CSRC, "incremental_duo.hip")   # two lanes per walker (round 6)
    for lo_, hi_ in DUO_DQ_RANGES:
        tasks.append((duo, os.path.join(OBJ, f"incremental_duo_{lo_}.o"),
                      [f"-DMCMC_DUO_DQ_LO={lo_}", f"-DMCMC_DUO_DQ_HI={hi_}"],
                      _digest([duo, inc_hdr] + hdrs, extra=f"duo{lo_}-{hi_}|{' '.join(FLAGS)}")))
    tasks.append((capi, os.path.join(OBJ, "capi.o"), [],
                  _digest([capi, root_hdr, pl_hdr, ck_hdr, comm_hdr] + hdrs, extra=" ".join(FLAGS))))
    jobs = jobs or min(len(tasks), os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        rebuilt = list(ex.map(lambda t: _compile(*t), tasks))
    objs = [t[1] for t in tasks]
    link_stamp = _digest(objs)
    stamp_file = LIB + ".stamp"
    need_link = any(rebuilt) or not os.path.exists(LIB)
    if not need_link and os.path.exists(stamp_file):
        with open(stamp_file) as f:
            need_link = f.read() != link_stamp
    if need_link:
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-ldl", "-o", LIB]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed: {' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        with open(stamp_file, "w") as f:
            f.write(link_stamp)
    if verbose:
        print(f"[cobaya_amd.build] {LIB}: dims {dims[0]}..{dims[-1]} ({len(dims)}), "
              f"{sum(rebuilt)} objects rebuilt")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
