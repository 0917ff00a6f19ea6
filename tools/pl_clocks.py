"""Where a set of 64 walkers spends its time in pl_fused_kernel: the shader clock of every wave at
the phase boundaries of its second set (a library built with -DPL_DEBUG_CLOCKS by
tools/exp_pl_variants.sh; the stamps themselves cost a few per cent).
    MCMC_HIP_LIB=cobaya_amd/csrc/_exp/lib_clk.so MCMC_HIP_LIB_COMPAT=1 python tools/pl_clocks.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from cobaya_amd import engine as E  # noqa: E402
from cobaya_amd import pliklite as P  # noqa: E402
from tests.pliklite_common import sampling_problem  # noqa: E402

n_lin, W = 26, 65536
ds = P.synthetic_dataset(0)
target = P.BinnedGaussian.from_dataset(ds)
emu = P.synthetic_emulator(n_lin, ds.lmax)
kinds, a, b, C = sampling_problem(target, emu)
eng = E.Engine(n_lin + 1, W, group_size=256, seed=3)
eng.set_prior(kinds, a, b)
eng.set_target_binned_gaussian(target, emu, calib_index=n_lin)
eng.set_proposal_cov(C)
rng = np.random.default_rng(1)
eng.set_state(np.concatenate((emu.theta0, [1.0])) + rng.standard_normal((W, n_lin + 1)) @ np.linalg.cholesky(C).T)
eng.step(12)
eng.sync()
lib = ctypes.CDLL(os.environ["MCMC_HIP_LIB"])
log = np.zeros((256, 8, 64), dtype=np.uint64)
rc = lib.mcmc_hip_debug_pl_clocks(log.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = log.astype(np.int64)
NG = 5
names = ["produce(0)", "barrier"]
for m in range(NG):
    names += [f"c{m} early produce", f"c{m} loop S+L", f"c{m} loop L", f"c{m} loop none", f"c{m} late produce", f"c{m} barrier"]
d = np.diff(t[:, :, :3 + 6 * NG], axis=2)             # [block][wave][phase]
tot = t[:, :, 2 + 6 * NG] - t[:, :, 0]
print(f"a set of 64 walkers: {tot.mean():.0f} shader clocks (min {tot.min()}, max {tot.max()}) "
      f"= {tot.mean() / 2.4e3:.1f} us at 2.4 GHz; x 4 sets = {4 * tot.mean() / 2.4e6:.3f} ms")
print(f"{'phase':22s} {'q<4':>9s} {'q>=4':>9s}   (mean shader clocks over 256 workgroups)")
for i, nme in enumerate(names):
    print(f"{nme:22s} {d[:, :4, i].mean():9.0f} {d[:, 4:, i].mean():9.0f}")
# per chunk: the wall time of the chunk (barrier to barrier, any wave) against its MFMAs
print("chunk: clocks barrier-to-barrier | MFMAs per SIMD x 64 clocks | ratio")
mf = {"cons": [], "prod": 64}
for m in range(NG):
    t0 = t[:, 0, 2 + 6 * m].astype(float)
    t1 = t[:, 0, 8 + 6 * m].astype(float)
    # MFMAs of a SIMD's two waves in chunk m: groups above 2 x 128, the diagonal group 2 x 72,
    # (the first group lacks `shift` virtual tiles), + the producers of chunk m + 1 (2 x 32)
    cons = 2 * (128 * (NG - 1 - m) + 72)
    prod = 64 if m + 1 < NG else 0
    ideal = (cons + prod) * 64
    print(f"  {m}: {np.mean(t1 - t0):9.0f} | {ideal:9d} | {ideal / np.mean(t1 - t0):.3f}")
