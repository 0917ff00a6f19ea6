#!/usr/bin/env python3
"""Start / end clock and hardware placement of every workgroup of the incremental step kernel
(config 2), read back from a build with -DEXP_BLOCK_TIMES:

    DQ_LO=1 DQ_HI=8 tools/exp_inc_variants.sh bt "-DEXP_BLOCK_TIMES"
    MCMC_HIP_LIB=$PWD/cobaya_amd/csrc/_exp/lib_bt.so python tools/block_times.py

(add -DEXP_NO_ROTATE to see the age-ordered arbitration: workgroups ending between
0.68 and 1.20 ms in three clusters)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cobaya_amd import engine as E
lib = E.load_library()
raw = C.CDLL(os.environ["MCMC_HIP_LIB"])
g = np.load(os.path.join(os.path.dirname(E.__file__), "..", "tests", "golden", "targets.npz"))
mean, cov = g["mean_d30"], g["cov_d30"]
d, W = 30, 65536
eng = E.Engine(d, W, group_size=256, seed=1, incremental=True, basis_group_size=1024)
eng.set_prior([0]*d, [0.0]*d, [1.0]*d); eng.set_target_gaussian_mixture([mean], [cov]); eng.set_proposal_cov(cov)
rng = np.random.default_rng(1)
eng.set_state(np.clip(mean + rng.standard_normal((W, d))*np.sqrt(np.diag(cov)), 1e-6, 1-1e-6))
for _ in range(40): eng.step(1200)
eng.sync()
buf = np.zeros(8192, dtype=np.uint64)
rc = raw.mcmc_hip_debug_block_times(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(4096, 2)[:1024].astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # wall_clock64: 100 MHz -> us
print("rc", rc, "start us: min %.1f max %.1f" % (st.min(), st.max()))
print("end   us: min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (en.min(), *np.percentile(en, [10, 50, 90]), en.max()))
dur = en - st
print("dur   us: min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max()))
pl = np.zeros(2 * 4 * 4096, dtype=np.uint32)
raw.mcmc_hip_debug_wave_place(pl.ctypes.data_as(C.c_void_p))
pl = pl.reshape(4096, 4, 2)[:1024]
hw, xcc = pl[:, :, 0], pl[:, :, 1] & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu          # per wave
print("distinct CUs used:", len(np.unique(cuid)))
from collections import Counter
blocks_per_cu = Counter(cuid[:, 0].tolist())
print("blocks per CU histogram:", sorted(Counter(blocks_per_cu.values()).items()))
waves_per_simd = Counter((cuid * 4 + simd).ravel().tolist())
print("waves per SIMD histogram:", sorted(Counter(waves_per_simd.values()).items()))
same = [(len(set(simd[b].tolist()))) for b in range(1024)]
print("distinct SIMDs per block histogram:", sorted(Counter(same).items()))
nb = np.array([blocks_per_cu[c] for c in cuid[:, 0]])
for k in sorted(set(nb.tolist())):
    print("blocks on a CU holding %d blocks: n %d  dur mean %.1f min %.1f max %.1f" % (k, (nb == k).sum(), dur[nb == k].mean(), dur[nb == k].min(), dur[nb == k].max()))
slot0 = hw[:, 0] & 0xF
for sl in sorted(set(slot0.tolist())):
    m = slot0 == sl
    print("wave slot %d: n %d  dur mean %.1f  min %.1f  max %.1f" % (sl, m.sum(), dur[m].mean(), dur[m].min(), dur[m].max()))
percu = {}
for b in range(1024):
    percu.setdefault(int(cuid[b, 0]), []).append(dur[b])
spread = np.array([max(v) - min(v) for v in percu.values()])
cumax = np.array([max(v) for v in percu.values()])
print("per CU: spread of its 4 workgroups mean %.1f max %.1f; last-finisher mean %.1f min %.1f max %.1f" % (spread.mean(), spread.max(), cumax.mean(), cumax.min(), cumax.max()))
# by XCD guess: block index mod 8
for x in range(8):
    sel = np.arange(1024) % 8 == x
    print("blocks = %d mod 8: dur mean %.1f  end max %.1f" % (x, dur[sel].mean(), en[sel].max()))
