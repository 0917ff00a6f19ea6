#!/usr/bin/env python3
"""Step-kernel rate of the incremental kernel at given dimensions (engine level, box priors):
the measurement behind the occupancy table `inc_min_waves` (run once per experiment build,
MCMC_HIP_LIB=cobaya_amd/csrc/_exp/lib_<name>.so).   [MODE=0|1|2] tools/occupancy_sweep.py d [d ...]
MODE 0: every prior uniform on one interval; 1: uniform on different intervals; 2: every third
prior normal."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402

W = 65536
MODE = int(os.environ.get("MODE", "0"))
for d in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)
    mean = np.full(d, 0.5)
    eng = Engine(d, W, group_size=256, seed=1, incremental=True, basis_group_size=1024)
    kinds, lo, hi = [0] * d, [0.0] * d, [1.0] * d
    if MODE >= 1:
        lo = [-0.01 * (i % 3) for i in range(d)]
    if MODE == 2:
        for i in range(0, d, 3):
            kinds[i], lo[i], hi[i] = 1, 0.5, 0.3
    eng.set_prior(kinds, lo, hi)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    eng.set_state(np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6))
    spl = 10 * d
    eng.step(spl)
    eng.sync()
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    for _ in range(3):
        eng.step(spl)
    eng.sync()
    ms = eng.kernel_times()["step_ms"] / 3
    print(f"d={d:4d} dq={(d + 3) // 4:3d}: {W * spl / (ms * 1e-3):.3e} evals/s  {eng.last_step_kernel()}",
          flush=True)
    eng.close()
