#!/usr/bin/env python3
"""The to-the-letter control at engine level: shared_basis: False (a Haar basis per walker and cycle,
proposal.py:59-69) + evaluation: full at d = 30, 65 536 walkers -- basis_kernel and step kernel times
per launch of 4 d steps.   tools/basis_bench.py [d] [launches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, gs = 65536, 256
g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
if d == 30:
    mean, cov = g["mean_d30"], g["cov_d30"]
else:
    mean, cov = np.full(d, 0.5), np.eye(d) * 0.03 ** 2
eng = Engine(d, W, group_size=gs, seed=1, incremental=False, shared_basis=False)
eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
eng.set_target_gaussian_mixture([mean], [cov])
eng.set_proposal_cov(cov)
rng = np.random.default_rng(3)
eng.set_state(np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6))
spl = 4 * d
eng.step(spl)
eng.sync()
eng.enable_timing(True)
eng.kernel_times(reset=True)
for _ in range(launches):
    eng.step(spl)
eng.sync()
kt = eng.kernel_times()
print(f"d={d} W={W}: per launch of {spl} steps: basis {kt['basis_ms'] / launches:.4f} ms, step "
      f"{kt['step_ms'] / launches:.4f} ms -> {W * spl * launches / ((kt['basis_ms'] + kt['step_ms']) * 1e-3):.3e} evals/s  "
      f"{eng.last_step_kernel()}", flush=True)
eng.close()
