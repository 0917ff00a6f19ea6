#!/usr/bin/env python3
"""Step-kernel time of consecutive launches from a cold start (engine level, config 2): does
the GPU need a ramp before it reaches its steady rate?   tools/ramp_probe.py [launches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
mean, cov = g["mean_d30"], g["cov_d30"]
d, W = 30, 65536
eng = Engine(d, W, group_size=256, seed=1, incremental=True, basis_group_size=1024)
eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
eng.set_target_gaussian_mixture([mean], [cov])
eng.set_proposal_cov(cov)
rng = np.random.default_rng(1)
eng.set_state(np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6))
eng.enable_timing(True)
time.sleep(2.0)     # an idle device
out = []
t0 = time.perf_counter()
for i in range(n):
    eng.kernel_times(reset=True)
    eng.step(1200)
    eng.sync()
    out.append((time.perf_counter() - t0, eng.kernel_times()["step_ms"]))
print(" ".join("%.0f:%.3f" % (1e3 * t, ms) for t, ms in out))
