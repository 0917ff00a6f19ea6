#!/usr/bin/env python3
"""Throughput of the step kernels with PERIODIC parameters, full vs incremental evaluation
(engine level, 65 536 walkers): tools/periodic_bench.py [d:n_periodic ...]   (default 30:1 30:4)
The periodic intervals are +-4 sigma around the mode, so that walkers do cross the seam."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402


def run(d, n_per, inc, W=65536, gs=256, launches=4):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    sd = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(sd, sd)
    mean = np.full(d, 0.5)
    per = [int(i < n_per) for i in range(d)]
    lo = [0.5 - 4 * sd[i] if per[i] else 0.0 for i in range(d)]
    hi = [0.5 + 4 * sd[i] if per[i] else 1.0 for i in range(d)]
    eng = Engine(d, W, group_size=gs, seed=1, incremental=inc, basis_group_size=4096 if inc else None)
    eng.set_prior([0] * d, lo, hi, per)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    x0 = mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov))
    x0 = np.clip(x0, np.array(lo) + 1e-9, np.array(hi) - 1e-9)
    eng.set_state(x0)
    spl = 40 * d
    eng.step(spl)
    eng.sync()
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    for _ in range(launches):
        eng.step(spl)
    eng.sync()
    kt = eng.kernel_times()
    print(f"d={d} periodic={n_per} {'incremental' if inc else 'full       '}: step kernel "
          f"{kt['step_ms'] / launches:.3f} ms per {spl} steps = "
          f"{W * spl * launches / (kt['step_ms'] * 1e-3):.3e} evals/s  {eng.last_step_kernel()}",
          flush=True)
    eng.close()


if __name__ == "__main__":
    cases = [a.split(":") for a in sys.argv[1:]] or [("30", "1"), ("30", "4")]
    for d, n in cases:
        for inc in (False, True):
            run(int(d), int(n), inc)
