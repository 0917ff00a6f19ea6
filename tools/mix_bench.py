#!/usr/bin/env python3
"""Throughput of the incremental step kernel on a K-mode mixture at d = 30 (engine level):
tools/mix_bench.py [K ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402  (MCMC_HIP_LIB selects an experiment build)


def run(K, inc, d=30, W=65536, gs=256, launches=4):
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    mean, cov = g["mean_d30"], g["cov_d30"]
    rng = np.random.default_rng(11)
    means = [mean] + [np.clip(mean + rng.normal(size=d) * np.sqrt(np.diag(cov)), 0.05, 0.95)
                      for _ in range(K - 1)]
    eng = Engine(d, W, group_size=gs, seed=1, incremental=inc,
                 basis_group_size=1024 if inc else None)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    eng.set_target_gaussian_mixture(means, [cov] * K)
    eng.set_proposal_cov(cov)
    x0 = np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6)
    eng.set_state(x0)
    spl = 1200
    eng.step(spl)
    eng.sync()
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    for _ in range(launches):
        eng.step(spl)
    eng.sync()
    kt = eng.kernel_times()
    print(f"K={K} d={d} {'incremental' if inc else 'full       '}: step kernel "
          f"{kt['step_ms'] / launches:.3f} ms per {spl} steps = "
          f"{W * spl * launches / (kt['step_ms'] * 1e-3):.3e} evals/s  {eng.last_step_kernel()}",
          flush=True)
    eng.close()


if __name__ == "__main__":
    for K in [int(a) for a in sys.argv[1:]] or [2, 4]:
        for inc in (False, True):
            run(K, inc)
