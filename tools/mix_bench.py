#!/usr/bin/env python3
"""Throughput of the incremental step kernel on a K-mode mixture at d = 30 (engine level):
tools/mix_bench.py [K ...]   or   tools/mix_bench.py d:K [d:K ...]  (incremental only)
MIX_W=<walkers> (default 65536); MCMC_HIP_DUO=0 / 1 forces four / two lanes per walker."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402  (MCMC_HIP_LIB selects an experiment build)


def run(K, inc, d=30, W=int(os.environ.get("MIX_W", "65536")), gs=256, launches=4):
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    if d == 30:
        mean, cov = g["mean_d30"], g["cov_d30"]
    else:
        r0 = np.random.default_rng(d)
        A = r0.normal(size=(d, d))
        sd = 10 ** r0.uniform(-2, np.log10(0.05), size=d)
        c = A @ A.T / d + np.eye(d)
        cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(sd, sd)
        mean = np.full(d, 0.5)
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
    spl = 40 * d
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
    args = sys.argv[1:] or ["2", "4"]
    for a in args:
        if ":" in a:      # d:K -- incremental only (occupancy sweeps)
            d, K = (int(v) for v in a.split(":"))
            run(K, True, d=d, launches=2)
        else:
            for inc in (False, True):
                run(int(a), inc)
