#!/usr/bin/env python3
"""Quick throughput of the step kernels, full vs incremental evaluation, at the BASELINE
shapes (engine level, no sampler): [INC_ONLY=1] tools/inc_bench.py [d ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402


def target(d):
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    if f"mean_d{d}" in g:
        return g[f"mean_d{d}"], g[f"cov_d{d}"]
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    return np.full(d, 0.5), c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)


def run(d, W, inc, gs=256, launches=6):
    mean, cov = target(d)
    eng = Engine(d, W, group_size=gs, seed=1, incremental=inc)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    rng = np.random.default_rng(1)
    x0 = np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6)
    eng.set_state(x0)
    spl = 40 * d
    eng.step(spl)
    eng.sync()
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(launches):
        eng.step(spl)
    eng.sync()
    dt = time.perf_counter() - t0
    kt = eng.kernel_times()
    acc = eng.counters()["accepted"] / (W * spl * (launches + 1))
    print(f"d={d:4d} W={W} gs={gs} {'incremental' if inc else 'full       '}: "
          f"{W * spl * launches / dt:.3e} evals/s wall; step kernel {kt['step_ms'] / launches:.3f} ms "
          f"per {spl} steps ({W * spl * launches / (kt['step_ms'] * 1e-3):.3e} evals/s), "
          f"basis+whiten {kt['basis_ms'] / launches:.3f} ms, acc {acc:.3f}, "
          f"{eng.last_step_kernel()}", flush=True)
    eng.close()


if __name__ == "__main__":
    dims = [int(a) for a in sys.argv[1:]] or [30, 100]
    for d in dims:
        for inc in ((True,) if os.environ.get("INC_ONLY") else (False, True)):
            run(d, 65536, inc)
