#!/usr/bin/env python3
"""Throughput of the dragging step, full vs incremental evaluation, at the config-5 shape
(d = 27: 6 slow + 21 fast parameters with normal priors, 7 interpolation steps) and at d = 100."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine  # noqa: E402


def run(d, n_slow, n_drag, inc, normal, W=65536, gs=256, launches=4):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)
    mean = np.full(d, 0.5)
    kinds = [0] * n_slow + [1 if normal else 0] * (d - n_slow)
    a = [0.0] * n_slow + [0.5 if normal else 0.0] * (d - n_slow)
    b = [1.0] * n_slow + [0.3 if normal else 1.0] * (d - n_slow)
    eng = Engine(d, W, group_size=gs, seed=1, incremental=inc)
    eng.set_prior(kinds, a, b)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_blocking([list(range(n_slow)), list(range(n_slow, d))], [1, 2], 0, n_drag)
    eng.set_proposal_cov(cov)
    x0 = np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6)
    eng.set_state(x0)
    spl = 40 * n_slow
    eng.step(spl)
    eng.sync()
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    for _ in range(launches):
        eng.step(spl)
    eng.sync()
    kt = eng.kernel_times()
    evals = W * spl * launches * (1 + 2 * n_drag)
    acc = eng.counters()["accepted"] / (W * spl * (launches + 1))
    print(f"d={d} slow={n_slow} n_drag={n_drag} {'incremental' if inc else 'full       '}: step kernel "
          f"{kt['step_ms'] / launches:.3f} ms per {spl} dragging steps = "
          f"{evals / (kt['step_ms'] * 1e-3):.3e} evals/s; directions {kt['basis_ms'] / launches:.3f} ms; "
          f"acc {acc:.3f}; {eng.last_step_kernel()}", flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:   # d n_slow n_drag normal [d n_slow n_drag normal ...]: incremental only
        v = [int(x) for x in sys.argv[1:]]
        for k in range(0, len(v), 4):
            run(v[k], v[k + 1], v[k + 2], True, bool(v[k + 3]))
        sys.exit(0)
    for inc in (False, True):
        run(27, 6, 7, inc, True)
    run(30, 10, 4, False, False)
    run(30, 10, 4, True, False)
    run(100, 30, 4, True, False)
