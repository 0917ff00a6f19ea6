#!/usr/bin/env python3
"""Step-kernel throughput across MODEL configurations at 65 536 walkers (engine level,
incremental evaluation): which kernel serves (d, modes, periodic parameters) and at what rate --
the sweep behind "no supported configuration falls off a cliff" (VERDICT r2, item 5).

    tools/cliff_bench.py [d:K:n_periodic ...]      (default: the list below)

The periodic intervals are +-4 sigma around the first mode, so that walkers do cross the seam."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobaya_amd.engine import Engine, incremental_supported  # noqa: E402

DEFAULT = ["30:1:0", "30:1:1", "30:1:12", "30:2:0", "30:4:0", "30:5:0", "30:8:0", "30:16:0",
           "30:2:1", "30:5:3", "64:4:0", "64:8:0", "80:2:0", "100:1:12", "100:2:0", "100:4:0",
           "100:3:3", "128:2:0", "128:4:0"]


def run(d, K, n_per, W=65536, gs=256, launches=2):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    sd = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(sd, sd)
    mean = np.full(d, 0.5)
    means = [mean] + [np.clip(mean + rng.normal(size=d) * sd, 0.05, 0.95) for _ in range(K - 1)]
    differ = n_per < 0      # d:K:-1 -- no periodic parameter, but bounds that differ (MODE 1)
    n_per = max(n_per, 0)
    per = [int(i < n_per) for i in range(d)]
    ns = float(os.environ.get("CLIFF_PER_SIGMAS", 4))   # (wide intervals: the seam is never reached)
    lo = [0.5 - ns * sd[i] if per[i] else 0.0 for i in range(d)]
    hi = [0.5 + ns * sd[i] if per[i] else 1.0 for i in range(d)]
    if differ:
        lo[0] = -0.125
    if not incremental_supported(d, K, n_per, 0, W, 4096):
        print(f"d={d} K={K} periodic={n_per}: not served incrementally", flush=True)
        return
    # (CLIFF_MAX_TRIES: timing experiments on builds whose chains do not move)
    mt = float(os.environ.get("CLIFF_MAX_TRIES", 0)) or None
    eng = Engine(d, W, group_size=gs, seed=1, incremental=True, basis_group_size=4096, max_tries=mt)
    eng.set_prior([0] * d, lo, hi, per)
    eng.set_target_gaussian_mixture(means, [cov] * K)
    eng.set_proposal_cov(cov)
    x0 = mean + rng.standard_normal((W, d)) * sd
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
    acc = eng.counters()["accepted"] / (W * spl * (launches + 1))
    print(f"d={d} K={K} periodic={n_per}: step kernel {kt['step_ms'] / launches:.3f} ms per {spl} "
          f"steps = {W * spl * launches / (kt['step_ms'] * 1e-3):.3e} evals/s, directions "
          f"{kt['basis_ms'] / launches:.3f} ms, acc {acc:.3f}  {eng.last_step_kernel()}", flush=True)
    eng.close()


if __name__ == "__main__":
    for a in (sys.argv[1:] or DEFAULT):
        d, K, n = (int(v) for v in a.split(":"))
        run(d, K, n)
