#!/usr/bin/env python3
"""Step kernel at the config-5 SHAPE (d = 27: 6 uniform + 21 normal priors, `gaussian` likelihood),
from scratch and incrementally, 65 536 walkers -- or at any d:n_uniform given on the command line
(incremental only), e.g. `tools/config5_bench.py 64:10 84:12` for MODE 2 at large dimensions."""
import sys
sys.path.insert(0, ".")
import numpy as np
from cobaya_amd.engine import Engine


def run(d, n_uni, modes=(False, True), W=65536):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    cov = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)
    mean = np.full(d, 0.5)
    kinds = [0] * n_uni + [1] * (d - n_uni)
    a = [0.0] * n_uni + [0.5] * (d - n_uni)
    b = [1.0] * n_uni + [0.3] * (d - n_uni)
    for inc in modes:
        eng = Engine(d, W, group_size=256, seed=1, incremental=inc,
                     basis_group_size=(4096 if d <= 64 else 16384) if inc else None)
        eng.set_prior(kinds, a, b)
        eng.set_target_gaussian(mean, cov)
        eng.set_proposal_cov(cov)
        eng.set_state(np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6))
        spl = 40 * d
        eng.step(spl)
        eng.sync()
        eng.enable_timing(True)
        eng.kernel_times(reset=True)
        n = 5 if d <= 32 else 2
        for _ in range(n):
            eng.step(spl)
        eng.sync()
        kt = eng.kernel_times()
        print(f"d={d} ({n_uni} uniform + {d - n_uni} normal priors)", "incremental" if inc else "full",
              kt["step_ms"] / n, "ms per", spl, "->", W * spl * n / (kt["step_ms"] * 1e-3), "evals/s;",
              eng.last_step_kernel(), "dirs", kt["basis_ms"] / n, flush=True)
        eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for arg in sys.argv[1:]:
            d, n_uni = (int(v) for v in arg.split(":"))
            run(d, n_uni, modes=(True,))
    else:
        run(27, 6)
