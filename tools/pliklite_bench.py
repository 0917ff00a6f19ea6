"""Engine-level timing of the binned (plik-lite) target at the benchmark size: the three kernels
of a step (pl_walker, pl_residual, pl_chi2) by HIP events.
    python tools/pliklite_bench.py [n_lin=26] [walkers=65536] [steps=40]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cobaya_amd import engine as E  # noqa: E402
from cobaya_amd import pliklite as P  # noqa: E402
from tests.pliklite_common import sampling_problem  # noqa: E402

n_lin = int(sys.argv[1]) if len(sys.argv) > 1 else 26
W = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ds = P.synthetic_dataset(0)
target = P.BinnedGaussian.from_dataset(ds)
emu = P.synthetic_emulator(n_lin, ds.lmax)
kinds, a, b, C = sampling_problem(target, emu)
eng = E.Engine(n_lin + 1, W, group_size=256, seed=3)
eng.set_prior(kinds, a, b)
eng.set_target_binned_gaussian(target, emu, calib_index=n_lin)
eng.set_proposal_cov(C)
rng = np.random.default_rng(1)
x0 = np.concatenate((emu.theta0, [1.0])) + rng.standard_normal((W, n_lin + 1)) @ np.linalg.cholesky(C).T
eng.set_state(x0)
eng.step(10)
eng.sync()
eng.enable_timing(True)
eng.kernel_times(reset=True)
walls = []
for _ in range(4):      # (the first timed call also creates the events of the timing itself)
    t0 = time.perf_counter()
    eng.step(steps)
    t1 = time.perf_counter()
    eng.sync()
    walls.append((time.perf_counter() - t0, t1 - t0))
print("wall per call (total, host-side launch):", " ".join(f"{a * 1e3:.1f}/{b * 1e3:.1f} ms" for a, b in walls))
wall = min(a for a, _ in walls)
t = eng.binned_kernel_times()
n = target.n_bins
flops = n * (n + 1.0)          # executed by the triangular product (2 per multiply-add)
ms = t["chi2_ms"] / t["launches"][2]
steps_timed = 4 * steps
print(f"n_bins {n}  walkers {W}  steps {steps}  kernel {eng.last_step_kernel()}")
print(f"per step: walker {t['walker_ms'] / steps_timed:.4f} ms  residual {t['residual_ms'] / steps_timed:.4f} ms  "
      f"chi2 {ms:.4f} ms  wall {wall / steps * 1e3:.4f} ms")
print(f"chi2 kernel: {W * flops / ms * 1e-9:.2f} TFLOP/s executed "
      f"({W * 2.0 * n * n / ms * 1e-9:.2f} counting the reference's 2 n^2), "
      f"{W / ms * 1e3:.3e} evals/s; whole step {W * steps / wall:.3e} evals/s, "
      f"acceptance {eng.counters()['accepted'] / (W * (4 * steps + 10)):.3f}")
