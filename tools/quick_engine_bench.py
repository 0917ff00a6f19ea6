"""Developer probe: raw engine throughput (no learn checkpoints). Not the judged bench."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobaya_amd import engine as E
if os.environ.get("MCMC_HIP_LIB"):  # experiment builds (tools/exp_variants.sh)
    E.load_library(os.environ["MCMC_HIP_LIB"])

d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
W = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
gs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
spl = int(sys.argv[4]) if len(sys.argv) > 4 else 10 * d
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "targets.npz"))
if d in (30, 100):
    mean, cov = g[f"mean_d{d}"], g[f"cov_d{d}"]
else:
    rng = np.random.default_rng(d); A = rng.normal(size=(d, d)); cov = (A @ A.T / d + np.eye(d)) * 1e-3; mean = np.full(d, 0.5)
# QB_OWN=1: shared_basis False (a Haar basis per walker: the to-the-letter control)
eng = E.Engine(d, W, group_size=gs, seed=1, shared_basis=not os.environ.get("QB_OWN"))
n_norm = int(os.environ.get("QB_NORM", "0"))  # the last n_norm priors normal (config-5 shape)
sd = np.sqrt(np.diag(cov))
kinds = [0] * (d - n_norm) + [1] * n_norm
eng.set_prior(kinds, [0.0] * (d - n_norm) + list(mean[d - n_norm:] + 0.5 * sd[d - n_norm:]),
              [1.0] * (d - n_norm) + list(2.0 * sd[d - n_norm:]))
n_modes = int(os.environ.get("QB_MODES", "1"))  # K > 1: shifted copies of the target
rngm = np.random.default_rng(11)
means = [mean] + [np.clip(mean + rngm.normal(size=d) * np.sqrt(np.diag(cov)), 0.05, 0.95)
                  for _ in range(n_modes - 1)]
eng.set_target_gaussian_mixture(means, [cov] * n_modes)
evals_per_step = 1
if os.environ.get("QB_BLOCKS"):  # "n_slow,oversample_fast[,drag_steps]": two blocks, slow first
    parts = [int(v) for v in os.environ["QB_BLOCKS"].split(",")]
    n_slow, over = parts[0], parts[1]
    drag = parts[2] if len(parts) > 2 else 0
    eng.set_blocking([list(range(n_slow)), list(range(n_slow, d))], [1, over],
                     0 if drag else -1, drag)
    evals_per_step = 1 + 2 * drag
eng.set_proposal_cov(cov)
rng = np.random.default_rng(1)
eng.set_state(np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6))
eng.enable_timing(True)
eng.step(spl); eng.sync(); eng.kernel_times(reset=True)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    eng.step(spl)
    eng.accumulate_moments()
eng.sync()
dt = time.perf_counter() - t0
kt = eng.kernel_times()
ev = W * spl * n * evals_per_step
print(f"d={d} W={W} gs={gs} spl={spl}: {ev/dt:.3e} evals/s wall; step kernel {kt['step_ms']/n:.3f} ms/launch "
      f"=> {W*spl*evals_per_step/(kt['step_ms']/n*1e-3):.3e} evals/s; basis {kt['basis_ms']/n:.3f} ms; moments {kt['moments_ms']/n:.3f} ms; "
      f"acc={eng.counters()['accepted']/(W*spl*(n+1)):.3f}")
