#!/usr/bin/env python3
"""Benchmark of the mcmc_hip hot path: log-posterior evaluations per second (whole job).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): 30-dim single-mode
gaussian_mixture (target from info_random_gaussian_mixture(default_rng(0)), committed as
tests/golden/targets.npz), U(0,1) priors, 65 536 walkers per GPU initialised from the ref
pdf N(mu_i, sigma_i), proposal covariance = target covariance, proposal_scale 2.4, T = 1,
seed 1.  Synthetic data; inputs are resident in HBM when the timed region starts.

One bench "step" = one pass of the hot path over the whole ensemble = one call of the engine:
ONE fused launch of `steps_per_launch` = 40 d Metropolis steps for every walker (Haar-basis
generation + step kernel; `--steps-per-launch 4800` runs four such launches per call off one
set of directions: +3.8 % whole job, not the default -- R-1 is estimated from the per-call moment
snapshots) plus one moment snapshot; learn/convergence checkpoints (read-back of the sufficient
statistics, the all-reduce across ranks, R-1, proposal refresh) run inside the timed region
at the reference cadence `learn_every = 40d` accepted steps per chain (mcmc.yaml:22).
One evaluation = one Metropolis step of one walker = one Model.logposterior call of the
reference (mcmc.py:559).

For N > 1 the driver launches one process per GPU with torch.distributed.run; walkers shard
by rank (walker_offset = rank * n_walkers, weak scaling), the only collective is the
per-checkpoint all-reduce of pooled sufficient statistics (RCCL).

Before the W warmup steps the device is spun up with untimed launches of the same step for
`--spinup-ms` (default 60) milliseconds: an idle MI355X runs the same kernel 18 % slower until
it has been under load for about 35 ms (tools/ramp_probe.py), which is longer than W = 5
warmup steps of 1.3 ms.  The timed region is untouched: exactly K full steps.

Output (rank 0, stdout), every line a JSON object shorter than 4 KB:
  1. the headline -- metric, value, config, roofline, cpu_baseline -- as soon as its timed region
     and the CPU baseline are done (`"stage": "headline"`), BEFORE any variant runs;
  2. one `{"bench_variant": {...}}` line per extra measurement (each under its own try/except);
  3. the headline again as the LAST line (`"stage": "final"`, plus one number per variant).
The uncompacted record (certificates, full roofline blocks) goes to `bench_variants.json`
(under gpurun_out/ when that directory exists).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def algo_bytes_per_eval(d):
    """x read + written (16 d), logpost r/w (16), weight r/w (8), state persisted
    every step."""
    return 16 * d + 24


def algo_flops_per_eval(d):
    """Arithmetic of ONE log-posterior evaluation as the reference performs it: the triangular
    whitening y = L^-1 (t - mu) (d (d + 1) flops) + proposal axpy, deviation, chi2 and commit
    (4 d)."""
    return d * (d + 1) + 4 * d


HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: dense FP64, vector = matrix


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--walkers", type=int, default=65536, help="walkers per GPU")
    ap.add_argument("--dim", type=int, default=30)
    ap.add_argument("--group-size", type=int, default=None,
                    help="walkers per Haar-basis group (default: the sampler's choice, 256 at "
                         "the benchmark size)")
    ap.add_argument("--basis-group-size", type=int, default=None,
                    help="walkers sharing one Haar basis (default: the sampler's choice)")
    ap.add_argument("--steps-per-launch", type=int, default=None,
                    help="default: the sampler's, 40 d")
    ap.add_argument("--emit", choices=("snapshots", "chains"), default="snapshots",
                    help="chains: every accepted row is stored with its weight (the reference's "
                         "own semantics, mcmc.py:691-707) and drained to the host every launch")
    ap.add_argument("--evaluation", choices=("auto", "full", "incremental"), default="auto",
                    help="auto (the sampler's default): incremental evaluation where it applies")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra, separately labelled measurements (evaluation: full, "
                         "emit: chains, d = 100, the config-5 shape, planck_pliklite)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed launches before the W warmup steps until the device has been "
                         "under load this long (a cold MI355X runs the same kernel 18 %% slower "
                         "for its first ~35 ms: tools/ramp_probe.py); 0 = none")
    ap.add_argument("--checkpoint-on", dest="device_checkpoint", choices=("device", "reduce", "host"),
                    default=None,
                    help="device: window sums, the all-reduce (in place on the engine's stream), "
                         "R-1 and the proposal refresh on the device (`device_checkpoint: True`); "
                         "reduce: window sums + all-reduce on the device, the solve on the host "
                         "beside the next launch (`device_checkpoint: reduce`); host: all of it "
                         "from the pinned read-back beside the next launch.  Default: the "
                         "sampler's (host on one GPU, reduce for N > 1)")
    ap.add_argument("--checkpoint-lag", type=int, default=None,
                    help="launches between the request of a checkpoint and its processing on the "
                         "host (`checkpoint_lag`; default: the sampler's)")
    ap.add_argument("--attach-comm", action="store_true",
                    help="single process: attach a ONE-rank RCCL communicator, so that the "
                         "checkpoint queues the same ncclAllReduce an N-GPU job does")
    ap.add_argument("--device-checkpoint", dest="device_checkpoint", action="store_const",
                    const="device", help="= --checkpoint-on device")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cross-check-seconds", type=float, default=1.0,
                    help="after the K timed steps, time the same loop again for at least this long "
                         "and report it as `cross_check` (0 = skip)")
    ap.add_argument("--workload", choices=("gaussian_mixture", "pliklite"), default="gaussian_mixture",
                    help="pliklite: ONLY the planck_pliklite variant (613 bins, d = 27), as its "
                         "own line -- the command the profiles of pl_chi2_kernel are taken with")
    return ap.parse_args()


def target(d):
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    if f"mean_d{d}" in g:
        return g[f"mean_d{d}"], g[f"cov_d{d}"]
    rng = np.random.default_rng(d)  # secondary synthetic variant for other dimensions
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    c = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)
    return np.full(d, 0.5), c


def make_info(d, mean, cov, walkers, group_size, spl, emit="snapshots", evaluation="auto",
              normal_from=None):
    """normal_from = k: the config-5 SHAPE -- parameters k.. get normal priors N(0.5, 0.3) and the
    likelihood is `gaussian` (delta^T Sigma^-1 delta), as the d = 27 stand-in of SURVEY 8d."""
    names = [f"a__{i}" for i in range(d)]
    sig = np.sqrt(np.diag(cov))
    params = {n: {"prior": {"min": 0.0, "max": 1.0},
                  "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(sig[i])}}
              for i, n in enumerate(names)}
    like = {"gaussian_mixture": {"means": [mean], "covs": [cov], "input_params_prefix": "a_"}}
    if normal_from is not None:
        for n in names[normal_from:]:
            params[n]["prior"] = {"dist": "norm", "loc": 0.5, "scale": 0.3}
        like = {"gaussian": {"mean": mean, "cov": cov, "input_params_prefix": "a_"}}
    return {
        "likelihood": like,
        "params": params,
        "sampler": {"mcmc_hip": {
            "seed": 1, "n_walkers": walkers, "group_size": group_size,
            "steps_per_launch": spl, "covmat": cov, "covmat_params": names,
            "Rminus1_stop": 0.0,  # never declare convergence inside the benchmark
            # snapshots: nothing is stored in the timed region (max_rows 0); chains: every
            # accepted row crosses PCIe and lands in host memory, as the reference stores it
            "learn_proposal": True, "emit": emit, "evaluation": evaluation,
            "max_rows": 0 if emit == "snapshots" else 1 << 22}},
    }


def pliklite_problem(n_lin):
    """BASELINE configs[4]'s arithmetic on synthetic data (the Planck files cannot be downloaded
    here): plik-lite-shaped data set (613 bins), linear Cl(theta) of n_lin parameters + the
    calibration A_planck with the reference's prior (planck_calib.yaml), uniform boxes of +- 8
    posterior sigmas on the others, proposal covariance = Fisher estimate."""
    from cobaya_amd import pliklite as P
    ds = P.synthetic_dataset(0)
    tgt = P.BinnedGaussian.from_dataset(ds)
    emu = P.synthetic_emulator(n_lin, ds.lmax)
    C = P.fisher_covariance(tgt, emu)
    sig = np.sqrt(np.diag(C))
    params = {n: {"prior": {"min": float(emu.theta0[i] - 8 * sig[i]),
                            "max": float(emu.theta0[i] + 8 * sig[i])},
                  "ref": {"dist": "norm", "loc": float(emu.theta0[i]), "scale": float(sig[i])}}
              for i, n in enumerate(emu.names)}
    params["A_planck"] = {"prior": {"dist": "norm", "loc": 1.0, "scale": 0.0025},
                          "ref": {"dist": "norm", "loc": 1.0, "scale": 0.002}}
    return ds, tgt, emu, C, params


def make_pliklite_info(n_lin, walkers, group_size, spl):
    ds, tgt, emu, C, params = pliklite_problem(n_lin)
    return {
        "likelihood": {"plik_lite": {"class": "planck_pliklite", "dataset": ds,
                                     "cl_emulator": emu}},
        "params": params,
        "sampler": {"mcmc_hip": {
            "seed": 1, "n_walkers": walkers, "group_size": group_size, "steps_per_launch": spl,
            "covmat": C, "covmat_params": list(params), "Rminus1_stop": 0.0,
            "learn_proposal": True, "emit": "snapshots", "max_rows": 0}},
        "_certify_against": (np.concatenate((emu.theta0, [1.0])), C, 1),
    }, tgt


def cpu_baseline(d, mean, cov, group_size, seconds, incremental):
    """The CPU oracle (oracle/mcmc_oracle.c, the bit-exact port of the ensemble algorithm, in
    the same evaluation mode as the GPU run) timed on this host's cores on a bounded sample of
    the same workload."""
    from oracle import cbind as O
    threads = O.max_threads()
    T = O.proposal_transform(cov, 2.4)
    prob = O.Problem(d, [0] * d, [0.0] * d, [1.0] * d, means=mean, covs=cov, T=T,
                     group_size=group_size, seed=1, incremental=incremental)
    rng = np.random.default_rng(1)
    W = group_size * max(threads, 1) * 4
    x0 = np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6)
    st = O.State(prob, x0)
    st.run(d, n_threads=threads)  # warm the thread pool and the caches (untimed)
    steps, dt, chunk = 0, 0.0, 4 * d
    while dt < seconds:           # whole cycles until the time budget is used
        t0 = time.perf_counter()
        st.run(chunk, n_threads=threads)
        dt += time.perf_counter() - t0
        steps += chunk
    return {"value": W * steps / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{W} walkers x {steps} steps of the same d={d} workload, "
                      f"{'incremental' if incremental else 'full'} evaluation, "
                      f"{dt:.1f} s on {threads} OpenMP threads (oracle/mcmc_oracle.c)"}


def cpu_baseline_pliklite(n_lin, seconds):
    """The oracle's steps on the binned (plik-lite) target on this host's cores."""
    from oracle import cbind as O
    threads = O.max_threads()
    ds, tgt, emu, C, params = pliklite_problem(n_lin)
    kinds = np.array([0] * n_lin + [1], dtype=np.int32)
    a = np.array([params[n]["prior"]["min"] for n in emu.names] + [1.0])
    b = np.array([params[n]["prior"]["max"] for n in emu.names] + [0.0025])
    B = O.Binned(tgt.bin_table(), tgt.weights, tgt.X_data, cov=tgt.cov, theta0=emu.theta0,
                 D0=emu.D0, J=emu.J, calib=n_lin)
    prob = O.Problem(n_lin + 1, kinds, a, b, T=O.proposal_transform(C, 2.4), group_size=64, seed=1,
                     binned=B)
    rng = np.random.default_rng(1)
    W = 64 * max(threads // 16, 1) * 16
    x0 = np.concatenate((emu.theta0, [1.0])) + rng.standard_normal((W, n_lin + 1)) @ np.linalg.cholesky(C).T
    st = O.State(prob, x0)
    st.run(1, n_threads=threads)
    steps, dt = 0, 0.0
    while dt < seconds:
        t0 = time.perf_counter()
        st.run(2, n_threads=threads)
        dt += time.perf_counter() - t0
        steps += 2
    return {"value": W * steps / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{W} walkers x {steps} steps of the same {tgt.n_bins}-bin workload, "
                      f"{dt:.1f} s on {threads} OpenMP threads (oracle/mcmc_oracle.c, orc_binned)"}


PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16, one direction (what a drained row crosses once)


def pcie_roofline(rows, d, dt, kernel):
    """`emit: chains`: every accepted row (8 (d + 5) bytes) crosses PCIe once -- the roof that
    binds the reference's own product above ~1e9 evals/s (0.3 rows per evaluation at d = 30)."""
    gbs = rows * 8.0 * (d + 5) / dt / 1e9
    return {"bound": "pcie", "achieved": gbs, "peak": PCIE_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / PCIE_PEAK_GBS, "kernel": kernel,
            "note": "device-to-host row traffic of the drain over the PCIe Gen5 x16 rate"}


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # wave-instructions/s: 1024 SIMDs, one per 4 clocks, 2.4 GHz


def algo_flops_incremental(d):
    """FP64 arithmetic one INCREMENTAL evaluation executes on step_inc_kernel since round 4 (the
    log-likelihood is carried): trial x, the chain y.u, and the commits of x and y -- four fused
    multiply-adds per dimension (rounds 2-3: five -- trial y and its square instead of y.u)."""
    return 8 * d


def csrc_sha16():
    """sha256[:16] over the kernel sources (cobaya_amd/csrc/*.hip, *.h, sorted by name): lets a
    bench line say whether the committed counter passes were taken on the code that runs now."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "cobaya_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


ACCEPTANCE_GATE = (0.15, 0.5)   # a healthy Metropolis chain of this proposal (mcmc.py:545-562, 670-748)
KL_GATE = 0.07                  # the reference's own tolerance: tests/common_sampler.py:18, 152-161


def expected_moments(info):
    """Mean and covariance of the posterior an input describes, where they are known in closed
    form: a gaussian_mixture (moments of the mixture), or a `gaussian` likelihood times normal
    priors (product of Gaussians).  The uniform boxes are wide against the target (their
    truncation is ignored: < 1e-6 of the mass at the benchmark targets).  None otherwise."""
    like = info["likelihood"]
    names = list(info["params"])
    d = len(names)
    if "gaussian_mixture" in like:
        gm = like["gaussian_mixture"]
        means = np.atleast_2d(np.asarray(gm["means"], dtype=float))
        covs = np.asarray(gm["covs"], dtype=float)
        covs = covs if covs.ndim == 3 else covs[None]
        K = len(means)
        w = np.asarray(gm.get("weights") if gm.get("weights") is not None else np.full(K, 1.0 / K), dtype=float)
        w = w / w.sum()
        m = w @ means
        C = sum(w[k] * (covs[k] + np.outer(means[k], means[k])) for k in range(K)) - np.outer(m, m)
        return m, C, K
    if "gaussian" in like:
        mean = np.asarray(like["gaussian"]["mean"], dtype=float)
        P = np.linalg.inv(np.asarray(like["gaussian"]["cov"], dtype=float))
        h = P @ mean
        for i, n in enumerate(names):
            pr = info["params"][n]["prior"]
            if pr.get("dist") == "norm":
                P[i, i] += 1.0 / pr["scale"] ** 2
                h[i] += pr["loc"] / pr["scale"] ** 2
        C = np.linalg.inv(P)
        return C @ h, C, 1
    return None


def certify(info, x, accepted, evals, gate=True, approx=None):
    """What the timed region produced, beside how long it took (VERDICT r4 "Next round" 1b): the
    acceptance rate over the timed steps and the moments of the ensemble the region leaves --
    `x` [W][d], one sample per walker -- against the known posterior: max |mean error| / sigma,
    max error of the covariance in units of sigma_i sigma_j, and the Gaussian KL divergence the
    reference's sampler tests gate on (tests/common_sampler.py:152-161; cobaya/tools.py:745).
    `ok` is False when the acceptance rate leaves [0.15, 0.5] or (single Gaussian targets) the KL
    exceeds 0.07 -- a kernel that stopped accepting, or walked somewhere else, fails the run."""
    rate = accepted / max(evals, 1.0)
    exp = approx or expected_moments(info)
    # (the [0.15, 0.5] window is the single Gaussian's, whose acceptance under this proposal is
    # 0.30 whatever the dimension; mixtures and the plik-lite posterior -- proposed with a Fisher
    # estimate -- are held to "moves, and does not accept everything")
    single = exp is not None and exp[2] == 1 and approx is None
    gate_acc = ACCEPTANCE_GATE if single else (0.05, 0.8)
    out = {"acceptance_rate": rate, "accepted": int(accepted), "evaluations": float(evals),
           "acceptance_gate": list(gate_acc)}
    ok = gate_acc[0] <= rate <= gate_acc[1]
    if exp is not None and x is not None:
        m0, C0, K = exp
        sig = np.sqrt(np.diag(C0))
        m = x.mean(0)
        C = np.cov(x.T)
        dm = m - m0
        kl = 0.5 * (np.trace(np.linalg.solve(C, C0)) + dm @ np.linalg.solve(C, dm) - len(m0)
                    + np.linalg.slogdet(C)[1] - np.linalg.slogdet(C0)[1])
        pc = {"samples": int(len(x)), "of": "the ensemble at the end of the timed region (one "
              "sample per walker)", "max_mean_err_over_sigma": float(np.max(np.abs(dm) / sig)),
              "max_cov_err_over_sigma_i_sigma_j": float(np.max(np.abs(C - C0) / np.outer(sig, sig))),
              "KL": float(kl), "KL_gate": KL_GATE,
              # what a PERFECT sampler scores with this many independent samples
              "KL_sampling_floor": len(m0) * (len(m0) + 3) / (4.0 * len(x)),
              "sampling_error_of_a_mean_over_sigma": float(1.0 / np.sqrt(len(x)))}
        gated = bool(gate and K == 1 and approx is None)
        pc["gated"] = gated
        if approx is not None:
            pc["note"] = ("against the Gaussian (Fisher) approximation of a posterior that is not "
                          "Gaussian in the calibration parameter: reported, not gated")
        elif K > 1:
            pc["note"] = "moments of the mixture (KL between Gaussians of those moments: reported, not gated)"
        if gated and not kl <= KL_GATE:
            ok = False
        out["posterior_check"] = pc
    out["ok"] = bool(ok)
    return out


def measured_traffic(d, walkers, spl, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/traffic.json, written by tools/collect_evidence.py from `rocprofv3 --pmc
    FETCH_SIZE` / `WRITE_SIZE` runs of this same command).  bench.py cannot read hardware
    counters itself; the entry names the files and the commit it was measured at."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tj):
        return None, None
    with open(tj) as f:
        table = json.load(f)
    for key, t in reversed(list(table.items())):     # (the most recent measurement first)
        if (t.get("d"), t.get("walkers"), t.get("steps_per_launch")) == (d, walkers, spl) and \
                t.get("kernel", kernel).split("(")[0].strip() == kernel.split("(")[0].strip():
            return t.get("hbm_bytes_per_launch"), {
                "sq_insts_valu_per_launch": t.get("sq_insts_valu"),
                "file": "profiles/traffic.json#" + key, "pmc": t.get("pmc_file"),
                "measured_at_commit": t.get("commit"),
                # True: the kernel sources are byte for byte the ones the counters were taken on
                # (commits after the measurement touched documentation / tests only)
                "kernel_sources_unchanged_since_measurement":
                    (t.get("csrc_sha16") == csrc_sha16()) if t.get("csrc_sha16") else None,
                "note": "PMC passes of an earlier run of this command, not of this run"}
    return None, None


def run_timed(a, d, mean, cov, emit, steps, warmup, evaluation=None, info=None, cross_check_s=0.0):
    """W untimed + K timed bench steps of one sampler; returns the raw measurements.  `info`:
    an explicit input (variants); default: the d-dim gaussian_mixture workload."""
    from cobaya_amd import dist
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    size = dist.size()
    if info is None:
        # (None: the sampler's own default, 40 d)
        spl_req = a.steps_per_launch or (40 * d if emit == "chains" else None)
        info = make_info(d, mean, cov, a.walkers, a.group_size, spl_req, emit,
                         evaluation or a.evaluation)
        if a.basis_group_size and (evaluation or a.evaluation) != "full":
            info["sampler"]["mcmc_hip"]["basis_group_size"] = a.basis_group_size
    if getattr(a, "checkpoint_lag", None):
        info["sampler"]["mcmc_hip"]["checkpoint_lag"] = int(a.checkpoint_lag)
    if a.device_checkpoint is not None:   # (default: the sampler's -- reduce for N > 1)
        info["sampler"]["mcmc_hip"]["device_checkpoint"] = {"device": True, "reduce": "reduce",
                                                            "host": False}[a.device_checkpoint]
    approx = info.pop("_certify_against", None)   # (pliklite: the Fisher approximation)
    sampler = MCMCHip(info["sampler"]["mcmc_hip"], ProblemSpec.from_info(info))
    eng = sampler.engine
    spl = int(sampler.steps_per_launch)   # chains: capped by the device row buffer
    rows_kept = [0]

    if emit == "chains":   # count what the drains deliver
        store = sampler._store_rows

        def counting_store(rows, **kw):
            rows_kept[0] += len(rows)
            store(rows, **kw)
        sampler._store_rows = counting_store

    # one bench step = one pass of the sampler's own hot loop (EnsembleMCMC.advance): a fused
    # launch, the moment snapshot, emission, and -- when due -- the learn/convergence
    # checkpoint, processed while the next launch runs
    one_step = sampler.advance

    sampler._next_ckpt = sampler._checkpoint_steps()
    # device spin-up (untimed, reported as config.device_spinup_ms): the clocks of an idle
    # MI355X need ~35 ms under load to reach their steady state -- with W = 5 and K = 20 the
    # whole measurement would otherwise sit on that ramp
    # (the same number of launches on every rank: checkpoints hold a collective)
    if a.spinup_ms > 0:
        one_step()            # (the first launch also allocates the direction buffers)
        eng.sync()
        t_spin = time.perf_counter()
        one_step()
        eng.sync()
        t_one = max(time.perf_counter() - t_spin, 1e-4)
        n_spin = np.array([min(500.0, math.ceil(1e-3 * a.spinup_ms / t_one))])
        n_spin = int(round(float(dist.all_reduce_sum(n_spin)[0]) / size))
        for _ in range(n_spin):
            one_step()
        eng.sync()
    for _ in range(warmup):
        one_step()
    eng.sync()
    dist.all_reduce_sum(np.zeros(4))  # the collective path is initialised before timing
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    n_ckpt0, rows_kept[0] = sampler.i_learn, 0
    dist.barrier()
    eng.sync()
    acc0 = eng.counters()["accepted"]
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    if sampler._ckpt_pending:      # a checkpoint requested by the last launch belongs to it
        sampler._finish_checkpoint()
        sampler._after_checkpoint()
    eng.sync()
    dist.barrier()
    dt = time.perf_counter() - t0
    if size > 1:   # MAX over ranks
        dt = float(dist.all_reduce_max(np.array([dt]))[0])
    # what the region produced (outside the clock): accepted steps of all ranks, and rank 0's
    # ensemble as it stands at the end of the K timed steps
    accepted = float(dist.all_reduce_sum(np.array([float(eng.counters()["accepted"] - acc0)]))[0])
    x_end = eng.get_state()["x"]
    kt = eng.kernel_times()
    if sampler.spec.like_kind == "planck_pliklite":
        kt["binned"] = eng.binned_kernel_times()
    cross = None
    if cross_check_s > 0:
        # a second, LONG timed region of the same loop (>= cross_check_s seconds): the K-step
        # region above lasts tens of milliseconds -- too short for a sampling monitor (the
        # driver's smi samples saw an idle device in round 3); same bracketing, same clock
        n_x = int(max(steps, math.ceil(cross_check_s / max(dt / steps, 1e-6))))
        n_x = int(round(float(dist.all_reduce_max(np.array([float(n_x)]))[0])))
        dist.barrier()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(n_x):
            one_step()
        if sampler._ckpt_pending:
            sampler._finish_checkpoint()
            sampler._after_checkpoint()
        eng.sync()
        dist.barrier()
        dx = time.perf_counter() - t0
        if size > 1:
            dx = float(dist.all_reduce_max(np.array([dx]))[0])
        cross = {"steps": n_x, "seconds": dx, "ms_per_step": 1e3 * dx / n_x,
                 "value": float(a.walkers) * size * spl * n_x / dx}
    res = {"dt": dt, "spl": spl, "kt": kt, "kernel": eng.last_step_kernel(),
           "n_modes": int(getattr(sampler.spec, "n_modes", 1) or 1),
           "evaluation": "incremental" if sampler.incremental else "full",
           "group_size": int(sampler.group_size),
           "basis_group_size": int(sampler.basis_group_size), "n_ckpt": sampler.i_learn - n_ckpt0,
           "checkpoint_lag": int(sampler.checkpoint_lag),
           "checkpoint_on": ("host" if not sampler._device_ckpt else
                             "device" if sampler._ckpt_solve_on_device else
                             "device (window sums + all-reduce), host (solve)"),
           "rows": rows_kept[0], "evals": float(a.walkers) * size * spl * steps,
           "rows_in_store": int(sampler._n_rows), "drain_slots": int(getattr(eng, "drain_slots", 0)),
           "rows_copied_on_host": int(sum(len(r) for r in sampler._rows if not sampler._is_slot_view(r))),
           "cross_check": cross}
    res["certificate"] = certify(info, x_end, accepted, res["evals"],
                                 approx=approx)
    sampler.close()
    return res


def gaussian_roofline(m, d, walkers, steps):
    """The roofline block of a Gaussian(-mixture) workload from the raw measurements of
    `run_timed` (HIP-event kernel time of THIS run; counters from the committed PMC passes)."""
    spl, kt, dt = m["spl"], m["kt"], m["dt"]
    # one bench step = one engine.step(spl) call; the engine splits it into several kernel
    # launches when the directions of spl steps exceed its 256 MiB buffer (d = 100)
    launches_per_step = kt["step_launches"] / max(steps, 1)
    step_ms = kt["step_ms"] / max(kt["step_launches"], 1)      # per KERNEL launch
    evals_per_launch = walkers * spl / max(launches_per_step, 1)
    flops = algo_flops_per_eval(d) * evals_per_launch
    algo_bytes = algo_bytes_per_eval(d) * evals_per_launch
    tflops = flops / (step_ms * 1e-3) / 1e12 if step_ms > 0 else None
    algo_gbs = algo_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else None
    kernel = m["kernel"]
    on_matrix_cores = "mfma" in kernel
    # (the counter passes are per KERNEL launch: 40 d steps each, however many a call holds)
    traffic, traffic_source = measured_traffic(
        d, walkers, int(round(spl / max(launches_per_step, 1))), kernel)
    overlapped = m["evaluation"] == "incremental" and not os.environ.get("MCMC_HIP_NO_PREFETCH")
    common = {
        "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel,
        "kernel_ms_per_launch": step_ms, "kernel_launches_per_step": launches_per_step,
        "evals_per_kernel_launch": evals_per_launch,
        # SURVEY 8d's HBM figure, kept for reference: what the state would move if it were
        # persisted every step.  It is NOT a bandwidth the kernel achieves (x_peak may
        # exceed 1): compare `traffic`, the bytes that really cross HBM.
        "algorithmic_hbm": {
            "bytes_per_eval": algo_bytes_per_eval(d), "bytes_per_launch": algo_bytes,
            "GBps": algo_gbs, "x_peak": algo_gbs / HBM_PEAK_GBS if algo_gbs else None,
            "measured_fraction_of_algorithmic": (traffic / algo_bytes) if traffic else None},
        # incremental evaluation: the directions of the NEXT launch are computed on a second
        # stream behind the step kernel, beside the moment snapshot and the refresh of y
        # (capi.hip, DirSet); their elapsed time is then not part of the critical path
        "basis_kernel_ms_per_launch": kt["basis_ms"] / max(steps, 1),
        "basis_on_second_stream": overlapped,
        "moments_ms_per_launch": kt["moments_ms"] / max(steps, 1),
        "host_and_checkpoint_ms_per_step": 1e3 * dt / steps - (
            kt["step_ms"] + (0.0 if overlapped else kt["basis_ms"]) + kt["moments_ms"])
        / max(steps, 1)}
    if m["evaluation"] == "incremental":
        # O(d) per step: most of the instructions are not FP64 multiply-adds (two compares
        # per dimension for the prior support, Philox, two logarithms, a square root), so
        # the roof that binds is the VALU ISSUE rate -- one wave-instruction per SIMD every
        # four clocks.  achieved = SQ_INSTS_VALU of one launch (PMC pass of this command,
        # see traffic_source) / HIP-event duration of the kernel in THIS run.
        insts = (traffic_source or {}).get("sq_insts_valu_per_launch")
        ach = insts / (step_ms * 1e-3) if insts and step_ms > 0 else None
        n_modes = max(1, int(m.get("n_modes", 1)))
        fl = algo_flops_incremental(d) if n_modes == 1 else (4 + 6 * n_modes) * d
        tf = fl * evals_per_launch / (step_ms * 1e-3) / 1e12
        if ach is None:
            # no counter pass of this kernel is committed (profiles/traffic.json): the fraction is
            # the executed FP64 arithmetic over the dense FP64 peak, measured live
            return {
                "bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / FP64_PEAK_TFLOPS, "flops_per_eval_executed": fl,
                "note": "executed FP64 flops of the incremental step (trial x, the chains over the "
                        "dimensions per mode, the commits of x and the residuals) over the dense FP64 "
                        "peak; no PMC pass of this kernel is committed, so no issue-slot fraction",
                **common}
        return {
            "bound": "valu_issue", "achieved": ach / 1e9 if ach else None,
            "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
            "frac": ach / VALU_ISSUE_PEAK if ach else None,
            # beside `frac`: the two other roofs, so that nobody has to derive them -- executed
            # FP64 arithmetic over the dense FP64 peak, and the bytes that really crossed HBM
            # (PMC FETCH_SIZE + WRITE_SIZE of one launch) over the kernel's duration and 8 TB/s
            "fp64_frac": tf / FP64_PEAK_TFLOPS,
            "hbm": {"bytes_per_launch": traffic,
                    "GBps": traffic / (step_ms * 1e-3) / 1e9 if traffic and step_ms > 0 else None,
                    "peak_GBps": HBM_PEAK_GBS,
                    "frac": traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    if traffic and step_ms > 0 else None,
                    "note": "the state lives in registers for the fused launch: HBM does not bind "
                            "(north_star's >= 60 % of the HBM roofline is unmeetable by design)"},
            "fp64": {"flops_per_eval_executed": algo_flops_incremental(d),
                     "achieved_tflops": tf, "frac_of_peak": tf / FP64_PEAK_TFLOPS,
                     "flops_per_eval_from_scratch": algo_flops_per_eval(d),
                     "equivalent_from_scratch_tflops": tflops},
            **common}
    # The fused launch keeps the walker state in registers for `spl` steps, so the roof
    # that binds is FP64 arithmetic -- vector FMA for d <= 56, the matrix cores above --
    # not HBM.  achieved = algorithmic flops of one launch / HIP-event kernel duration.
    return {
        "bound": "mfma" if on_matrix_cores else "fp64_valu",
        "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": tflops / FP64_PEAK_TFLOPS if tflops else None,
        "flops_per_eval": algo_flops_per_eval(d),
        "algorithmic_flops_per_launch": flops, **common}


def pliklite_roofline(m, n_bins, walkers, steps):
    """The binned (plik-lite) target: the dominant kernel is the triangular FP64 GEMM on the
    matrix cores, Y = L^-1 Delta (n_bins (n_bins + 1) flops per evaluation; the reference's
    Sigma^-1 delta . delta is 2 n_bins^2), fed by delta through HBM (8 n_bins B written by
    pl_residual_kernel, read once per wave of pl_chi2_kernel's workgroup)."""
    kb = m["kt"]["binned"]
    n_chi2 = max(kb["launches"][2], 1)
    chi2_ms = kb["chi2_ms"] / n_chi2
    flops = float(n_bins) * (n_bins + 1) * walkers
    tf = flops / (chi2_ms * 1e-3) / 1e12
    spl = m["spl"]
    traffic, traffic_source = measured_traffic(27, walkers, spl, m["kernel"])
    return {
        "bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": tf / FP64_PEAK_TFLOPS, "kernel": m["kernel"], "kernel_ms_per_launch": chi2_ms,
        "evals_per_kernel_launch": walkers, "flops_per_eval": n_bins * (n_bins + 1),
        "flops_per_eval_as_the_reference_counts": 2 * n_bins * n_bins,
        "achieved_counting_2n2_tflops": 2.0 * n_bins * n_bins * walkers / (chi2_ms * 1e-3) / 1e12,
        "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_hbm": {"bytes_per_eval": 16 * n_bins,
                            "note": "delta written once and read once per evaluation"},
        "other_kernels_ms_per_metropolis_step": {
            "pl_walker_kernel": kb["walker_ms"] / max(steps * spl, 1),
            "pl_residual_kernel": kb["residual_ms"] / max(steps * spl, 1)},
        "metropolis_step_ms": m["dt"] * 1e3 / max(steps * spl, 1)}


def main_pliklite(a, rank, size):
    """`--workload pliklite`: the planck_pliklite variant alone, as its own JSON line."""
    spl = a.steps_per_launch or 24
    info, tgt = make_pliklite_info(26, a.walkers, a.group_size, spl)
    m = run_timed(a, 27, None, None, "snapshots", a.steps, a.warmup, info=info)
    out = None
    if rank == 0:
        out = {
            "metric": "log-posterior evals/sec (whole node), 27-dim planck_pliklite (613 bins)",
            "value": m["evals"] / m["dt"], "unit": "evals/s", "n_gpus": size, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * m["dt"] / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (plik-lite-shaped; the Planck data is not available offline)",
            "config": {"workload": "BASELINE configs[4] arithmetic: planck_pliklite, 613 bins, "
                                   "26-parameter linear Cl(theta) + A_planck, "
                                   f"{a.walkers} walkers per GPU",
                       "d": 27, "walkers_per_gpu": a.walkers, "group_size": m["group_size"],
                       "metropolis_steps_per_launch": m["spl"], "evaluation": m["evaluation"],
                       "learn_checkpoints_in_timed_region": m["n_ckpt"]},
            "acceptance_rate": m["certificate"]["acceptance_rate"],
            "accepted": m["certificate"]["accepted"],
            "posterior_check": m["certificate"].get("posterior_check"),
            "certified": bool(m["certificate"]["ok"]),
            "roofline": pliklite_roofline(m, tgt.n_bins, a.walkers, a.steps),
            "cpu_baseline": None if a.no_cpu_baseline else cpu_baseline_pliklite(26, a.cpu_seconds)}
        line = dict(out, roofline=compact_roofline(out["roofline"]),
                    posterior_check=compact_certificate(m["certificate"]))
        line["roofline"]["other_kernels_ms_per_metropolis_step"] = \
            out["roofline"]["other_kernels_ms_per_metropolis_step"]
        line["roofline"]["metropolis_step_ms"] = out["roofline"]["metropolis_step_ms"]
        print(dumps(line), flush=True)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: spawn the N ranks here (one
    process per GPU, the environment torch.distributed.run would give them, 127.0.0.1
    rendezvous), wait, and pass rank 0's JSON line through.  Fewer visible GPUs than ranks: the
    ranks share devices and the collective is the gloo stand-in (RCCL refuses two ranks on one
    device) -- the line says so in `collective.backend`."""
    import socket
    import subprocess
    # the port stays RESERVED (bound, SO_REUSEADDR) until the ranks are spawned: another
    # process cannot take it in between; rank 0's store binds it with SO_REUSEADDR as well
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if r == 0:
            sock.close()
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    # all ranks are watched TOGETHER: a rank that dies while the others sit in a collective
    # (ncclCommInitRank waits 300 s, gloo 30 min) ends the job at once
    rc = 0
    try:
        live = list(procs)
        while live and not rc:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = rc or code
            if live and not rc:
                time.sleep(0.05)
    finally:
        for p in procs:        # a rank that failed must not leave the others in a collective
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc


LINE_LIMIT = 4096   # every JSON line this file prints is shorter (a 25.6 KB line was not parsed)


def _round(o, digits=7):
    """Floats to `digits` significant digits, recursively: the lines are read by people and by a
    parser with a bounded buffer; NaN / infinities become None (strict JSON)."""
    if isinstance(o, float):
        return float(f"{o:.{digits}g}") if math.isfinite(o) else None
    if isinstance(o, (np.floating,)):
        return _round(float(o), digits)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, dict):
        return {k: _round(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v, digits) for v in o]
    return o


def dumps(o):
    return json.dumps(_round(o), separators=(",", ":"))


ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "issue_frac", "fp64_frac", "hbm_frac",
                 "traffic", "kernel", "kernel_ms_per_launch", "kernel_launches_per_step",
                 "evals_per_kernel_launch", "basis_kernel_ms_per_launch", "basis_on_second_stream",
                 "moments_ms_per_launch", "host_and_checkpoint_ms_per_step", "flops_per_eval",
                 "flops_per_eval_executed", "kernel_ms_per_launch_step_plus_basis")


def compact_roofline(r):
    """The roofline block of a printed line: scalars only, every fraction under its own name --
    `issue_frac` (VALU wave-instructions over one per SIMD per 4 clocks: issue-slot occupancy
    against a NOMINAL rate, not a physical roof), `fp64_frac` (executed FP64 flops over the dense
    FP64 peak), `hbm_frac` (PMC bytes over 8 TB/s) -- and `frac` = the one `bound` names."""
    if not r:
        return None
    out = {k: r[k] for k in ROOFLINE_KEYS if r.get(k) is not None}
    if r.get("bound") == "valu_issue":
        out["issue_frac"] = r.get("frac")
    elif r.get("bound") in ("fp64_valu", "mfma") and "fp64_frac" not in out:
        out["fp64_frac"] = r.get("frac")
    hbm = r.get("hbm") or {}
    if hbm.get("frac") is not None:
        out["hbm_frac"] = hbm["frac"]
    elif r.get("traffic") and r.get("kernel_ms_per_launch"):
        out["hbm_frac"] = r["traffic"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    ts = r.get("traffic_source") or {}
    if ts:
        out["counters"] = {"file": ts.get("file"), "commit": ts.get("measured_at_commit"),
                           "kernel_sources_unchanged_since_measurement":
                               ts.get("kernel_sources_unchanged_since_measurement")}
    ah = r.get("algorithmic_hbm") or {}
    if ah.get("bytes_per_eval") is not None:
        out["algorithmic_bytes_per_eval"] = ah["bytes_per_eval"]
        if ah.get("x_peak") is not None:
            out["algorithmic_hbm_x_peak"] = ah["x_peak"]
    return out


def compact_certificate(c):
    if not c:
        return None
    pc = c.get("posterior_check") or {}
    return {"ok": c.get("ok"), "acceptance_rate": c.get("acceptance_rate"), "KL": pc.get("KL"),
            "KL_gated": pc.get("gated"), "max_mean_err_over_sigma": pc.get("max_mean_err_over_sigma"),
            "max_cov_err_over_sigma_i_sigma_j": pc.get("max_cov_err_over_sigma_i_sigma_j")}


def variant_line(v):
    """One variant as its own stdout line, `{"bench_variant": {...}}` (no top-level `metric`: the
    headline is the only line that carries one)."""
    keep = ("tag", "variant", "value", "unit", "ms_per_step", "steps", "warmup",
            "metropolis_steps_per_launch", "evaluation", "kernel", "kernel_ms_per_launch",
            "basis_kernel_ms_per_launch", "headline_over_this", "accepted_rows_per_s",
            "row_bytes_per_s", "rows_retained_on_host", "error")
    e = {k: v[k] for k in keep if v.get(k) is not None}
    e["certificate"] = compact_certificate(v.get("certificate"))
    e["roofline"] = compact_roofline(v.get("roofline"))
    if v.get("cpu_baseline"):
        e["cpu_baseline"] = v["cpu_baseline"]
    s = dumps({"bench_variant": e})
    if len(s) >= LINE_LIMIT:      # (a long label or error text)
        e["variant"] = str(e.get("variant", ""))[:200]
        if "error" in e:
            e["error"] = str(e["error"])[:300]
        e.pop("cpu_baseline", None)
        s = dumps({"bench_variant": e})
    return s


def headline_line(out, variants, variants_file, stage):
    """THE line (VERDICT r5 "Next round" 1): compact, < 4 KB, printed when the headline's timed
    region and the CPU baseline are done (`stage: "headline"`, before any variant runs) and
    again as the LAST line of stdout (`stage: "final"`, with a one-number summary per variant)."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                                "acceptance_rate", "accepted", "certified")}
    line["posterior_check"] = {k: (out.get("posterior_check") or {}).get(k)
                               for k in ("samples", "KL", "KL_gate", "gated", "max_mean_err_over_sigma",
                                         "max_cov_err_over_sigma_i_sigma_j")} \
        if out.get("posterior_check") else None
    x = out.get("cross_check")
    line["cross_check"] = {k: x[k] for k in ("steps", "seconds", "ms_per_step", "value")} if x else None
    line["collective"] = out.get("collective")
    line["roofline"] = compact_roofline(out["roofline"])
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = dict(cb, sample=cb["sample"][:160]) if cb else None
    line["stage"] = stage
    line["variants_file"] = variants_file
    line["variants_certified"] = all((v.get("certificate") or {}).get("ok", False) or "error" in v
                                     for v in variants) if variants else None
    line["variants"] = {v.get("tag", str(i)): (v.get("value") if "error" not in v else "error")
                        for i, v in enumerate(variants)}
    s = dumps(line)
    for drop in ("variants", "posterior_check", "cross_check"):   # (never needed so far)
        if len(s) < LINE_LIMIT:
            break
        line.pop(drop, None)
        s = dumps(line)
    if len(s) >= LINE_LIMIT and isinstance(line.get("collective"), dict):
        line["collective"] = {k: v for k, v in line["collective"].items() if not isinstance(v, (list, str))}
        s = dumps(line)
    return s


def std_entry(tag, label, v, n_v, w_v, roofline, **extra):
    """The record of one variant from the raw measurements of `run_timed`."""
    e = {"tag": tag, "variant": label, "certificate": v["certificate"],
         "value": v["evals"] / v["dt"], "unit": "evals/s", "ms_per_step": 1e3 * v["dt"] / n_v,
         "steps": n_v, "warmup": w_v, "metropolis_steps_per_launch": v["spl"],
         "evaluation": v["evaluation"], "kernel": v.get("kernel"),
         "kernel_ms_per_launch": v["kt"]["step_ms"] / max(1, v["kt"]["step_launches"]),
         "roofline": roofline}
    e.update(extra)
    return e


def variant_list(a, d, mean, cov, m):
    """The extra, separately labelled measurements of a default run: (tag, callable) pairs, each
    run under its own try/except by `main` and printed as its own line."""
    W = a.walkers
    out = []
    headline = (d, W, a.emit) == (30, 65536, "snapshots")

    def gauss(tag, label, dd, n_v, w_v, info=None, evaluation=None, emit="snapshots", **extra):
        def run():
            mm, cc = (mean, cov) if dd == d else target(dd)
            v = run_timed(a, dd, mm, cc, emit, n_v, w_v, evaluation=evaluation,
                          info=info() if callable(info) else info)
            return std_entry(tag, label, v, n_v, w_v, gaussian_roofline(v, dd, W, n_v), **extra)
        out.append((tag, run))

    if m["evaluation"] == "incremental":
        # the same workload with every trial evaluated from scratch (O(d^2) per step): the
        # round-1 path, kept as `evaluation: full`
        gauss("full", "evaluation: full (every trial evaluated from scratch)", d,
              max(a.steps // 2, 10), 4, evaluation="full", emit=a.emit)
    if not headline:
        return out

    def control():
        # the price of fidelity: the reference-faithful control -- every walker draws its OWN
        # Haar basis per cycle (proposal.py:59-69 to the letter: `shared_basis: False`) and every
        # trial is evaluated from scratch, on the un-paired variate stream (0.33 threshold at 24
        # bits, 52-bit uniforms).  Same workload, same walkers.
        info_c = make_info(d, mean, cov, W, a.group_size, 4 * d, evaluation="full")
        info_c["sampler"]["mcmc_hip"]["shared_basis"] = False
        n_v = 3
        v = run_timed(a, d, mean, cov, "snapshots", n_v, 1, info=info_c)
        # what this path computes per evaluation: the from-scratch log-posterior (d (d + 1) + 4 d
        # flops) plus its share of a private Haar basis per cycle of d steps (Householder
        # construction ~ 2 d^3, V = T R: d^3 -> 3 d^2 per step); time = step + basis kernels
        fl = algo_flops_per_eval(d) + 3 * d * d
        ms = (v["kt"]["step_ms"] + v["kt"]["basis_ms"]) / n_v
        tf = fl * W * v["spl"] / (ms * 1e-3) / 1e12
        return std_entry(
            "control", "to-the-letter control: shared_basis: False (a Haar basis per walker per "
            "cycle) + evaluation: full", v, n_v, 1,
            {"bound": "fp64_valu", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS, "achieved": tf,
             "frac": tf / FP64_PEAK_TFLOPS, "flops_per_eval": fl, "kernel": v["kernel"],
             "kernel_ms_per_launch_step_plus_basis": ms},
            basis_kernel_ms_per_launch=v["kt"]["basis_ms"] / n_v,
            headline_over_this=(m["evals"] / m["dt"]) / (v["evals"] / v["dt"]))
    out.append(("control", control))

    def chains(tag, label, n_v, w_v, retained, **opts):
        def run():
            # the reference stores EVERY accepted row (mcmc.py:691-707, collection.py:402-427);
            # same workload with those semantics: rows cross PCIe and land in host memory
            info_r = make_info(d, mean, cov, W, a.group_size, 40 * d, "chains")
            info_r["sampler"]["mcmc_hip"].update(opts)
            v = run_timed(a, d, mean, cov, "chains", n_v, w_v, info=info_r)
            return std_entry(tag, label, v, n_v, w_v, pcie_roofline(v["rows"], d, v["dt"], v["kernel"]),
                             accepted_rows_per_s=v["rows"] / v["dt"],
                             row_bytes_per_s=v["rows"] * 8 * (d + 5) / v["dt"],
                             rows_retained_on_host=retained,
                             rows_in_store_at_end=v.get("rows_in_store"), drain_slots=v.get("drain_slots"),
                             stored_rows_copied_on_host=v.get("rows_copied_on_host"))
        out.append((tag, run))

    chains("chains", "emit: chains (every accepted row drained to a pinned host ring at PCIe speed; a "
           "launch's 4.7 M rows exceed max_rows, so the host does NOT retain them here)", 40, 4, False)
    # thinned ON THE DEVICE (`emit_thin`, the rule of OneSamplePoint.add_to_collection with
    # output_thin, collection.py:1373-1383, applied where the rows are produced): a row goes out
    # per 40 units of weight -- at an acceptance rate of 0.3 one accepted row in twelve
    chains("chains_thin40", "emit: chains thinned by 40 on the device (emit_thin: 40; rows of weight "
           "sum // 40 as the reference's output_thin writes them), drained to the pinned host ring",
           40, 4, False, emit_thin=40, max_rows=0)
    # rows RETAINED (max_rows = 16.7 M rows: the last ~3 launches, then the oldest half is
    # dropped): read in place in the engine's ring of pinned drain slots, or copied out of it
    chains("chains_ring", "emit: chains, rows retained on the host (read in place in the engine's "
           "pinned drain ring, sized to outlive the max_rows window: no second host copy)", 12, 8, True,
           max_rows=1 << 24, drain_ring_bytes=1 << 34)
    chains("chains_copy", "emit: chains, rows retained on the host (copied out of the pinned ring "
           "into the sampler's own memory every launch: drain_copy: True)", 4, 1, True,
           max_rows=1 << 24, drain_copy=True)

    # BASELINE configs[3]: the 100-dim gaussian_mixture, same walkers (default path) ...
    gauss("d100", "BASELINE configs[3]: 100-dim single-mode gaussian_mixture, 65536 walkers", 100, 6, 2)

    def info_d100_bounds():
        # ... with the bounds a real model has: every parameter its own box (prior.py:733-763;
        # +-0.5 around the mode, shifted per parameter) instead of one [0, 1] for all
        m4, c4 = target(100)
        info = make_info(100, m4, c4, W, a.group_size, None)
        for i, n in enumerate(info["params"]):
            info["params"][n]["prior"] = {"min": float(-0.01 * (i % 7)), "max": float(1.0 + 0.01 * (i % 5))}
        return info
    gauss("d100_bounds", "BASELINE configs[3] with per-parameter bounds (every parameter its own "
          "box, as real models have): 100-dim, 65536 walkers", 100, 6, 2, info=info_d100_bounds)
    # ... and the path north_star names for it: every trial from scratch, the
    # dense L^-1 (t - mu) contraction on the FP64 matrix cores (step_mfma_kernel)
    gauss("d100_mfma", "BASELINE configs[3] on the matrix cores: 100-dim gaussian_mixture, evaluation: "
          "full (dense Sigma^-1 x contraction via FP64 MFMA, LDS-staged L^-1 tiles), 65536 walkers",
          100, 3, 1, evaluation="full")
    # configs[4]'s SHAPE (SURVEY 8d "Config 5"): d = 27, 6 uniform + 21 normal priors, a
    # `gaussian` likelihood with a seeded SPD covariance -- synthetic, no Planck data
    gauss("config5_shape", "BASELINE configs[4] shape: 27-dim `gaussian` likelihood, 6 uniform + 21 "
          "normal priors (synthetic stand-in), 65536 walkers", 27, 10, 3,
          info=lambda: make_info(27, *target(27), W, a.group_size, None, normal_from=6))

    sig = np.sqrt(np.diag(cov))

    def mixture_info(K):
        rng8 = np.random.default_rng(8)
        info = make_info(d, mean, cov, W, a.group_size, None)
        others = [np.clip(mean + rng8.normal(size=d) * sig, 0.05, 0.95) for _ in range(7)]
        if K == 2:   # (the second mode of the round-5 line: the eighth draw of the same stream)
            others = [np.clip(mean + rng8.normal(size=d) * sig, 0.05, 0.95)]
        info["likelihood"] = {"gaussian_mixture": {"means": [mean] + others[:K - 1], "covs": [cov] * K,
                                                   "input_params_prefix": "a_"}}
        return info
    # off the single-mode path: mixtures (gaussian_mixture.py:156-163, the metric's namesake)
    gauss("mix8", "8-mode gaussian_mixture at d = 30 (the general incremental kernels), 65536 walkers",
          d, 4, 2, info=lambda: mixture_info(8))
    gauss("mix2", "2-mode gaussian_mixture at d = 30 (step_duo_mix_kernel: two lanes per walker), 65536 walkers",
          d, 6, 2, info=lambda: mixture_info(2))
    gauss("mix3", "3-mode gaussian_mixture at d = 30 (step_duo_mix_kernel, x in LDS), 65536 walkers",
          d, 6, 2, info=lambda: mixture_info(3))
    gauss("mix4", "4-mode gaussian_mixture at d = 30 (step_inc_mix_kernel), 65536 walkers",
          d, 6, 2, info=lambda: mixture_info(4))

    def info_periodic():
        # a periodic parameter (prior.py:658-676; step_inc_kernel<.., periodic>): the first
        # parameter on an interval of +-4 sigma around the mode, so that walkers do cross the seam
        info = make_info(d, mean, cov, W, a.group_size, None)
        info["params"]["a__0"]["prior"] = {"min": float(mean[0] - 4 * sig[0]),
                                           "max": float(mean[0] + 4 * sig[0])}
        info["params"]["a__0"]["periodic"] = True
        return info
    gauss("periodic", "d = 30 with one periodic parameter (interval of +-4 sigma), 65536 walkers",
          d, 8, 2, info=info_periodic)

    def pliklite():
        # configs[4]'s ARITHMETIC: the plik-lite likelihood (planck_pliklite.py:143-155) -- 613
        # bins, chi2 = delta^T Sigma^-1 delta on the matrix cores -- with a 26-parameter linear
        # Cl(theta) + A_planck (d = 27); synthetic plik-lite-shaped data (the Planck files and a
        # Boltzmann code are not available offline).  40 warm-up calls = 960 Metropolis steps:
        # the walkers start from the DIAGONAL reference pdf and the certificate compares the
        # ensemble with the correlated posterior
        n_v, spl6 = 8, 24
        info6, tgt6 = make_pliklite_info(26, W, a.group_size, spl6)
        v = run_timed(a, 27, None, None, "snapshots", n_v, 40, info=info6)
        e = std_entry("pliklite", "BASELINE configs[4] arithmetic: planck_pliklite (613 bins, FP64-MFMA "
                      "triangular GEMM), 26-parameter linear Cl(theta) + A_planck, 65536 walkers; "
                      "synthetic plik-lite-shaped data", v, n_v, 40,
                      pliklite_roofline(v, tgt6.n_bins, W, n_v))
        e["kernel_ms_per_launch"] = e["roofline"]["kernel_ms_per_launch"]
        if not a.no_cpu_baseline:
            e["cpu_baseline"] = cpu_baseline_pliklite(26, 3.0)
        return e
    out.append(("pliklite", pliklite))
    return out


def variants_path():
    """Where the full (uncompacted) record of a run goes: gpurun_out/ when it is there (it is
    merged back from the GPU box), else the repository root (git-ignored)."""
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d if os.path.isdir(d) else ROOT, "bench_variants.json")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))
    from cobaya_amd import dist

    dist.init_from_env()
    if a.attach_comm and dist.native() is None and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        dist.init_native_comm(0, 1, dist.default_device())
    rank, size = dist.rank(), dist.size()
    if size != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={size}", file=sys.stderr)
    if a.workload == "pliklite":
        return main_pliklite(a, rank, size)
    d = a.dim
    mean, cov = target(d)
    m = run_timed(a, d, mean, cov, a.emit, a.steps, a.warmup, cross_check_s=a.cross_check_seconds)
    collective = dist.describe()
    if size > 1:
        # diagnostics of a scaling run: the step-kernel time of every rank and the cost of the
        # checkpoint's all-reduce (2 d^2 + d + 5 doubles), measured after the timed region
        per = np.zeros(size)
        per[rank] = m["kt"]["step_ms"] / max(m["kt"]["step_launches"], 1)
        dist.all_reduce_sum(per)
        wall = np.zeros(size)
        wall[rank] = m["dt"]
        dist.all_reduce_sum(wall)
        buf = np.zeros(2 * d * d + d + 5)
        dist.all_reduce_sum(buf)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce_sum(buf)
        collective = dict(collective or {})
        collective.update(per_rank_step_kernel_ms=[float(x) for x in per],
                          per_rank_timed_region_s=[float(x) for x in wall],
                          # host buffer -> pinned -> H2D -> ncclAllReduce -> D2H, synchronous
                          # (what a host-path checkpoint pays)
                          checkpoint_allreduce_us=1e6 * (time.perf_counter() - t0) / 20,
                          checkpoint_allreduce_doubles=len(buf))
        comm = dist.native()
        if comm is not None:
            # what the device checkpoint queues: ncclAllReduce in place on a stream, no copies
            # (20 back to back between two HIP events: mcmc_hip_comm_time_allreduce)
            dist.barrier()
            collective["checkpoint_allreduce_in_stream_us"] = comm.time_allreduce(len(buf), 20)
    out, variants, vfile = None, [], None
    if rank == 0:
        spl, dt = m["spl"], m["dt"]
        cert = m["certificate"]
        out = {
            "metric": "log-posterior evals/sec (whole node), %d-dim gaussian_mixture" % d,
            "value": m["evals"] / dt, "unit": "evals/s", "n_gpus": size, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[1]: 30-dim single-mode gaussian_mixture, "
                             "65536 walkers per MI355X" if (d, a.walkers) == (30, 65536)
                             else f"{d}-dim single-mode gaussian_mixture, {a.walkers} walkers "
                                  "per GPU (non-default)"),
                "d": d, "walkers_per_gpu": a.walkers, "group_size": m["group_size"],
                "basis_group_size": m["basis_group_size"], "emit": a.emit,
                "device_spinup_ms": a.spinup_ms, "evaluation": m["evaluation"],
                "metropolis_steps_per_launch": spl,
                "evals_per_step": a.walkers * size * spl,
                "learn_checkpoints_in_timed_region": m["n_ckpt"],
                "checkpoint_lag_launches": m["checkpoint_lag"],
                "checkpoint_on": m["checkpoint_on"],
                "parallelism": f"walkers sharded over {size} GPU(s); one all-reduce per "
                               "checkpoint"},
            # what the timed region produced (certify()): a run that stopped accepting or left
            # the target fails (exit code 3) instead of posting a rate
            "acceptance_rate": cert["acceptance_rate"], "accepted": cert["accepted"],
            "posterior_check": cert.get("posterior_check"),
            "certified": bool(cert["ok"]),
            "cross_check": m["cross_check"],
            "collective": collective,
            "roofline": gaussian_roofline(m, d, a.walkers, a.steps),
            "cpu_baseline": None,
        }
        if size == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(d, mean, cov, m["group_size"], a.cpu_seconds,
                                                   m["evaluation"] == "incremental")
            except Exception as e:   # noqa: BLE001  (the headline is printed whatever happens here)
                print(f"[bench] cpu_baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
        todo = variant_list(a, d, mean, cov, m) if size == 1 and not a.no_variants else []
        vfile = os.path.relpath(variants_path(), ROOT) if todo else None
        # THE line, before any variant runs: nothing below can lose it
        print(headline_line(out, [], vfile, "headline"), flush=True)
        for tag, run in todo:
            try:
                v = run()
            except Exception as e:   # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                v = {"tag": tag, "variant": tag, "error": f"{type(e).__name__}: {e}"}
            variants.append(v)
            print(variant_line(v), flush=True)
        if todo:
            try:
                with open(variants_path(), "w") as f:
                    json.dump(_round(dict(out, variants=variants), 12), f, indent=1)
            except OSError as e:
                print(f"[bench] could not write {vfile}: {e}", file=sys.stderr)
                vfile = None
        # ... and again as the LAST line of stdout
        print(headline_line(out, variants, vfile, "final"), flush=True)
    dist.barrier()
    dist.shutdown()     # (the communicator is destroyed while every rank is still there)
    if out is not None:
        for v in variants:   # reported, not fatal: only the headline's certificate gates the exit code
            c = v.get("certificate") or {}
            if "error" in v or not c.get("ok", False):
                print(f"[bench] variant NOT CERTIFIED: {v.get('tag')}: {v.get('error') or c}", file=sys.stderr)
        if not out["certified"]:
            c = m["certificate"]
            print(f"[bench] NOT CERTIFIED: headline: acceptance {c['acceptance_rate']:.3f}, "
                  f"posterior_check {c.get('posterior_check')}", file=sys.stderr)
            sys.exit(3)
    return out


if __name__ == "__main__":
    main()
