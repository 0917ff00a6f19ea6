#!/usr/bin/env python3
"""Benchmark of the mcmc_hip hot path: log-posterior evaluations per second (whole job).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): 30-dim single-mode
gaussian_mixture (target from info_random_gaussian_mixture(default_rng(0)), committed as
tests/golden/targets.npz), U(0,1) priors, 65 536 walkers per GPU initialised from the ref
pdf N(mu_i, sigma_i), proposal covariance = target covariance, proposal_scale 2.4, T = 1,
seed 1.  Synthetic data; inputs are resident in HBM when the timed region starts.

One bench "step" = one pass of the hot path over the whole ensemble = ONE fused launch of
`steps_per_launch` Metropolis steps for every walker (Haar-basis generation + step kernel)
plus one moment snapshot; learn/convergence checkpoints (read-back of the sufficient
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

Prints ONE JSON line on rank 0.
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
                    help="default 40*d (the sampler's default)")
    ap.add_argument("--emit", choices=("snapshots", "chains"), default="snapshots",
                    help="chains: every accepted row is stored with its weight (the reference's "
                         "own semantics, mcmc.py:691-707) and drained to the host every launch")
    ap.add_argument("--evaluation", choices=("auto", "full", "incremental"), default="auto",
                    help="auto (the sampler's default): incremental evaluation where it applies")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra, separately labelled measurements (emit: chains)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed launches before the W warmup steps until the device has been "
                         "under load this long (a cold MI355X runs the same kernel 18 %% slower "
                         "for its first ~35 ms: tools/ramp_probe.py); 0 = none")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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


def make_info(d, mean, cov, walkers, group_size, spl, emit="snapshots", evaluation="auto"):
    names = [f"a__{i}" for i in range(d)]
    sig = np.sqrt(np.diag(cov))
    return {
        "likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov],
                                            "input_params_prefix": "a_"}},
        "params": {n: {"prior": {"min": 0.0, "max": 1.0},
                       "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(sig[i])}}
                   for i, n in enumerate(names)},
        "sampler": {"mcmc_hip": {
            "seed": 1, "n_walkers": walkers, "group_size": group_size,
            "steps_per_launch": spl, "covmat": cov, "covmat_params": names,
            "Rminus1_stop": 0.0,  # never declare convergence inside the benchmark
            # snapshots: nothing is stored in the timed region (max_rows 0); chains: every
            # accepted row crosses PCIe and lands in host memory, as the reference stores it
            "learn_proposal": True, "emit": emit, "evaluation": evaluation,
            "max_rows": 0 if emit == "snapshots" else 1 << 22}},
    }


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


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # wave-instructions/s: 1024 SIMDs, one per 4 clocks, 2.4 GHz


def algo_flops_incremental(d):
    """FP64 arithmetic one INCREMENTAL evaluation executes: trial x, trial y, chi2, and the
    commits of x and y -- five fused multiply-adds per dimension."""
    return 10 * d


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
    for key, t in table.items():
        if (t.get("d"), t.get("walkers"), t.get("steps_per_launch")) == (d, walkers, spl) and \
                t.get("kernel", kernel).split("(")[0].strip() == kernel.split("(")[0].strip():
            return t.get("hbm_bytes_per_launch"), {
                "sq_insts_valu_per_launch": t.get("sq_insts_valu"),
                "file": "profiles/traffic.json#" + key, "pmc": t.get("pmc_file"),
                "measured_at_commit": t.get("commit"),
                "note": "PMC passes of an earlier run of this command, not of this run"}
    return None, None


def run_timed(a, d, mean, cov, emit, steps, warmup, evaluation=None):
    """W untimed + K timed bench steps of one sampler; returns the raw measurements."""
    from cobaya_amd import dist
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    size = dist.size()
    spl_req = a.steps_per_launch or 40 * d
    info = make_info(d, mean, cov, a.walkers, a.group_size, spl_req, emit,
                     evaluation or a.evaluation)
    if a.basis_group_size and (evaluation or a.evaluation) != "full":
        info["sampler"]["mcmc_hip"]["basis_group_size"] = a.basis_group_size
    sampler = MCMCHip(info["sampler"]["mcmc_hip"], ProblemSpec.from_info(info))
    eng = sampler.engine
    spl = int(sampler.steps_per_launch)   # chains: capped by the device row buffer
    rows_kept = [0]

    if emit == "chains":   # count what the drains deliver
        store = sampler._store_rows

        def counting_store(rows):
            rows_kept[0] += len(rows)
            store(rows)
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
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    if sampler._ckpt_pending:      # a checkpoint requested by the last launch belongs to it
        sampler._finish_checkpoint()
    eng.sync()
    dist.barrier()
    dt = time.perf_counter() - t0
    if size > 1:   # MAX over ranks
        import torch
        import torch.distributed as td
        t = torch.tensor([dt], dtype=torch.float64)
        if td.get_backend() == "nccl":
            t = t.cuda(dist.local_rank())
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.cpu()[0])
    kt = eng.kernel_times()
    res = {"dt": dt, "spl": spl, "kt": kt, "kernel": eng.last_step_kernel(),
           "evaluation": "incremental" if sampler.incremental else "full",
           "group_size": int(sampler.group_size),
           "basis_group_size": int(sampler.basis_group_size), "n_ckpt": sampler.i_learn - n_ckpt0,
           "checkpoint_lag": int(sampler.checkpoint_lag),
           "rows": rows_kept[0], "evals": float(a.walkers) * size * spl * steps}
    sampler.close()
    return res


def main():
    a = parse()
    from cobaya_amd import dist

    dist.init_from_env()
    rank, size = dist.rank(), dist.size()
    if size != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={size}; launch with "
              "torch.distributed.run for N > 1", file=sys.stderr)
    d = a.dim
    mean, cov = target(d)
    m = run_timed(a, d, mean, cov, a.emit, a.steps, a.warmup)
    collective = dist.describe()
    variants = []
    if size == 1 and not a.no_variants and m["evaluation"] == "incremental":
        # the same workload with every trial evaluated from scratch (O(d^2) per step): the
        # round-1 path, kept as `evaluation: full`
        v = run_timed(a, d, mean, cov, a.emit, max(a.steps // 2, 10), 4, evaluation="full")
        n_v = max(a.steps // 2, 10)
        v_ms = v["kt"]["step_ms"] / max(v["kt"]["step_launches"], 1)
        v_launches = v["kt"]["step_launches"] / n_v
        variants.append({
            "variant": "evaluation: full (every trial evaluated from scratch)",
            "value": v["evals"] / v["dt"], "unit": "evals/s", "ms_per_step": 1e3 * v["dt"] / n_v,
            "steps": n_v, "warmup": 4, "kernel": v["kernel"], "kernel_ms_per_launch": v_ms,
            "fp64_tflops": algo_flops_per_eval(d) * a.walkers * v["spl"] / max(v_launches, 1)
            / (v_ms * 1e-3) / 1e12,
            "fp64_frac_of_peak": algo_flops_per_eval(d) * a.walkers * v["spl"] / max(v_launches, 1)
            / (v_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS})
    if a.emit == "snapshots" and size == 1 and not a.no_variants and (d, a.walkers) == (30, 65536):
        # the reference stores EVERY accepted row (mcmc.py:691-707, collection.py:402-427);
        # same workload with those semantics: rows cross PCIe and are kept on the host
        v = run_timed(a, d, mean, cov, "chains", 40, 4)
        variants.append({
            "variant": "emit: chains (every accepted row drained to the host, PCIe-inclusive)",
            "value": v["evals"] / v["dt"], "unit": "evals/s",
            "ms_per_step": 1e3 * v["dt"] / 40, "steps": 40, "warmup": 4,
            "metropolis_steps_per_launch": v["spl"], "kernel": v["kernel"],
            "kernel_ms_per_launch": v["kt"]["step_ms"] / 40,
            "accepted_rows_per_s": v["rows"] / v["dt"],
            "row_bytes_per_s": v["rows"] * 8 * (d + 5) / v["dt"]})
    out = None
    if rank == 0:
        spl, kt, dt = m["spl"], m["kt"], m["dt"]
        # one bench step = one engine.step(spl) call; the engine splits it into several kernel
        # launches when the directions of spl steps exceed its 256 MiB buffer (d = 100)
        launches_per_step = kt["step_launches"] / max(a.steps, 1)
        step_ms = kt["step_ms"] / max(kt["step_launches"], 1)      # per KERNEL launch
        evals_per_launch = a.walkers * spl / max(launches_per_step, 1)
        flops = algo_flops_per_eval(d) * evals_per_launch
        algo_bytes = algo_bytes_per_eval(d) * evals_per_launch
        tflops = flops / (step_ms * 1e-3) / 1e12 if step_ms > 0 else None
        algo_gbs = algo_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else None
        kernel = m["kernel"]
        on_matrix_cores = "mfma" in kernel
        traffic, traffic_source = measured_traffic(d, a.walkers, spl, kernel)
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
            "basis_kernel_ms_per_launch": kt["basis_ms"] / max(a.steps, 1),
            "basis_on_second_stream": overlapped,
            "moments_ms_per_launch": kt["moments_ms"] / max(a.steps, 1),
            "host_and_checkpoint_ms_per_step": 1e3 * dt / a.steps - (
                kt["step_ms"] + (0.0 if overlapped else kt["basis_ms"]) + kt["moments_ms"])
            / max(a.steps, 1)}
        if m["evaluation"] == "incremental":
            # O(d) per step: most of the instructions are not FP64 multiply-adds (two compares
            # per dimension for the prior support, Philox, two logarithms, a square root), so
            # the roof that binds is the VALU ISSUE rate -- one wave-instruction per SIMD every
            # four clocks.  achieved = SQ_INSTS_VALU of one launch (PMC pass of this command,
            # see traffic_source) / HIP-event duration of the kernel in THIS run.
            insts = (traffic_source or {}).get("sq_insts_valu_per_launch")
            ach = insts / (step_ms * 1e-3) if insts and step_ms > 0 else None
            tf = algo_flops_incremental(d) * evals_per_launch / (step_ms * 1e-3) / 1e12
            roofline = {
                "bound": "valu_issue", "achieved": ach / 1e9 if ach else None,
                "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
                "frac": ach / VALU_ISSUE_PEAK if ach else None,
                "fp64": {"flops_per_eval_executed": algo_flops_incremental(d),
                         "achieved_tflops": tf, "frac_of_peak": tf / FP64_PEAK_TFLOPS,
                         "flops_per_eval_from_scratch": algo_flops_per_eval(d),
                         "equivalent_from_scratch_tflops": tflops},
                **common}
        else:
            # The fused launch keeps the walker state in registers for `spl` steps, so the roof
            # that binds is FP64 arithmetic -- vector FMA for d <= 56, the matrix cores above --
            # not HBM.  achieved = algorithmic flops of one launch / HIP-event kernel duration.
            roofline = {
                "bound": "mfma" if on_matrix_cores else "fp64_valu",
                "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / FP64_PEAK_TFLOPS if tflops else None,
                "flops_per_eval": algo_flops_per_eval(d),
                "algorithmic_flops_per_launch": flops, **common}
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
                "parallelism": f"walkers sharded over {size} GPU(s); one all-reduce per "
                               "checkpoint"},
            "collective": collective,
            "roofline": roofline,
            "variants": variants,
        }
        if size == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, mean, cov, m["group_size"], a.cpu_seconds,
                                               m["evaluation"] == "incremental")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
