"""GPU tests of the drop-in boundary: `run(info)` with `sampler: mcmc_hip` on the inputs the
reference's own tests use (docs quickstart = BASELINE config 1; tests/common_sampler.py fixed
3-d Gaussian with a deliberately bad initial proposal), judged by the reference's own bar
KL(truth || sample) <= 0.07 (tests/common_sampler.py:18,152-161) and much tighter ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cobaya_amd import run  # noqa: E402
from cobaya_amd.sampler import LoggedError  # noqa: E402
from tests.test_host_logic import QUICK  # noqa: E402


def kl_norm(m1, S1, m2, S2):
    """KL(N1 || N2), cobaya/tools.py:732-743."""
    d = len(m1)
    S2i = np.linalg.inv(S2)
    return 0.5 * (np.trace(S2i @ S1) + (m1 - m2) @ S2i @ (m1 - m2) - d
                  + np.linalg.slogdet(S2)[1] - np.linalg.slogdet(S1)[1])


def test_quickstart_chains_mode(tmp_path):
    """BASELINE config 1 through the plugin surface, every accepted row kept with its
    integer weight (reference semantics), derived parameters, chain file."""
    info = dict(QUICK)
    info["sampler"] = {"mcmc_hip": {"seed": 3, "n_walkers": 256, "group_size": 64,
                                    "steps_per_launch": 50, "emit": "chains",
                                    "max_samples": 150000, "Rminus1_stop": 0.0,
                                    "burn_in": 20}}
    info["output"] = str(tmp_path / "chains" / "quick")
    updated, sampler = run(info)
    assert updated["sampler"]["mcmc_hip"]["proposal_scale"] == 2.4
    prod = sampler.products()
    coll, prog = prod["sample"], prod["progress"]
    df = coll.data
    assert list(df.columns) == ["weight", "minuslogpost", "a", "b", "derived_a", "derived_b",
                                "minuslogprior", "minuslogprior__0", "chi2",
                                "chi2__gaussian_mixture"]
    assert len(df) >= 100000 and sampler.n() >= 150000
    w = df["weight"].to_numpy()
    assert np.all(w == np.round(w)) and w.min() >= 1
    acc = len(df) / w.sum()
    assert 0.15 < acc < 0.6
    mean, cov = coll.mean(), coll.cov()
    tm, tc = np.array([0.2, 0.0]), np.array([[0.1, 0.05], [0.05, 0.2]])
    assert kl_norm(tm, tc, mean, cov) < 0.07  # the reference's own bar, vs the likelihood
    # the actual posterior = likelihood x N(0,1) prior on b (x the mild truncation a > -0.5)
    P = np.linalg.inv(tc) + np.diag([0.0, 1.0])
    pc = np.linalg.inv(P)
    pm = pc @ np.linalg.inv(tc) @ tm
    assert kl_norm(pm, pc, mean, cov) < 0.003
    # stored row quantities reproduce (tests/common_sampler.py:346-372)
    x = df[["a", "b"]].to_numpy()[:200]
    Linv = np.linalg.inv(np.linalg.cholesky(tc))
    np.testing.assert_allclose(df[["derived_a", "derived_b"]].to_numpy()[:200],
                               (x - tm) @ Linv.T, rtol=1e-10, atol=1e-12)
    chi2 = np.einsum("ni,ij,nj->n", x - tm, np.linalg.inv(tc), x - tm) + np.log(
        (2 * np.pi) ** 2 * np.linalg.det(tc))
    np.testing.assert_allclose(df["chi2"].to_numpy()[:200], chi2, rtol=1e-10)
    lp = -np.log(3.5) - 0.5 * np.log(2 * np.pi) - 0.5 * x[:, 1] ** 2
    np.testing.assert_allclose(-df["minuslogprior"].to_numpy()[:200], lp, rtol=1e-12)
    np.testing.assert_allclose(df["minuslogpost"].to_numpy()[:200],
                               df["minuslogprior"].to_numpy()[:200] + 0.5 * chi2, rtol=1e-10)
    assert {"N", "timestamp", "acceptance_rate", "Rminus1", "Rminus1_cl"} == set(prog.columns)
    assert len(prog) >= 2 and np.all(prog["Rminus1"].to_numpy(dtype=float) > 0)
    # chain file in the reference's text format
    back = np.loadtxt(tmp_path / "chains" / "quick.1.txt")
    assert back.shape == df.shape
    np.testing.assert_allclose(back[:, 2:4], df[["a", "b"]].to_numpy(), rtol=1e-7)
    cm = np.loadtxt(tmp_path / "chains" / "quick.covmat")
    assert cm.shape == (2, 2)
    sampler.close()


def test_fixed3_learns_from_bad_initial_proposal():
    """tests/test_mcmc.py:22-82 / common_sampler.py:24-50,78-161 of the reference: fixed 3-d
    Gaussian, initial proposal deliberately bad (3x too wide, no correlations), covariance
    learning on, run to convergence of R-1 of means; KL <= 0.07."""
    tm = np.array([-0.48591462, 0.10064559, 0.64406749])
    tc = np.array([[0.00078333, 0.00033134, -0.0002923],
                   [0.00033134, 0.00218118, -0.00170728],
                   [-0.0002923, -0.00170728, 0.00676922]])
    info = {
        "likelihood": {"gaussian_mixture": {"means": [tm], "covs": [tc],
                                            "input_params_prefix": "a_",
                                            "output_params_prefix": "", "derived": True}},
        "params": {**{f"a__{i}": {"prior": {"min": -1, "max": 1},
                                  "ref": {"dist": "norm", "loc": float(tm[i]), "scale": 0.2},
                                  "proposal": float(3 * np.sqrt(tc[i, i]))} for i in range(3)},
                   "_0": None, "_1": None, "_2": None},
        "sampler": {"mcmc_hip": {"seed": 11, "n_walkers": 1024, "group_size": 64,
                                 "steps_per_launch": "20d", "max_tries": ".inf",
                                 "burn_in": "100d", "Rminus1_stop": 0.02,
                                 "max_samples": 1e8}},
    }
    updated, sampler = run(info)
    assert sampler.converged
    learned = sampler.proposer.get_covariance()
    np.testing.assert_allclose(learned, tc, rtol=0.25, atol=2e-5)
    # keep sampling with the learned proposal and judge the snapshots
    sampler.converged = False
    sampler.Rminus1_stop = 0.0
    sampler.snapshot_every = 60
    sampler._rows.clear()
    sampler._n_rows = 0
    sampler.max_samples = sampler.n() + 2e6
    sampler.run()
    coll = sampler.products()["sample"]
    assert len(coll) >= 20 * 1024
    assert np.all(coll["weight"] == 1)
    mean, cov = coll.mean(), coll.cov()
    assert kl_norm(tm, tc, mean, cov) < 0.07
    assert kl_norm(tm, tc, mean, cov) < 0.01
    prog = sampler.progress
    assert prog["acceptance_rate"].iloc[-1] > 0.2
    sampler.close()


def test_unsupported_inputs_raise():
    info = dict(QUICK)
    info["sampler"] = {"mcmc_hip": {"drag": True, "n_walkers": 64}}
    # one likelihood = one speed: the reference refuses the split too (model.py:1402-1407)
    with pytest.raises(LoggedError, match="all parameters have the same speed"):
        run(info)
    info["sampler"] = {"mcmc": {}}
    with pytest.raises(LoggedError, match="only runs"):
        run(info)


def test_config5_standin_d27_gaussian_with_normal_priors():
    """BASELINE config 5 stand-in: 27-d `gaussian` likelihood (delta^T S^-1 delta
    form) with 6 uniform + 21 `norm` priors -- SYNTHETIC: the real plik-lite data and a
    Boltzmann code are not available.  Posterior = likelihood x normal priors, analytic."""
    d = 27
    rng = np.random.default_rng(27)
    A = rng.normal(size=(d, d))
    corr = A @ A.T / d + np.eye(d)
    s = 10 ** rng.uniform(-2, -1.3, size=d)
    cov = corr / np.sqrt(np.outer(np.diag(corr), np.diag(corr))) * np.outer(s, s)
    mean = rng.uniform(0.4, 0.6, size=d)
    names = [f"p{i}" for i in range(d)]
    params, prior_prec, prior_mu = {}, np.zeros(d), np.zeros(d)
    for i, n in enumerate(names):
        if i < 6:
            params[n] = {"prior": {"min": 0.0, "max": 1.0},
                         "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(s[i])}}
        else:
            loc, sc = float(mean[i] + 0.5 * s[i]), float(2.0 * s[i])
            params[n] = {"prior": {"dist": "norm", "loc": loc, "scale": sc},
                         "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(s[i])}}
            prior_prec[i], prior_mu[i] = 1 / sc ** 2, loc
    info = {"likelihood": {"gaussian": {"mean": mean, "cov": cov, "normalized": True}},
            "params": params,
            "sampler": {"mcmc_hip": {"seed": 5, "n_walkers": 16384, "steps_per_launch": "10d",
                                     "covmat": cov, "covmat_params": names,
                                     "Rminus1_stop": 0.0, "max_samples": 6e7,
                                     "snapshot_every": 270, "burn_in": 0}}}
    updated, sampler = run(info)
    coll = sampler.products()["sample"]
    P = np.linalg.inv(cov) + np.diag(prior_prec)
    pc = np.linalg.inv(P)
    pm = pc @ (np.linalg.inv(cov) @ mean + prior_prec * prior_mu)
    # drop the first third (walkers start from the ref pdf, not the posterior)
    n0 = len(coll) // 3
    m, c = coll.mean(first=n0), coll.cov(first=n0)
    sig = np.sqrt(np.diag(pc))
    assert len(coll) - n0 >= 10 * 16384
    assert np.max(np.abs(m - pm) / sig) < 0.04
    assert np.max(np.abs(c - pc) / np.outer(sig, sig)) < 0.05
    assert kl_norm(pm, pc, m, c) < 0.07 and kl_norm(pm, pc, m, c) < 0.01
    # stored prior/likelihood columns are consistent with the definitions
    x = coll.data[names].to_numpy()[-50:]
    lp = -6 * 0.0 + np.sum(-np.log(2.0 * s[6:]) - 0.5 * np.log(2 * np.pi)
                           - 0.5 * ((x[:, 6:] - prior_mu[6:]) * np.sqrt(prior_prec[6:])) ** 2,
                           axis=1)
    np.testing.assert_allclose(-coll["minuslogprior"].to_numpy()[-50:], lp, rtol=1e-11)
    sampler.close()


def test_config4_d100_posterior():
    """BASELINE config 4: 100-dim single-mode gaussian_mixture (golden target fixture), column-
    sweep kernels; mean/cov of the ensemble against the analytic target."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    mean, cov = g["mean_d100"], g["cov_d100"]
    d = 100
    names = [f"a__{i}" for i in range(d)]
    sig = np.sqrt(np.diag(cov))
    info = {"likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov],
                                                "input_params_prefix": "a_"}},
            "params": {n: {"prior": {"min": 0.0, "max": 1.0},
                           "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(sig[i])}}
                       for i, n in enumerate(names)},
            "sampler": {"mcmc_hip": {"seed": 9, "n_walkers": 16384, "steps_per_launch": "4d",
                                     "covmat": cov, "covmat_params": names, "Rminus1_stop": 0.0,
                                     "max_samples": 2.2e7, "snapshot_every": 800}}}
    updated, sampler = run(info)
    assert sampler.group_size == 256   # (the moments pass the group through LDS in slices)
    coll = sampler.products()["sample"]
    n0 = len(coll) // 2
    m, c = coll.mean(first=n0), coll.cov(first=n0)
    assert len(coll) - n0 >= 2 * 16384
    assert np.max(np.abs(m - mean) / sig) < 0.05
    assert np.max(np.abs(c - cov) / np.outer(sig, sig)) < 0.08
    assert float(sampler.progress["acceptance_rate"].iloc[-1]) > 0.2
    sampler.close()


def test_config4_d100_posterior_full_size_to_one_percent():
    """BASELINE config 4 at the benchmark size, the north star's bar: 100-dim golden target,
    65 536 walkers, the sampler's defaults (incremental evaluation, a Haar basis per 16 384
    walkers above d = 64); ensemble mean within 1 % of sigma and covariance within 1 % after >= 1e6 accepted
    steps, from moment snapshots (as test_posterior_moments_full_size does at d = 30)."""
    import os
    from cobaya_amd.engine import Engine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    mean, cov = g["mean_d100"], g["cov_d100"]
    d, W = 100, 65536
    eng = Engine(d, W, group_size=256, seed=4, incremental=True, basis_group_size=16384)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    rng = np.random.default_rng(2)
    x0 = mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov))
    eng.set_state(np.clip(x0, 1e-6, 1 - 1e-6))
    eng.step(40 * d)          # burn-in from the ref pdf (narrower than the posterior)
    eng.set_moment_shift(mean)
    for _ in range(160):      # (32 000 steps of 65 536 walkers: 80 ms of step kernels)
        eng.step(2 * d)
        eng.accumulate_moments()
    eng.sync()
    assert "step_inc_kernel<25" in eng.last_step_kernel()
    n, gs, S = eng.read_moments()
    N = n * W
    m = gs.sum(0) / N
    c = S / N - np.outer(m, m)
    assert eng.counters()["accepted"] >= 1e6
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(m) / sig) < 0.01
    assert np.max(np.abs((c - cov) / np.outer(sig, sig))) < 0.01


def test_bench_two_ranks_share_one_gpu():
    """The N > 1 launch path of bench.py end to end (torch.distributed.run, walker sharding by
    rank, the per-checkpoint all-reduce, MAX-over-ranks timing) with 2 ranks on this one GPU
    and the gloo backend (RCCL needs one GPU per rank)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MCMC_HIP_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "6",
           "--walkers", "16384", "--steps-per-launch", "900"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["config"]["evals_per_step"] == 2 * 16384 * 900
    assert res["config"]["learn_checkpoints_in_timed_region"] >= 1
    assert res["value"] > 1e8 and res["cpu_baseline"] is None


@pytest.mark.parametrize("lag", [1, 2])
def test_resume_continues_bit_identically(tmp_path, lag):
    """checkpoint files + resume.  One run of 2N launches equals a run of N
    launches, a checkpoint, a fresh process-like restart (`resume: True`) and N more: walkers,
    counters and the R-1 window all come back (the reference cannot do this: its RNG state is
    not saved, sampler.py:373).  `checkpoint_lag: 2` is the default of multi-process runs (a
    checkpoint is processed two launches after its request)."""
    from cobaya_amd.sampler import MCMCHip
    from cobaya_amd.model import ProblemSpec

    def make(prefix, resume, max_samples):
        info = dict(QUICK)
        opts = {"seed": 21, "n_walkers": 512, "group_size": 64, "steps_per_launch": 40,
                "max_samples": max_samples, "Rminus1_stop": 0.0, "learn_every": "20d",
                "checkpoint_lag": lag}
        return MCMCHip(opts, ProblemSpec.from_info(info), output=prefix, resume=resume)

    a = make(str(tmp_path / "a"), False, 60000)
    a.run()
    ref = a.engine.get_full_state()
    steps_total = a.n_steps_raw
    n_prog = len(a.progress)
    a.close()

    b1 = make(str(tmp_path / "b"), False, 30000)
    b1.run()
    assert b1.n_steps_raw < steps_total
    b1.close()
    for ext in (".checkpoint", ".covmat", ".progress", ".1.state.npz", ".1.txt"):
        assert (tmp_path / ("b" + ext)).exists(), ext
    b2 = make(str(tmp_path / "b"), True, 60000)
    assert b2.n_steps_raw == b1.n_steps_raw and len(b2.progress) == len(b1.progress)
    b2.run()
    got = b2.engine.get_full_state()
    assert b2.n_steps_raw == steps_total and len(b2.progress) == n_prog
    for k in ("x", "logpost", "weight", "n_accept", "burn_left", "prior_rej"):
        assert np.array_equal(got[k], ref[k]), k
    assert int(got["step"]) == int(ref["step"])
    pa = a.progress["Rminus1"].to_numpy(dtype=float)
    pb = b2.progress["Rminus1"].to_numpy(dtype=float)
    np.testing.assert_allclose(pb, pa, rtol=1e-12)
    with pytest.raises(LoggedError, match="different number of chains"):
        MCMCHip({"seed": 21, "n_walkers": 256, "group_size": 64}, ProblemSpec.from_info(QUICK),
                output=str(tmp_path / "b"), resume=True)
    b2.close()


def test_temperature_2_samples_the_tempered_posterior():
    """tests/test_mcmc.py:22-82 runs the reference at temperature 1 and 2; at T the chain
    samples p^(1/T) (mcmc.py:127-130, 438-440, 682): for a Gaussian target the covariance is
    T times larger, and the stored minuslogpost is -logpost/T (collection.py:530-532)."""
    tm = np.array([-0.48591462, 0.10064559, 0.64406749])
    tc = np.array([[0.00078333, 0.00033134, -0.0002923],
                   [0.00033134, 0.00218118, -0.00170728],
                   [-0.0002923, -0.00170728, 0.00676922]])
    names = ["a__0", "a__1", "a__2"]
    info = {"likelihood": {"gaussian_mixture": {"means": [tm], "covs": [tc],
                                                "input_params_prefix": "a_"}},
            "params": {n: {"prior": {"min": -2, "max": 2},
                           "ref": {"dist": "norm", "loc": float(tm[i]), "scale": 0.05}}
                       for i, n in enumerate(names)},
            "sampler": {"mcmc_hip": {"seed": 2, "temperature": 2, "n_walkers": 4096,
                                     "group_size": 64, "steps_per_launch": "10d",
                                     "covmat": tc, "covmat_params": names, "Rminus1_stop": 0.0,
                                     "max_samples": 4e6, "snapshot_every": 60}}}
    updated, sampler = run(info)
    np.testing.assert_allclose(sampler.proposer.get_covariance(), 2 * tc, rtol=0.3, atol=1e-5)
    coll = sampler.products()["sample"]
    n0 = len(coll) // 3
    m, c = coll.mean(first=n0, tempered=True), coll.cov(first=n0, tempered=True)
    assert kl_norm(tm, 2 * tc, m, c) < 0.01
    # detempered statistics (the default of mean/cov, collection.py:859-981) and an explicitly
    # detempered copy both recover the unit-temperature posterior
    assert kl_norm(tm, tc, coll.mean(first=n0), coll.cov(first=n0)) < 0.01
    unit = coll.copy()
    unit.reset_temperature()
    assert unit.temperature == 1 and kl_norm(tm, tc, unit.mean(first=n0), unit.cov(first=n0)) < 0.01
    row = coll.data.iloc[-1]
    logpost = -(row["minuslogprior"] + 0.5 * row["chi2"])
    assert row["minuslogpost"] == pytest.approx(-logpost / 2.0, rel=1e-12)
    sampler.close()


def _two_speed_info(sampler_opts):
    """A slow 2-d Gaussian on a_* and a 50x faster 3-d Gaussian on b_* (the shape of the
    reference's tests/common_sampler.py:192-260 speed test)."""
    rng = np.random.default_rng(17)
    A = rng.normal(size=(3, 3))
    cov_b = (A @ A.T / 3 + np.eye(3)) * 0.01
    info = {
        "likelihood": {
            "slow": {"class": "gaussian_mixture", "means": [[0.2, 0.0]],
                     "covs": [[[0.1, 0.05], [0.05, 0.2]]], "input_params_prefix": "a_",
                     "speed": 1},
            "fast": {"class": "gaussian_mixture", "means": [[0.5, 0.4, 0.6]], "covs": [cov_b],
                     "input_params_prefix": "b_", "speed": 50}},
        "params": {
            "a_0": {"prior": {"min": -3, "max": 3}, "proposal": 0.3,
                    "ref": {"dist": "norm", "loc": 0.2, "scale": 0.3}},
            "b_0": {"prior": {"min": 0, "max": 1}, "proposal": 0.1,
                    "ref": {"dist": "norm", "loc": 0.5, "scale": 0.1}},
            "a_1": {"prior": {"min": -3, "max": 3}, "proposal": 0.4,
                    "ref": {"dist": "norm", "loc": 0.0, "scale": 0.4}},
            "b_1": {"prior": {"min": 0, "max": 1}, "proposal": 0.1,
                    "ref": {"dist": "norm", "loc": 0.4, "scale": 0.1}},
            "b_2": {"prior": {"min": 0, "max": 1}, "proposal": 0.1,
                    "ref": {"dist": "norm", "loc": 0.6, "scale": 0.1}}},
        "sampler": {"mcmc_hip": sampler_opts}}
    tm = np.array([0.2, 0.5, 0.0, 0.4, 0.6])
    tc = np.zeros((5, 5))
    tc[np.ix_([0, 2], [0, 2])] = [[0.1, 0.05], [0.05, 0.2]]
    tc[np.ix_([1, 3, 4], [1, 3, 4])] = cov_b
    return info, tm, tc


def test_oversampled_blocks_from_likelihood_speeds():
    """(f)1 end to end: two likelihoods of different speed -> two parameter blocks, the fast
    one oversampled (model.py:1340-1467), thinned chain output (mcmc.py:377-389), one chi2
    column per likelihood; the posterior is recovered."""
    info, tm, tc = _two_speed_info({
        "seed": 5, "n_walkers": 512, "group_size": 64, "emit": "chains", "oversample_power": 0.4,
        "steps_per_launch": 60, "max_samples": 400000, "Rminus1_stop": 0.0, "burn_in": 10,
        "learn_proposal": True})
    updated, sampler = run(info)
    assert sampler.blocks == [["a_0", "a_1"], ["b_0", "b_1", "b_2"]]
    assert sampler.oversampling_factors == [1, 4] and not sampler.drag
    assert sampler.cycle_length == 14 and sampler.output_thin == 3
    assert updated["sampler"]["mcmc_hip"]["blocking"] == [[1, ["a_0", "a_1"]],
                                                          [4, ["b_0", "b_1", "b_2"]]]
    coll = sampler.products()["sample"]
    df = coll.data
    assert list(df.columns)[-3:] == ["chi2", "chi2__slow", "chi2__fast"]
    np.testing.assert_allclose(df["chi2__slow"] + df["chi2__fast"], df["chi2"], rtol=1e-9,
                               atol=1e-9)
    w = df["weight"].to_numpy()
    assert np.all(w == np.round(w)) and w.min() >= 1
    assert kl_norm(tm, tc, coll.mean(), coll.cov()) < 0.01
    sampler.close()


def test_dragging_from_likelihood_speeds():
    """(f)1 end to end: `drag: True` splits the blocks into slow and fast (mcmc.py:333-360)
    and runs the dragging step; the posterior is recovered (the reference's own bar is
    KL <= 0.07, tests/common_sampler.py:18)."""
    info, tm, tc = _two_speed_info({
        "seed": 6, "n_walkers": 1024, "group_size": 64, "drag": True, "oversample_power": 0.4,
        "steps_per_launch": 70, "max_samples": 1500000, "Rminus1_stop": 0.0,
        "learn_proposal": True})
    updated, sampler = run(info)
    assert sampler.drag and sampler.i_last_slow_block == 0
    assert sampler.drag_interp_steps == 6          # round(4 * 3 / 2)
    assert sampler.cycle_length == 2 and sampler.steps_per_launch == 10
    coll = sampler.products(skip_samples=0.3)["sample"]
    assert len(coll) >= 10 * 1024
    assert kl_norm(tm, tc, coll.mean(), coll.cov()) < 0.01
    c = sampler.engine.counters()
    assert 0.1 < c["accepted"] / (c["steps"] * 1024) < 0.9
    sampler.close()


def test_config2_full_size_run_converges():
    """BASELINE config 2 through the plugin surface at full size: 30-d single-mode
    gaussian_mixture, 65 536 walkers, learning on, run to the reference's default stopping
    rule (R-1 < 0.01 twice, then the bounds criterion); posterior mean and covariance within
    2 % from the few snapshots such a short run keeps (the 1 % of the north star is checked
    on 20 snapshots by test_posterior_moments_full_size), far inside the reference's KL <= 0.07."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    mean, cov = g["mean_d30"], g["cov_d30"]
    d = 30
    names = [f"a__{i}" for i in range(d)]
    sig = np.sqrt(np.diag(cov))
    info = {
        "likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov],
                                            "input_params_prefix": "a_"}},
        "params": {n: {"prior": {"min": 0.0, "max": 1.0},
                       "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(sig[i])},
                       "proposal": float(sig[i])} for i, n in enumerate(names)},
        "sampler": {"mcmc_hip": {"seed": 11, "n_walkers": 65536, "Rminus1_stop": 0.01,
                                 "Rminus1_cl_stop": 0.2,
                                 "max_samples": 2 * 10 ** 10}}}   # bounds the run to seconds
    updated, sampler = run(info)
    assert sampler.converged
    prog = sampler.products()["progress"]
    assert float(prog["Rminus1"].to_numpy(dtype=float)[-1]) < 0.01
    coll = sampler.products(skip_samples=0.3)["sample"]
    m, c = coll.mean(), coll.cov()
    # one ensemble snapshot alone pins a mean to 0.4 % of sigma (65 536 walkers); the run stops
    # after a few of them, so the largest of 30 deviations sits around 1 %
    assert np.max(np.abs(m - mean) / sig) < 0.02, (len(coll), np.max(np.abs(m - mean) / sig))
    assert np.max(np.abs(c - cov) / np.outer(sig, sig)) < 0.02
    assert kl_norm(mean, cov, m, c) < 0.01
    # the learned proposal covariance is the posterior covariance
    learned = sampler.proposer.get_covariance()
    assert np.max(np.abs(learned - cov) / np.outer(sig, sig)) < 0.02
    sampler.close()


def test_config2_with_manual_blocking_recovers_the_posterior():
    """(f)1 at the benchmark size: the 30-d target with a manual blocking (10 slow + 20 fast
    parameters, the fast block oversampled 3x) on the hot two-wave kernel; converges by the
    default rule and recovers mean and covariance."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    mean, cov = g["mean_d30"], g["cov_d30"]
    names = [f"a__{i}" for i in range(30)]
    sig = np.sqrt(np.diag(cov))
    info = {
        "likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov],
                                            "input_params_prefix": "a_"}},
        "params": {n: {"prior": {"min": 0.0, "max": 1.0},
                       "ref": {"dist": "norm", "loc": float(mean[i]), "scale": float(sig[i])},
                       "proposal": float(sig[i])} for i, n in enumerate(names)},
        "sampler": {"mcmc_hip": {"seed": 13, "n_walkers": 65536, "Rminus1_stop": 0.01,
                                 "Rminus1_cl_stop": 0.2, "max_samples": 2 * 10 ** 10,
                                 "blocking": [[1, names[20:]], [3, names[:20]]]}}}
    updated, sampler = run(info)
    assert sampler.cycle_length == 10 + 3 * 20 and sampler.output_thin == 2
    assert sampler.converged
    coll = sampler.products(skip_samples=0.3)["sample"]
    m, c = coll.mean(), coll.cov()
    assert np.max(np.abs(m - mean) / sig) < 0.02
    assert np.max(np.abs(c - cov) / np.outer(sig, sig)) < 0.02
    assert kl_norm(mean, cov, m, c) < 0.01
    sampler.close()


def test_two_mode_mixture_at_d40_chains_mode():
    """d > 32 beyond the specialised kernels, through `run(info)`: a two-mode gaussian_mixture
    with a periodic parameter and emitted chains -- incremental since round 3 (the register-plane
    kernel emits at run time; it was `step_general_kernel`).  The modes are close enough for the
    walkers to cross; mean and covariance follow the mixture's."""
    d = 40
    rng = np.random.default_rng(40)
    mu1 = np.full(d, 0.496) + 0.002 * rng.standard_normal(d)   # 1.7 sigma apart in all
    mu2 = np.full(d, 0.504) + 0.002 * rng.standard_normal(d)
    sig = 0.03
    cov = np.eye(d) * sig ** 2
    names = [f"a__{i}" for i in range(d)]
    params = {n: {"prior": {"min": 0.0, "max": 1.0},
                  "ref": {"dist": "norm", "loc": 0.5, "scale": 0.03}} for n in names}
    params[names[3]]["periodic"] = True
    info = {"likelihood": {"gaussian_mixture": {"means": [mu1, mu2], "covs": [cov, cov],
                                                "weights": [0.5, 0.5],
                                                "input_params_prefix": "a_"}},
            "params": params,
            "sampler": {"mcmc_hip": {"seed": 11, "n_walkers": 1024, "group_size": 64,
                                     "steps_per_launch": 80, "emit": "chains", "burn_in": 50,
                                     "max_samples": 250000, "Rminus1_stop": 0.0}}}
    updated, sampler = run(info)
    assert sampler.incremental and "step_inc_regs_kernel<10, 2, periodic, emit>" in sampler.engine.last_step_kernel()
    coll = sampler.products()["sample"]
    m, c = coll.mean(), coll.cov()
    mean = 0.5 * (mu1 + mu2)
    truth = cov + 0.25 * np.outer(mu1 - mu2, mu1 - mu2)
    assert np.max(np.abs(m - mean)) < 0.2 * sig
    assert np.max(np.abs(np.sqrt(np.diag(c)) / np.sqrt(np.diag(truth)) - 1)) < 0.1


@pytest.mark.parametrize("periodic", [False, True])
def test_six_mode_mixture_runs_incrementally(periodic):
    """More than four modes (and, second case, a periodic parameter beside them) through
    `run(info)` with `evaluation: auto`: the run is incremental -- step_inc_mix_kernel, or with
    the periodic parameter the general kernels of incremental_any.hip, not the from-scratch
    fallback -- and the pooled walkers recover the
    mean and the covariance of the mixture (modes within 1.5 sigma of each other)."""
    d, K = 12, 6
    rng = np.random.default_rng(66)
    sig = 0.03
    mus = 0.5 + 0.02 * rng.standard_normal((K, d))
    cov = np.eye(d) * sig ** 2
    w = rng.uniform(0.5, 1.5, K)
    w /= w.sum()
    names = [f"a__{i}" for i in range(d)]
    params = {n: {"prior": {"min": 0.0, "max": 1.0},
                  "ref": {"dist": "norm", "loc": 0.5, "scale": 0.03}} for n in names}
    if periodic:   # +-3.3 sigma around the middle: the seam is crossed
        params[names[2]] = {"prior": {"min": 0.4, "max": 0.6}, "periodic": True,
                            "ref": {"dist": "norm", "loc": 0.5, "scale": 0.03}}
    info = {"likelihood": {"gaussian_mixture": {"means": mus, "covs": [cov] * K, "weights": w,
                                                "input_params_prefix": "a_"}},
            "params": params,
            "sampler": {"mcmc_hip": {"seed": 5, "n_walkers": 4096, "group_size": 64,
                                     "max_samples": 3_000_000, "Rminus1_stop": 0.0}}}
    updated, sampler = run(info)
    assert sampler.incremental
    kern = sampler.engine.last_step_kernel()
    # (six modes at d = 12: step_inc_mix_kernel since round 5; beside a periodic parameter the
    # register-plane kernel of incremental_any.hip)
    want = "step_inc_regs_kernel" if periodic else "step_inc_mix_kernel<3, 6"
    assert want in kern and ("periodic" in kern) == periodic, kern
    x = sampler.engine.get_state()["x"]
    mean = w @ mus
    truth = cov + (mus - mean).T @ np.diag(w) @ (mus - mean)
    keep = [i for i in range(d) if not (periodic and i == 2)]   # (the periodic one is cut at 3.3 sigma)
    se = np.sqrt(np.diag(truth) / 4096)
    assert np.max(np.abs(x.mean(axis=0) - mean)[keep] / se[keep]) < 4.5
    assert np.max(np.abs(x.std(axis=0)[keep] / np.sqrt(np.diag(truth))[keep] - 1)) < 0.06
    if periodic:
        assert np.all((x[:, 2] >= 0.4) & (x[:, 2] <= 0.6))
    sampler.close()


def test_three_mode_mixture_at_the_baseline_ensemble_size():
    """VERDICT r3 item 9 (Tier C): a K = 3 gaussian_mixture at d = 30 with 65 536 walkers on the
    default (incremental, shared-basis, paired-stream) path: the pooled ensemble recovers the
    analytic mean and covariance of the mixture to 1 % of sigma / 1 % -- the north star's bar --
    from 12 thinned snapshots."""
    d, K = 30, 3
    rng = np.random.default_rng(303)
    A = rng.normal(size=(d, d))
    corr = A @ A.T / d + np.eye(d)
    sg = 10 ** rng.uniform(-2, np.log10(0.04), size=d)
    cov = corr / np.sqrt(np.outer(np.diag(corr), np.diag(corr))) * np.outer(sg, sg)
    L = np.linalg.cholesky(cov)
    mus = 0.5 + (L @ (0.8 * rng.standard_normal((d, K)))).T      # modes ~ 1 sigma apart: they mix
    w = np.array([0.5, 0.3, 0.2])
    names = [f"a__{i}" for i in range(d)]
    info = {"likelihood": {"gaussian_mixture": {"means": mus, "covs": [cov] * K, "weights": w,
                                                "input_params_prefix": "a_"}},
            "params": {n: {"prior": {"min": 0.0, "max": 1.0},
                           "ref": {"dist": "norm", "loc": 0.5, "scale": float(sg[i])}}
                       for i, n in enumerate(names)},
            "sampler": {"mcmc_hip": {"seed": 9, "n_walkers": 65536, "covmat": cov, "covmat_params": names,
                                     "learn_proposal": False, "Rminus1_stop": 0.0, "snapshot_every": 600,
                                     "steps_per_launch": 600, "max_samples": 65536 * 600 * 22 * 0.45,
                                     "max_rows": 1 << 21}}}
    updated, sampler = run(info)
    name = sampler.engine.last_step_kernel()   # (65 536 walkers: two lanes per walker, step_duo_mix_kernel)
    assert sampler.incremental and ("step_inc" in name or "step_duo_mix" in name)
    coll = sampler.products(skip_samples=0.4)["sample"]
    assert len(coll) >= 8 * 65536
    mean = w @ mus
    truth = cov + (mus - mean).T @ np.diag(w) @ (mus - mean)
    st = np.sqrt(np.diag(truth))
    assert np.max(np.abs(coll.mean() - mean) / st) < 0.01
    assert np.max(np.abs(coll.cov() - truth) / np.outer(st, st)) < 0.01
    sampler.close()


def test_rccl_path_with_one_rank():
    """The RCCL path executes, inside the library: `mcmc_hip_comm_*` with a world of one on
    cuda:0 (no PyTorch in the process).  Every host-path checkpoint's all-reduce goes pinned ->
    H2D -> ncclAllReduce -> D2H (`mcmc_hip_comm_allreduce`), and the whole run -- R-1 table,
    learned proposal, final ensemble -- equals the run without a communicator, bit for bit."""
    import json
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = {}
    for mode in ("nccl", "none"):
        out = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py"), mode,
                              str(port)], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
        assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
        res[mode] = json.loads(lines[-1][7:])
    coll = res["nccl"]["collective"]
    assert (coll["backend"], coll["world_size"], coll["nranks_seen"]) == ("nccl", 1, 1)
    assert coll["library"].startswith("libmcmc_hip.so (RCCL ")
    assert res["none"]["collective"]["backend"] is None
    assert res["nccl"]["allreduce_identity"]
    assert len(res["nccl"]["progress"]) >= 3
    assert res["nccl"]["progress"] == res["none"]["progress"]
    assert res["nccl"]["proposal_cov"] == res["none"]["proposal_cov"]
    assert res["nccl"]["x_sum"] == res["none"]["x_sum"]
    assert 0 < res["nccl"]["Rminus1_cl"] == res["none"]["Rminus1_cl"] < 1


def test_rccl_communicator_after_pytorch_has_been_imported():
    """The library-loading order of a real multi-rank job (`dist.init_from_env`): PyTorch is
    imported and its gloo group formed FIRST (mapping whatever RCCL ships with it), then
    libmcmc_hip.so binds RCCL at run time and creates its communicator (one rank here).  The
    in-stream all-reduce of `device_checkpoint: reduce` runs through it, and the run equals the
    one without PyTorch in the process, bit for bit."""
    import json
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in ("nccl+torch", "nccl"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py"), mode,
                              str(port), "reduce"], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
        assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
        res[mode] = json.loads(lines[-1][7:])
    a, b = res["nccl+torch"], res["nccl"]
    assert a["collective"]["backend"] == "nccl" and a["collective"]["nranks_seen"] == 1
    assert a["collective"]["bootstrap"] == "torch.distributed gloo" and b["collective"]["bootstrap"] is None
    assert a["allreduce_identity"] and len(a["progress"]) >= 3
    assert a["progress"] == b["progress"] and a["proposal_cov"] == b["proposal_cov"]
    assert a["x_sum"] == b["x_sum"] and a["Rminus1_cl"] == b["Rminus1_cl"]


def test_device_checkpoint_through_the_sampler_and_rccl():
    """Row N2 / VERDICT r2 item 7: `device_checkpoint: True` -- window sums, R-1 and the proposal
    refresh on the device, in stream order.  (i) With the library's communicator attached (RCCL,
    one rank on cuda:0) `mcmc_hip_checkpoint_begin` queues ncclAllReduce IN PLACE on the engine's
    device buffer and stream: the run equals the one without a communicator, bit for bit.
    (ii) Against the host checkpoint: the first R-1 (before any refresh can make the chains
    differ in their last bits) agrees to rounding, and both runs learn a proposal close to the
    target's covariance."""
    import json
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = {}
    for mode, dev in (("nccl", "device"), ("none", "device"), ("none", "host"), ("nccl", "reduce"),
                      ("none", "reduce")):
        out = subprocess.run([sys.executable, os.path.join(here, "_rccl_worker.py"), mode,
                              str(port), dev], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
        assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
        res[mode, dev] = json.loads(lines[-1][7:])
    a, b, h = res["nccl", "device"], res["none", "device"], res["none", "host"]
    assert (a["collective"]["backend"], a["collective"]["nranks_seen"]) == ("nccl", 1)
    assert len(a["progress"]) >= 3
    assert a["progress"] == b["progress"] and a["proposal_cov"] == b["proposal_cov"]
    assert a["x_sum"] == b["x_sum"]
    pa, ph = np.array(b["progress"]), np.array(h["progress"])
    assert pa[0, 0] == ph[0, 0] and pa[0, 1] == ph[0, 1]          # N, acceptance rate
    np.testing.assert_allclose(pa[0, 2], ph[0, 2], rtol=1e-7)     # R-1 of the first checkpoint
    ca, ch = np.array(b["proposal_cov"]), np.array(h["proposal_cov"])
    sa = np.sqrt(np.diag(ch))
    assert np.max(np.abs(ca - ch) / np.outer(sa, sa)) < 0.15
    # (iii) `device_checkpoint: reduce` (round 4: the default for N > 1): window sums, payload and
    # ncclAllReduce on the device, in stream order; the reduced payload alone comes back and the
    # HOST solves it beside the next launch
    ra, rb = res["nccl", "reduce"], res["none", "reduce"]
    assert ra["collective"]["backend"] == "nccl" and len(ra["progress"]) >= 3
    assert ra["progress"] == rb["progress"] and ra["proposal_cov"] == rb["proposal_cov"]
    assert ra["x_sum"] == rb["x_sum"]
    pr = np.array(rb["progress"])
    assert pr[0, 0] == ph[0, 0] and pr[0, 1] == ph[0, 1]
    np.testing.assert_allclose(pr[0, 2], ph[0, 2], rtol=1e-7)
    cr = np.array(rb["proposal_cov"])
    assert np.max(np.abs(cr - ch) / np.outer(sa, sa)) < 0.15
    assert 0 < ra["Rminus1_cl"] == rb["Rminus1_cl"] < 1


def test_shared_and_own_basis_sample_the_same_posterior():
    """SURVEY 7 "hard parts" / App. D `shared_basis`: the design choice -- the walkers of a
    group share one Haar basis per cycle (plus a private sign) -- against the
    reference-faithful control `shared_basis: False`, where every walker draws its own basis
    like every chain of the reference (proposal.py:59-69).  Same target (BASELINE config 2),
    same budget: posterior mean and covariance agree with the truth and with each other, the
    acceptance rates agree, and R-1 has the same size."""
    t = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden",
                                           "targets.npz"))
    mean, cov = t["mean_d30"], t["cov_d30"]
    sig = np.sqrt(np.diag(cov))
    names = [f"p{i}" for i in range(30)]
    out = {}
    for shared in (True, False):
        info = {"likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov]}},
                "params": {n: {"prior": {"min": 0, "max": 1},
                               "ref": {"dist": "norm", "loc": float(mean[i]),
                                       "scale": float(sig[i])}} for i, n in enumerate(names)},
                "sampler": {"mcmc_hip": {"seed": 5, "n_walkers": 4096, "group_size": 64,
                                         "shared_basis": shared, "steps_per_launch": "4d",
                                         "covmat": cov, "covmat_params": names,
                                         "learn_proposal": False, "Rminus1_stop": 0.0,
                                         "max_samples": 1.5e6, "snapshot_every": 120}}}
        updated, s = run(info)
        # (round 6: own bases at d <= 32 run on step_kernel<.., own basis>)
        name = s.engine.last_step_kernel()
        assert name.startswith("mcmc::step_inc_kernel") if shared else "own basis" in name
        coll = s.products(skip_samples=0.3)["sample"]
        out[shared] = (coll.mean(), coll.cov(), float(s.progress["acceptance_rate"].iloc[-1]),
                       float(s.progress["Rminus1"].iloc[-1]), len(coll))
        s.close()
    for shared in (True, False):
        m, c, acc, r, n = out[shared]
        assert n >= 20 * 4096
        assert np.max(np.abs(m - mean) / sig) < 0.03
        assert np.max(np.abs(c - cov) / np.outer(sig, sig)) < 0.05
    assert abs(out[True][2] - out[False][2]) < 0.01
    assert 0.5 < out[True][3] / out[False][3] < 2.0
    assert np.max(np.abs(out[True][0] - out[False][0]) / sig) < 0.04
