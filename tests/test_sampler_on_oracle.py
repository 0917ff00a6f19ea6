"""The standalone sampler (`cobaya_amd.MCMCHip`, `cobaya_amd.run`) end to end on the CPU, with
the ctypes seam served by the oracle-backed double (tests/oracle_engine.py): the host logic
that `-m gpu` tests exercise on the device -- life cycle, checkpoint files, resume, the R-1
semantics -- runs here at small sizes.  (The product itself has no CPU path.)"""
import os

import numpy as np
import pytest

from cobaya_amd.model import ProblemSpec
from cobaya_amd.sampler import LoggedError, MCMCHip
from tests.oracle_engine import OracleEngine
from tests.test_host_logic import QUICK


class OnOracle(MCMCHip):
    _engine_factory = staticmethod(OracleEngine)


def make(prefix, max_samples, resume=False, force=False, **opts):
    o = {"seed": 21, "n_walkers": 128, "group_size": 64, "steps_per_launch": 40,
         "max_samples": max_samples, "Rminus1_stop": 0.0, "learn_every": "20d",
         "snapshot_every": 40}
    o.update(opts)
    return OnOracle(o, ProblemSpec.from_info(QUICK), output=prefix, resume=resume, force=force)


def lines(path):
    with open(path) as f:
        return f.read().splitlines()


@pytest.mark.parametrize("lag", [1, 2])
def test_resume_keeps_rows_and_continues_bit_identically(tmp_path, lag):
    """(`checkpoint_lag: 2` is the default of multi-process runs: a checkpoint is processed
    one launch later, the resumed run must still retrace the uninterrupted one.)"""
    p = str(tmp_path / "b")
    one = make(str(tmp_path / "a"), 40000, checkpoint_lag=lag)
    one.run()
    b1 = make(p, 20000, checkpoint_lag=lag)
    b1.run()
    head = lines(p + ".1.txt")
    for ext in (".checkpoint", ".covmat", ".progress", ".1.state.npz", ".1.txt"):
        assert os.path.exists(p + ext), ext
    b2 = make(p, 40000, resume=True, checkpoint_lag=lag)
    assert b2.n_steps_raw == b1.n_steps_raw and len(b2.progress) == len(b1.progress)
    b2.run()
    full = lines(p + ".1.txt")
    assert full[:len(head)] == head and len(full) > len(head)
    assert full == lines(str(tmp_path / "a") + ".1.txt")
    ref, got = one.engine.get_full_state(), b2.engine.get_full_state()
    for k in ("x", "logpost", "weight", "n_accept", "burn_left", "prior_rej"):
        assert np.array_equal(got[k], ref[k]), k
    assert int(got["step"]) == int(ref["step"])
    np.testing.assert_allclose(b2.progress["Rminus1"].to_numpy(float),
                               one.progress["Rminus1"].to_numpy(float), rtol=1e-12)
    assert len(b2.products()["sample"]) == len(full) - 1
    with pytest.raises(LoggedError, match="different number of chains"):
        make(p, 40000, resume=True, n_walkers=256)


def test_resuming_a_finished_run_leaves_the_chain_file_alone(tmp_path):
    """`_load_checkpoint` restored a finished run, the loop was skipped and
    the chain file was rewritten header-only."""
    p = str(tmp_path / "c")
    a = make(p, 15000)
    a.run()
    before = lines(p + ".1.txt")
    stamp = os.path.getmtime(p + ".1.txt")
    assert len(before) > 100
    b = make(p, 15000, resume=True)
    b.run()
    assert lines(p + ".1.txt") == before and os.path.getmtime(p + ".1.txt") == stamp
    assert len(b.products()["sample"]) == len(before) - 1
    # a converged checkpoint is honoured unless the stop criteria change (mcmc.py:1080-1088)
    c = make(str(tmp_path / "d"), 1e9, Rminus1_stop=0.3, Rminus1_cl_stop=1.0)
    c.run()
    assert c.converged
    n = lines(str(tmp_path / "d") + ".1.txt")
    again = make(str(tmp_path / "d"), 1e9, resume=True, Rminus1_stop=0.3, Rminus1_cl_stop=1.0)
    assert again.converged
    again.run()
    assert lines(str(tmp_path / "d") + ".1.txt") == n and again.n_steps_raw == c.n_steps_raw
    tighter = make(str(tmp_path / "d"), 1e9, resume=True, Rminus1_stop=0.1, Rminus1_cl_stop=1.0)
    assert not tighter.converged
    tighter.run()
    assert tighter.converged and tighter.n_steps_raw > c.n_steps_raw
    assert lines(str(tmp_path / "d") + ".1.txt")[:len(n)] == n


def test_old_output_needs_force_or_resume(tmp_path):
    p = str(tmp_path / "e")
    make(p, 5000).run()
    with pytest.raises(LoggedError, match="force"):
        make(p, 5000)
    with pytest.raises(LoggedError, match="not both"):
        make(p, 5000, resume=True, force=True)
    s = make(p, 5000, force=True)
    assert s.n_steps_raw == 0 and not os.path.exists(p + ".1.state.npz")
    s.run()
    assert os.path.exists(p + ".1.state.npz")


def test_rows_reach_the_chain_file_during_the_run(tmp_path):
    """mcmc.py:473-481, 697-699: the chain file follows the run every `output_every`, it is
    not written only at the end (a killed run keeps its rows)."""
    p = str(tmp_path / "f")
    seen = []
    s = make(p, 30000, output_every="0s",
             callback_function=lambda smp: seen.append(
                 len(lines(p + ".1.txt")) if os.path.exists(p + ".1.txt") else 0))
    s.run()
    assert len(seen) >= 3 and seen[1] > 1 and seen[-1] > seen[1]
    assert sorted(seen) == seen


def test_rminus1_is_quoted_per_walker_and_a_shared_transient_does_not_pass(tmp_path):
    """all groups start from ONE narrow ref pdf several sigma off the mode
    and relax together, so their means agree long before anything has mixed.  Per walker (x
    group_size) the statistic stays large until every walker has drawn ~1/Rminus1_stop
    independent samples; by then the transient is long gone from the later half of the run."""
    tm, tc = np.array([0.2, 0.0]), np.array([[0.1, 0.05], [0.05, 0.2]])
    info = {"likelihood": {"gaussian_mixture": {"means": [tm], "covs": [tc]}},
            "params": {"a": {"prior": {"min": -5, "max": 5},
                             "ref": {"dist": "norm", "loc": 2.2, "scale": 0.01},
                             "proposal": 0.02},
                       "b": {"prior": {"min": -5, "max": 5},
                             "ref": {"dist": "norm", "loc": -2.5, "scale": 0.01},
                             "proposal": 0.02}}}
    s = OnOracle({"seed": 4, "n_walkers": 512, "group_size": 64, "steps_per_launch": "10d",
                  "learn_every": "10d", "max_samples": 3e6, "snapshot_every": 20},
                 ProblemSpec.from_info(info))
    s.run()
    prog = s.progress
    r = prog["Rminus1"].to_numpy(float)
    assert s.converged and len(prog) >= 6
    # the early checkpoints (walkers still travelling, proposal far too small) are NOT
    # mistaken for convergence; per group the same numbers would be 64x smaller
    assert np.nanmin(r[:3]) > 0.05 and np.nanmin(r[:3]) / 64 < 0.01
    assert r[-1] < 0.01 and r[-2] < 0.01
    coll = s.products(skip_samples=0.5)["sample"]
    m, c = coll.mean(), coll.cov()
    assert np.all(np.abs(m - tm) < 0.05 * np.sqrt(np.diag(tc)) + 0.02)
    np.testing.assert_allclose(c, tc, rtol=0.1, atol=0.004)
    np.testing.assert_allclose(s.proposer.get_covariance(), tc, rtol=0.2, atol=0.01)


def test_a_single_group_is_split_or_refused():
    """The reference's single chain is split in time for its R-1 (mcmc.py:796-813,
    Rminus1_single_split); a single group of walkers is split into that many sub-groups where each
    is still a multiple of a wavefront -- and refused where it is not."""
    s = OnOracle({"n_walkers": 256, "group_size": 256, "seed": 3, "max_samples": 4000,
                  "Rminus1_stop": 0.0, "learn_every": "20d"}, ProblemSpec.from_info(QUICK))
    assert int(s.group_size) == 64 and s.engine.G == 4
    s.run()
    assert len(s.progress) >= 1 and np.isfinite(s.progress["Rminus1"].to_numpy(float)[-1])
    with pytest.raises(LoggedError, match="at least two groups"):
        OnOracle({"n_walkers": 64, "group_size": 64}, ProblemSpec.from_info(QUICK))
    with pytest.raises(LoggedError, match="multiple of group_size"):
        OnOracle({"n_walkers": 200, "group_size": 64}, ProblemSpec.from_info(QUICK))


def test_own_basis_option_and_incremental_option_reach_the_engine():
    """`shared_basis: False` and `evaluation` are plumbed to the engine; the combinations that
    make no sense are refused with the reference's kind of error."""
    spec = ProblemSpec.from_info(QUICK)
    s = OnOracle({"n_walkers": 128, "group_size": 64, "shared_basis": False, "seed": 1,
                  "max_samples": 3000, "Rminus1_stop": 0.0}, spec)
    assert s.engine.own_basis and not s.incremental
    s.run()
    assert s.n() >= 3000
    s2 = OnOracle({"n_walkers": 128, "group_size": 64, "seed": 1, "max_samples": 3000,
                   "Rminus1_stop": 0.0}, spec)
    assert s2.incremental and s2.engine.incremental      # auto: one mode, snapshots
    s3 = OnOracle({"n_walkers": 128, "group_size": 64, "evaluation": "full", "seed": 1}, spec)
    assert not s3.incremental
    s4 = OnOracle({"n_walkers": 128, "group_size": 64, "evaluation": "incremental",
                   "emit": "chains"}, spec)      # (round 3: accepted rows on the incremental path)
    assert s4.incremental and s4.engine.cap > 0
    with pytest.raises(LoggedError, match="incremental"):
        OnOracle({"n_walkers": 128, "group_size": 64, "evaluation": "incremental",
                  "shared_basis": False}, spec)
    with pytest.raises(LoggedError, match="evaluation must be"):
        OnOracle({"n_walkers": 128, "group_size": 64, "evaluation": "sometimes"}, spec)


@pytest.mark.parametrize("walkers, expect", [(4096, 256), (16384, 1024), (65536, 4096),
                                             (65536 + 1024, 1024)])
def test_default_width_of_the_basis_groups(walkers, expect):
    """`basis_group_size` left at its default: the R-1 group below 16384 walkers per process,
    1024 walkers from there, 4096 from 65536 walkers up (when they divide the ensemble) -- and
    only with incremental evaluation, the from-scratch kernels keep a basis per R-1 group."""
    from cobaya_amd.engine import EngineError
    seen = {}

    class Recorder:
        def __init__(self, d, W, **kw):
            seen.update(kw, W=W)
            raise EngineError(-1, "recorded")

    class Rec(MCMCHip):
        _engine_factory = staticmethod(Recorder)

    for evaluation, want in (("auto", expect), ("full", 256)):
        seen.clear()
        with pytest.raises(LoggedError, match="recorded"):
            Rec({"seed": 1, "n_walkers": walkers, "group_size": 256, "evaluation": evaluation},
                ProblemSpec.from_info(QUICK), output=None)
        assert seen["W"] == walkers and seen["basis_group_size"] == want
        assert seen["incremental"] == (evaluation == "auto")


def test_checkpoint_lag_moves_the_proposal_refresh_by_one_launch(tmp_path):
    """`checkpoint_lag` launches are queued between the request of a checkpoint and the
    upload of the refreshed proposal: 1 for a single process (the refresh follows the launch
    queued after the request), 2 by default with several processes (sampler.advance)."""
    logs = {}
    for lag in (1, 2):
        log = logs[lag] = []

        class Spy(OracleEngine):
            launches = 0

            def step(self, n):
                self.launches += 1
                return super().step(n)

            def request_moments(self):
                log.append(("request", self.launches))
                return super().request_moments()

            def set_proposal_cov(self, cov):
                log.append(("refresh", self.launches))
                return super().set_proposal_cov(cov)

        class S(MCMCHip):
            _engine_factory = staticmethod(Spy)

        s = S({"seed": 21, "n_walkers": 128, "group_size": 64, "steps_per_launch": 40,
               "max_samples": 30000, "Rminus1_stop": 0.0, "learn_every": "40d",
               "checkpoint_lag": lag}, ProblemSpec.from_info(QUICK), output=None)
        s.run()
        assert s._ckpt_lag == lag and len(s.progress) >= 3
    for lag, log in logs.items():
        req = [n for what, n in log if what == "request"]
        ref = [n for what, n in log if what == "refresh"][1:]   # (the first: initialize)
        assert len(ref) >= 3 and len(req) >= len(ref)
        assert all(b - a == lag for a, b in zip(req, ref)), (lag, log)
    with pytest.raises(LoggedError, match="checkpoint_lag"):
        make(str(tmp_path / "x"), 100, checkpoint_lag=0)


def test_periodic_parameters_are_sampled_incrementally(tmp_path):
    """A periodic parameter (prior.py:658-676) does not send the run to `evaluation: full`:
    Gaussian modes with periodic parameters, without dragging, are evaluated incrementally
    (`auto`); a dragging run with a periodic parameter is not.  The
    likelihood itself is not periodic, so the posterior of the periodic parameter is the
    Gaussian cut to its interval -- reached from both ends through the seam."""
    from scipy.stats import truncnorm
    info = {"likelihood": {"gaussian_mixture": {"means": [0.02, 0.5, 0.3],
                                                "covs": np.diag([0.0064, 0.004, 0.003]).tolist()}},
            "params": {"phase": {"prior": {"min": 0, "max": 0.2}, "periodic": True},
                       "b": {"prior": {"min": 0, "max": 1}},
                       "c": {"prior": {"min": 0, "max": 1}}}}
    spec = ProblemSpec.from_info(info)
    assert spec.periodic.tolist() == [1, 0, 0]
    s = OnOracle({"n_walkers": 512, "group_size": 64, "seed": 3, "max_samples": 150000,
                  "Rminus1_stop": 0.0, "learn_every": "20d"}, spec, output=str(tmp_path / "p"))
    assert s.incremental and s.engine.incremental
    s.run()
    x = s.engine.get_state()["x"]
    assert np.all((x[:, 0] >= 0) & (x[:, 0] <= 0.2))
    assert np.sum(x[:, 0] > 0.15) > 15 and np.sum(x[:, 0] < 0.05) > 60    # both ends of the seam
    tn = truncnorm((0 - 0.02) / 0.08, (0.2 - 0.02) / 0.08, loc=0.02, scale=0.08)
    assert abs(x[:, 0].mean() - tn.mean()) < 4 * tn.std() / np.sqrt(512)
    assert abs(x[:, 1].mean() - 0.5) < 4 * np.sqrt(0.004 / 512)
    # a mixture with a periodic parameter: incremental as well (the general kernel) ...
    two = dict(info, likelihood={"gaussian_mixture": {
        "means": [[0.02, 0.5, 0.3], [0.1, 0.4, 0.3]],
        "covs": [np.diag([0.0064, 0.004, 0.003]).tolist()] * 2}})
    s2 = OnOracle({"n_walkers": 128, "group_size": 64, "seed": 3}, ProblemSpec.from_info(two))
    assert s2.incremental
    # ... what stays with the from-scratch kernels: dragging with a periodic parameter
    drag = {"n_walkers": 128, "group_size": 64, "seed": 3, "drag": True,
            "blocking": [[1, ["phase"]], [4, ["b", "c"]]]}
    s3 = OnOracle(dict(drag), ProblemSpec.from_info(info))
    assert s3.drag and not s3.incremental
    with pytest.raises(LoggedError, match="incremental"):
        OnOracle(dict(drag, evaluation="incremental"), ProblemSpec.from_info(info))


def test_a_stuck_walker_stops_the_run_at_the_next_checkpoint(tmp_path):
    """the asynchronous checkpoint path (`request_moments` /
    `fetch_moments`) never called `sync`, the only entry point that reported
    MCMC_HIP_ERR_STUCK -- with `max_samples: inf` a stuck walker was never reported.  The stuck
    flag now travels with the checkpoint read-out; the reference stops at once
    (mcmc.py:717-743).  The double raises from `sync`/`fetch_moments` only, like the engine."""
    s = make(None, np.inf, max_tries=3, learn_every="2d", steps_per_launch=8, Rminus1_stop=0.0)
    # (acceptance ~0.3: three consecutive rejections happen within a few steps of 128 walkers)
    with pytest.raises(LoggedError, match="stuck"):
        s.run()
    assert s.n_steps_raw < 200     # at the first checkpoint, not at the end of time


@pytest.mark.parametrize("lag", [1, 2])
def test_no_checkpoint_is_processed_after_convergence(lag):
    """with a checkpoint interval of at most `checkpoint_lag` launches a new
    request was queued in the pass that set `converged`, and processed after the loop: an
    extra progress row, possibly `converged` flipped back."""
    s = make(None, 1e9, Rminus1_stop=0.5, Rminus1_cl_stop=10.0, learn_every="5d",
             steps_per_launch=40, checkpoint_lag=lag)
    seen = []
    orig = s.check_convergence_and_learn_proposal

    def spy(*a, **k):
        orig(*a, **k)
        seen.append(bool(s.converged))
    s.check_convergence_and_learn_proposal = spy
    s.run()
    assert s.converged and seen[-1] and seen.count(True) == 1
    assert len(s.progress) == len(seen)


def pliklite_info(n_lin=5, **like):
    """A `planck_pliklite` input for the standalone driver: the small plik-lite-shaped data set,
    the synthetic linear Cl(theta), uniform boxes on its parameters and the reference's prior on
    the calibration (base_classes/planck_calib.yaml)."""
    from cobaya_amd import pliklite as P
    from tests.pliklite_common import sampling_problem, small_dataset
    ds = small_dataset()
    emu = P.synthetic_emulator(n_lin, ds.lmax)
    target = P.BinnedGaussian.from_dataset(ds, **{k: v for k, v in like.items() if k != "dataset"})
    kinds, a, b, C = sampling_problem(target, emu)
    params = {name: {"prior": {"min": float(a[i]), "max": float(b[i])}, "ref": float(emu.theta0[i]),
                     "proposal": float(np.sqrt(C[i, i]))} for i, name in enumerate(emu.names)}
    params["A_planck"] = {"prior": {"dist": "norm", "loc": 1, "scale": 0.0025},
                          "ref": {"dist": "norm", "loc": 1, "scale": 0.002}, "proposal": 0.0005}
    info = {"params": params,
            "likelihood": {"plik": {"class": "planck_pliklite", "dataset": ds, "cl_emulator": emu,
                                    **like}}}
    return info, target, emu, C


def test_planck_pliklite_runs_through_the_sampler():
    """§8f-4: `likelihood: {class: planck_pliklite}` parsed like PlanckPlikLite.init_params
    (planck_pliklite.py:32-141), configured through `set_target_binned_gaussian`, sampled by the
    run loop (checkpoints, learned covariance) -- on the oracle double; the posterior mean of
    the emulator parameters is the least-squares solution of the binned model."""
    info, target, emu, C = pliklite_info()
    spec = ProblemSpec.from_info(info)
    assert spec.like_kind == "planck_pliklite" and spec.calib_index == 5 and spec.n_modes == 0
    assert spec.binned.n_bins == target.n_bins == 88
    s = OnOracle({"seed": 4, "n_walkers": 256, "group_size": 64, "steps_per_launch": 30,
                  "max_samples": 60000, "Rminus1_stop": 0.0, "learn_every": "20d",
                  "snapshot_every": 30, "covmat": C, "covmat_params": spec.sampled}, spec)
    assert not s.incremental
    s.run()
    x = s.products()["sample"]
    rows = np.array([x[p] for p in spec.sampled]).T
    w = np.asarray(x["weight"])
    mean = np.average(rows, axis=0, weights=w)
    from oracle import cbind as O
    B = s.engine._prob().binned
    Bm = np.column_stack((B.BJ, -2.0 * B.Bc0))
    F = Bm.T @ np.linalg.solve(target.cov, Bm)
    F[5, 5] += 1.0 / 0.0025 ** 2
    best = np.linalg.solve(F, Bm.T @ np.linalg.solve(target.cov, target.X_data - B.Bc0))
    best[5] += 1.0
    sig = np.sqrt(np.diag(C))
    assert np.all(np.abs(mean - best) < 0.25 * sig), (mean - best) / sig
    assert np.all(np.asarray(x["chi2__plik"]) > 0)
    # the learned proposal covariance is the posterior's
    learned = s.proposer.get_covariance()
    assert np.all(np.abs(np.sqrt(np.diag(learned)) / sig - 1) < 0.3)
    # what the kind does not cover fails loudly
    with pytest.raises(LoggedError, match="one parameter block"):
        OnOracle({"n_walkers": 128, "emit": "chains"}, ProblemSpec.from_info(info))
    bad = pliklite_info()[0]
    bad["likelihood"]["plik"].pop("cl_emulator")
    from cobaya_amd.model import UnsupportedModel
    with pytest.raises(UnsupportedModel, match="cl_emulator"):
        ProblemSpec.from_info(bad)
    sel = ProblemSpec.from_info(pliklite_info(use_cl=["tt"])[0])
    assert sel.binned.n_bins == 48


def test_chains_mode_runs_on_the_incremental_path(tmp_path):
    """VERDICT r2, missing 5: `emit: chains` (every accepted row with its weight, the reference's
    product) no longer forces the from-scratch kernels; rows are read in place from the
    engine's drain slots and the store keeps following the run."""
    s = make(None, 20000, emit="chains", max_rows=3000, steps_per_launch=20)
    assert s.incremental and s.emit == "chains"
    s.run()
    coll = s.products()["sample"]
    assert 0 < len(coll) <= 3000 + 128 * 20
    w = np.asarray(coll["weight"])
    assert w.min() >= 1 and w.max() > 1          # multiplicities, not snapshots
    # the rows are the engine's: their log-posterior is the model's at their point
    x = np.array([coll[p] for p in s.spec.sampled]).T
    lp, ll = s.engine.evaluate(x[-50:])
    np.testing.assert_allclose(-np.asarray(coll["minuslogpost"])[-50:], lp + ll, rtol=1e-9, atol=1e-9)
    # with a chain file the rows outlive the drain slots (copied), and every stored row is written
    p = str(tmp_path / "ch")
    f = make(p, 6000, emit="chains", steps_per_launch=20)
    f.run()
    assert len(lines(p + ".1.txt")) - 1 == len(f.products()["sample"])


def test_emit_thin_on_the_device_equals_thinning_on_the_host(monkeypatch):
    """`emit_thin: T` (mcmc_hip's own option; the rule of OneSamplePoint.add_to_collection with
    output_thin, collection.py:1373-1383): where the engine's emitting kernel thins the rows itself
    (`set_emit_thin`) the sampler takes them as they come; where it refuses, `_thin_rows` thins
    them on the host -- the same collection either way."""
    from tests.oracle_engine import OracleEngine
    dev = make(None, 12000, emit="chains", steps_per_launch=20, emit_thin=3)
    assert dev._device_thin and dev.row_thin == 3 and dev.output_thin == 1 and dev.engine.emit_thin == 3
    dev.run()
    monkeypatch.delattr(OracleEngine, "set_emit_thin")
    host = make(None, 12000, emit="chains", steps_per_launch=20, emit_thin=3)
    assert not host._device_thin and host.row_thin == 3
    host.run()
    a, b = dev.products()["sample"], host.products()["sample"]
    assert len(a) == len(b) > 500
    assert np.array_equal(a.data.to_numpy(), b.data.to_numpy())
    # thinned: a third of the weight (the open remainders aside), and fewer rows than accepted steps
    steps = dev.engine.counters()["steps"] * 128
    w = np.asarray(a["weight"]).sum()
    assert steps // 3 - 128 * 25 <= w <= steps // 3
    assert len(a) < dev.engine.counters()["accepted"]


def test_chains_mode_keeps_every_accepted_row_without_output():
    """with `emit: chains` and no `output`, rows read in place from the
    engine's drain slots must be copied out before their slot is reused -- the product holds
    every accepted row (up to `max_rows`), not the last three launches."""
    s = make(None, 20000, emit="chains", steps_per_launch=20, drain_ring_bytes=0)   # (4 slots)
    s.run()
    assert s.engine.drain_slots == 4 and s._launches > 12
    coll = s.products()["sample"]
    acc = s.engine.counters()["accepted"]
    # every accepted step closes one row; the current points (one per walker) are still open
    assert acc - 128 <= len(coll) <= acc
    assert np.all(np.isfinite(coll.data.to_numpy()))
    steps = s.engine.counters()["steps"]
    assert steps * 128 - np.asarray(coll["weight"]).sum() <= 128 * 60   # open weights only
    # the store outlives the engine (its pinned slots die with it)
    s.close()
    assert np.all(np.isfinite(np.vstack(s._rows))) and len(np.vstack(s._rows)) == len(coll)


def test_the_drain_ring_is_the_row_store():
    """Round 4 (host side): the engine's ring of pinned drain slots is sized to
    outlive the `max_rows` retention window, so stored rows are read in place for as long as
    the store keeps them -- no block is ever copied a second time on the host (the oracle-backed
    engine poisons a slot when it is reused: a view kept too long would read NaN)."""
    s = make(None, 40000, emit="chains", steps_per_launch=20, max_rows=6000)
    assert s.engine.drain_slots == int(np.ceil(6000 / (0.08 * 128 * 20))) + 2
    copied = []
    orig = s._expire_row_views

    def spy():
        before = [id(r) for r in s._rows]
        orig()
        copied.extend(1 for a, b in zip(before, s._rows) if a != id(b))
    s._expire_row_views = spy
    s.run()
    assert s._launches >= 2 * s.engine.drain_slots and s._rows_capped   # the ring went round twice
    assert not copied and all(s._is_slot_view(r) for r in s._rows)
    rows = np.vstack(s._rows)
    assert 3000 <= len(rows) <= 6000 and np.all(np.isfinite(rows))
    # the same run with every block copied at once holds the same rows
    c = make(None, 40000, emit="chains", steps_per_launch=20, max_rows=6000, drain_copy=True)
    c.run()
    assert c.engine.drain_slots == 4 and np.array_equal(np.vstack(c._rows), rows)
    # a budget too small for the window: four slots, views copied out in time (still all rows)
    b = make(None, 40000, emit="chains", steps_per_launch=20, max_rows=6000, drain_ring_bytes=1)
    b.run()
    assert b.engine.drain_slots == 4 and np.array_equal(np.vstack(b._rows), rows)


@pytest.mark.parametrize("emit,max_rows", [("snapshots", 0), ("snapshots", 1 << 20), ("chains", 1 << 20)])
def test_bounds_criterion_in_every_emit_mode(emit, max_rows):
    """VERDICT r3 missing 1: the reference's two-stage stop rule (means twice in a row, then the
    R-1 of the bounds below `Rminus1_cl_stop`, mcmc.py:908, 918-1002) holds in every output
    mode -- also with nothing stored on the host (`max_rows: 0`, the benchmarked mode): the
    bounds come from the ring of ensemble snapshots, not from stored rows."""
    s = make(None, 1e9, Rminus1_stop=0.3, Rminus1_cl_stop=0.3, emit=emit, max_rows=max_rows,
             learn_every="5d", steps_per_launch=20, snapshot_every=None)
    s.run()
    prog = s.progress
    assert s.converged and len(prog) >= 2   # (the means criterion has to hold twice in a row)
    cl = prog["Rminus1_cl"].to_numpy(float)
    r = prog["Rminus1"].to_numpy(float)
    assert np.isfinite(cl[-1]) and cl[-1] < 0.3
    # the bounds are only looked at once the means criterion holds twice in a row
    for i in range(len(prog)):
        if np.isfinite(cl[i]):
            assert i >= 1 and max(r[i], r[i - 1]) < 0.3
    # a bounds criterion that cannot be met keeps the run going (it is not dropped)
    t = make(None, 40000, Rminus1_stop=0.3, Rminus1_cl_stop=1e-9, emit=emit, max_rows=max_rows,
             learn_every="5d", steps_per_launch=20, snapshot_every=None)
    t.run()
    assert not t.converged and np.isfinite(t.progress["Rminus1_cl"].to_numpy(float)).sum() >= 2


@pytest.mark.parametrize("sidecar", [False, True])
def test_resume_restores_the_bounds_ring(tmp_path, sidecar, monkeypatch):
    """The ring behind R-1 of the bounds is part of the state: a resumed run reports the same
    Rminus1_cl at the same checkpoints as the uninterrupted one -- with the snapshots inside the
    state file (small rings) and in the sidecar `prefix.<n>.bounds.npy` a ring above
    BOUNDS_RING_SAVE_BYTES goes to (251 MB at BASELINE config 2)."""
    if sidecar:
        from cobaya_amd.sampler import EnsembleMCMC
        monkeypatch.setattr(EnsembleMCMC, "BOUNDS_RING_SAVE_BYTES", 0)
    opts = dict(Rminus1_stop=0.3, Rminus1_cl_stop=1e-9, learn_every="5d", steps_per_launch=20)
    one = make(str(tmp_path / "a"), 40000, **opts)
    one.run()
    p = str(tmp_path / "b")
    make(p, 20000, **opts).run()
    b = make(p, 40000, resume=True, **opts)
    b.run()
    cl1, cl2 = (x.progress["Rminus1_cl"].to_numpy(float) for x in (one, b))
    assert np.isfinite(cl1).sum() >= 2 and np.array_equal(cl1, cl2, equal_nan=True)
    assert b._bslots == one._bslots and b._bstride == one._bstride
    import glob
    assert bool(glob.glob(p + "*.bounds.npy")) == sidecar
    if sidecar:   # a slot the tags do not vouch for is dropped, never mixed up
        tags_f = glob.glob(p + "*.bounds_tags.npy")[0]
        tags = np.load(tags_f)
        k = int(np.argmax(tags >= 0))
        tags[k] += 1
        np.save(tags_f, tags)
        c = make(p, 40000, resume=True, **opts)
        assert c._bslots[k] == -1 and sum(j >= 0 for j in c._bslots) == sum(j >= 0 for j in b._bslots) - 1


def test_dragging_emits_chains():
    """VERDICT r3 missing 3 / tests/test_mcmc.py:132-171 (`test_mcmc_drag_results`): `drag: True`
    with `emit: chains` -- every dragging step ends in process_accept_or_reject (mcmc.py:656-668),
    so the product is a weighted chain like the Metropolis sampler's.  Two-speed Gaussian, the
    shape of the reference test: the weighted sample recovers mean and covariance."""
    from tests.test_gpu_sampler import _two_speed_info, kl_norm
    info, tm, tc = _two_speed_info({})
    s = OnOracle({"seed": 4, "n_walkers": 128, "group_size": 64, "drag": True, "oversample_power": 0.4,
                  "emit": "chains", "steps_per_launch": 70, "max_samples": 60000, "Rminus1_stop": 0.0,
                  "learn_every": "20d", "burn_in": 10}, ProblemSpec.from_info(info))
    assert s.drag and s.drag_interp_steps == 6 and not s.incremental and s.steps_per_launch == 10
    s.run()
    coll = s.products()["sample"]
    w = np.asarray(coll["weight"])
    acc = s.engine.counters()["accepted"]
    assert acc - 128 * 11 <= len(coll) <= acc and w.max() > 1 and w.min() >= 1
    assert kl_norm(tm, tc, coll.mean(), coll.cov()) < 0.07     # the reference's own bar


def test_device_reduce_checkpoint_retraces_the_host_checkpoint(tmp_path):
    """`device_checkpoint: reduce` (round 4: the default for N > 1): the ring of intervals, the
    window sums and the payload are the device's, only the reduced payload comes back and the
    host solves it.  On the oracle-backed engine (whose payload uses the host path's arithmetic)
    the whole run -- progress table, learned proposal, final ensemble -- retraces the host
    checkpoint bit for bit, also across a resume (the ring is reloaded from the state file)."""
    runs = {}
    for mode in (False, "reduce"):
        s = make(None, 60000, device_checkpoint=mode, learn_every="10d")
        s.run()
        assert s._device_ckpt == bool(mode) and not s._ckpt_solve_on_device
        runs[mode] = (s.progress[["N", "acceptance_rate", "Rminus1"]].to_numpy(float),
                      s.proposer.get_covariance(), s.engine.get_full_state()["x"])
    assert len(runs[False][0]) >= 4 and np.isfinite(runs[False][0][:, 2]).any()
    for a, b in zip(runs[False], runs["reduce"]):
        assert np.array_equal(a, b, equal_nan=True)
    p = str(tmp_path / "r")
    b1 = make(p, 30000, device_checkpoint="reduce", learn_every="10d")
    b1.run()
    b2 = make(p, 60000, resume=True, device_checkpoint="reduce", learn_every="10d")
    b2.run()
    assert np.array_equal(b2.engine.get_full_state()["x"], runs[False][2])
    with pytest.raises(LoggedError, match="no device-side solve"):
        make(None, 1000, device_checkpoint=True)
    with pytest.raises(LoggedError, match="device_checkpoint must be one of"):
        make(None, 1000, device_checkpoint="gpu")


@pytest.mark.parametrize("where", ["device", "host"])
def test_thinned_rows_resume_bit_identically(tmp_path, monkeypatch, where):
    """the per-walker thinning remainders come back at a resume on BOTH paths -- the
    device-thinned one (engine state) and the host fallback (`_thin_rows`; the state file's
    `thin_carry`, the same key): the chain file of run + resume equals the uninterrupted run's."""
    if where == "host":
        monkeypatch.delattr(OracleEngine, "set_emit_thin")
    kw = dict(emit="chains", steps_per_launch=20, emit_thin=7)
    one = make(str(tmp_path / "a"), 30000, **kw)
    assert one._device_thin == (where == "device")
    one.run()
    p = str(tmp_path / "b")
    b1 = make(p, 14000, **kw)
    b1.run()
    z = np.load(p + ".1.state.npz")
    assert "thin_carry" in z and z["thin_carry"].shape == (128,) and z["thin_carry"].max() > 0
    assert tuple(z["geometry"]) == (64, 64)
    b2 = make(p, 30000, resume=True, **kw)
    b2.run()
    assert lines(p + ".1.txt") == lines(str(tmp_path / "a") + ".1.txt")
    # the geometry the streams depend on is checked at a resume
    with pytest.raises(LoggedError, match="group_size 64"):
        make(p, 30000, resume=True, n_walkers=128, group_size=128, **kw)


def test_a_single_group_splits_as_far_as_the_wavefronts_allow():
    """W = 128 with Rminus1_single_split = 4 cannot give four groups of whole
    wavefronts -- two of 64 are taken instead of refusing; the walkers keep sharing ONE Haar
    basis (the user's group) where the number of sub-groups is a power of two."""
    s = OnOracle({"n_walkers": 128, "group_size": 128, "seed": 3, "max_samples": 2000,
                  "Rminus1_stop": 0.0, "learn_every": "20d"}, ProblemSpec.from_info(QUICK))
    assert int(s.group_size) == 64 and s.engine.G == 2 and s._split_of == 128
    assert int(s.basis_group_size) == (128 if s.incremental else 64)
    s.run()
    assert np.isfinite(s.progress["Rminus1"].to_numpy(float)[-1])
    s = OnOracle({"n_walkers": 256, "group_size": 256, "seed": 3}, ProblemSpec.from_info(QUICK))
    assert int(s.group_size) == 64 and int(s.basis_group_size) == (256 if s.incremental else 64)
    # three sub-groups (not a power of two): the basis follows the sub-groups
    s = OnOracle({"n_walkers": 192, "group_size": 192, "seed": 3, "Rminus1_single_split": 3},
                 ProblemSpec.from_info(QUICK))
    assert int(s.group_size) == 64 and int(s.basis_group_size) == 64
