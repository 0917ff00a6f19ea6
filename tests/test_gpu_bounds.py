"""R-1 of the confidence-interval bounds on the device (mcmc.py:918-1002):
`mcmc_hip_bounds_*` through the C ABI against the oracle's restatement of GetDist's
`confidence` (oracle/ref_numpy.py; parity with GetDist itself unpinned -- it is absent).  The
bounds are order statistics, so the comparison is exact."""
import numpy as np
import pytest

from cobaya_amd.engine import Engine, EngineError
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def engine(d, W, gs):
    e = Engine(d, W, group_size=gs, device=0, seed=3)
    e.set_prior([0] * d, [-50.0] * d, [50.0] * d)
    e.set_target_one()
    e.set_proposal_cov(np.eye(d))
    return e


@pytest.mark.parametrize("d,W,gs,n_slots,window", [
    (3, 256, 64, 4, [2]), (5, 1024, 64, 8, [1, 5, 6]), (30, 2048, 128, 16, list(range(4, 16))),
    (30, 1024, 256, 64, list(range(64))), (100, 512, 64, 6, [5, 0, 3]), (1, 128, 64, 2, [0, 1])])
def test_bounds_are_the_oracles_order_statistics(d, W, gs, n_slots, window):
    rng = np.random.default_rng(d * 1000 + W)
    e = engine(d, W, gs)
    e.set_state(rng.normal(size=(W, d)))
    shift = rng.normal(size=d) * 0.1
    e.set_moment_shift(shift)
    e.bounds_configure(n_slots)
    ring = {}
    for k in set(window):
        x = rng.normal(size=(W, d)) * 10 ** rng.uniform(-3, 1, size=d)
        x[rng.random(size=x.shape) < 0.05] = 0.0           # ties, and both zeros
        x[rng.random(size=x.shape) < 0.02] = -0.0
        x[:, 0] = np.round(x[:, 0], 1)                      # many duplicates
        e.bounds_set_slot(k, x)
        np.testing.assert_array_equal(e.bounds_get_slot(k), x)
        ring[k] = x
    G = W // gs
    for limfrac in (0.475, 0.025, 1.0 / 3.0, 0.5, 1e-6):
        stats, b = e.bounds_statistics(window, limfrac, want_bounds=True)
        chains = [np.vstack([ring[k][g * gs:(g + 1) * gs] for k in window]) for g in range(G)]
        ref = R.bounds_of_chains(chains, [np.ones(len(c)) for c in chains], 2 * limfrac)
        assert np.array_equal(b, ref)          # (-0.0 == 0.0: numpy's sort does not order them either)
        pay = R.bounds_payload(b, shift)
        assert np.array_equal(stats, pay), np.max(np.abs(stats - pay))
        cov = np.diag(rng.uniform(0.5, 2.0, size=d))
        assert abs(R.rminus1_of_bounds_from_payload(stats, cov) - R.rminus1_of_bounds(ref - shift[None, :, None], cov)) < 1e-9
    e.close()


def test_bounds_snapshot_takes_the_current_points_in_stream_order():
    d, W, gs = 4, 512, 64
    rng = np.random.default_rng(1)
    e = engine(d, W, gs)
    e.set_state(rng.normal(size=(W, d)))
    e.bounds_configure(3)
    seen = []
    for k in range(3):
        e.step(7)
        e.bounds_snapshot(k)          # queued behind the steps, before the next ones
        e.step(1)
        e.sync()
        seen.append(e.bounds_get_slot(k))
    x_end = e.get_state()["x"]
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[2], x_end)
    g = engine(d, W, gs)             # the same chain, stopped where the snapshots were taken
    rng = np.random.default_rng(1)
    g.set_state(rng.normal(size=(W, d)))
    for k in range(3):
        g.step(7)
        g.sync()
        assert np.array_equal(g.get_state()["x"], seen[k]), k
        g.step(1)
    g.close()
    e.close()


def test_bounds_arguments_are_checked():
    e = engine(3, 256, 256)
    with pytest.raises(EngineError, match="bounds_configure"):
        e.bounds_statistics([0], 0.475)
    e.bounds_configure(80)
    with pytest.raises(EngineError, match="at most"):
        e.bounds_statistics(list(range(65)), 0.475)
    with pytest.raises(EngineError, match="slot"):
        e.bounds_snapshot(80)
    with pytest.raises(EngineError, match="limfrac"):
        e.bounds_statistics([0], 1.5)
    e.close()


def test_sampler_stops_on_the_bounds_criterion_with_nothing_stored():
    """The benchmarked output mode (`emit: snapshots`, `max_rows: 0`) now runs the reference's
    two-stage stop rule: Rminus1_cl in `progress` comes from the device ring."""
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    from tests.test_host_logic import QUICK
    s = MCMCHip({"seed": 4, "n_walkers": 4096, "group_size": 64, "steps_per_launch": 20,
                 "learn_every": "5d", "Rminus1_stop": 0.1, "Rminus1_cl_stop": 0.15, "max_rows": 0,
                 "max_samples": 1e9}, ProblemSpec.from_info(QUICK))
    s.run()
    prog = s.progress
    cl = prog["Rminus1_cl"].to_numpy(float)
    assert s.converged and np.isfinite(cl[-1]) and cl[-1] < 0.15
    assert prog["Rminus1"].iloc[-1] < 0.1 and len(s.products()["sample"]) == 0
    # n = 64 walkers x snapshots per chain: the bound's sampling noise, sqrt(q(1-q)/n)/pdf, ~ 0.05-0.15
    assert cl[-1] > 0.01
    s.close()


def test_getdist_pin_tool_on_the_device(capsys, monkeypatch):
    """`tools/check_getdist_bounds.py --device`: the committed reference chains (golden G7) through
    `mcmc_hip_bounds_statistics` against GetDist's `confidence` where GetDist is installed, against
    its restatement otherwise -- exact either way."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_getdist_bounds as T
    monkeypatch.setattr(sys, "argv", ["check_getdist_bounds.py", "--device"])
    assert T.main() == 0
    out = capsys.readouterr().out
    assert out.count("device bounds") == len(T.LIMFRACS) and "DIFFERENT" not in out
