"""Pins oracle/ref_numpy.py (numpy-stream restatement) against the golden vectors that
tests/golden/make_golden.py produced by importing the reference (Tier A)."""
import json

import numpy as np
import pytest

from oracle import ref_numpy as R


def test_g1_transforms(golden):
    g = golden("g1_transforms")
    for name in ("d2", "d3", "d30", "d100", "d5_2blocks", "d5_3blocks"):
        flat, lens = g[name + "_blocks"], g[name + "_blocklens"]
        blocks, k = [], 0
        for n in lens:
            blocks.append([int(i) for i in flat[k:k + n]])
            k += n
        _, _, tr = R.transforms_from_cov(g[name + "_cov"], blocks)
        for i, t in enumerate(tr):
            np.testing.assert_allclose(t, g[f"{name}_transform{i}"], rtol=1e-14, atol=0)


def test_g1_rejects_non_pd():
    with pytest.raises(ValueError):
        R.transforms_from_cov(np.array([[1.0, 2.0], [2.0, 1.0]]), [[0, 1]])
    with pytest.raises(ValueError):
        R.transforms_from_cov(np.array([[1.0, 0.5], [0.1, 1.0]]), [[0, 1]])


def test_g4_prior(golden):
    g = golden("g4_prior")
    kinds = g["kinds"]
    a = np.where(kinds == 0, g["bounds"][:, 0], g["loc"])
    b = np.where(kinds == 0, g["bounds"][:, 1], g["scale"])
    pr = R.Prior(kinds, a, b, periodic=g["periodic"])
    assert pr.uniform_logp == float(g["uniform_logp"])
    got = np.array([pr.logp(p) for p in g["points"]])
    ref = g["logprior"]
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    assert np.isinf(ref).sum() > 10 and (~np.isinf(ref)).sum() > 3
    m = ~np.isinf(ref)
    np.testing.assert_allclose(got[m], ref[m], rtol=1e-15)
    wrapped = np.array([pr.reduce_periodic(p.copy()) for p in g["points"]])
    assert np.array_equal(wrapped, g["wrapped"])


def test_g5_loglike(golden):
    g = golden("g5_loglike")
    for tag in ("gm_d2_K1", "gm_d3_K1", "gm_d3_K3", "gm_d4_K2", "gm_d30_K1", "gm_d30_K3",
                "gm_d100_K1"):
        tgt = R.GaussianMixtureTarget(g[tag + "_means"], g[tag + "_covs"],
                                      g[tag + "_weights"])
        got = np.array([tgt.loglike(p) for p in g[tag + "_points"]])
        np.testing.assert_allclose(got, g[tag + "_loglike"], rtol=1e-12, atol=1e-11)
        der = np.array([tgt.derived(p) for p in g[tag + "_points"]])
        np.testing.assert_allclose(der, g[tag + "_derived"], rtol=1e-9, atol=1e-10)
    for tag in ("gauss_d3_norm1", "gauss_d27_norm1", "gauss_d27_norm0"):
        tgt = R.GaussianTarget(g[tag + "_mean"], g[tag + "_cov"],
                               normalized=tag.endswith("1"))
        got = np.array([tgt.loglike(p) for p in g[tag + "_points"]])
        np.testing.assert_allclose(got, g[tag + "_loglike"], rtol=1e-12, atol=1e-11)


PRIORS = {
    "quick": dict(kinds=[0, 1], a=[-0.5, 0.0], b=[3.0, 1.0]),
    "fixed3": dict(kinds=[0, 0, 0], a=[-1.0] * 3, b=[1.0] * 3),
    "d30": dict(kinds=[0] * 30, a=[0.0] * 30, b=[1.0] * 30),
    "d4": dict(kinds=[0] * 4, a=[0.0] * 4, b=[1.0] * 4),
}


def replay(g, name, learn, record_draws=False):
    key = lambda k: g[f"{name}__{k}"]  # noqa: E731
    rng = np.random.Generator(np.random.PCG64())
    rng.bit_generator.state = json.loads(str(key("rng_state")))
    prior = R.Prior(**PRIORS[name.split("_")[0]])
    target = R.GaussianMixtureTarget(key("means"), key("covs"), key("weights"))
    chain = R.RefChain(
        prior, target, key("cov0"), key("x0"), rng, temperature=float(key("temperature")),
        proposal_scale=float(key("proposal_scale")), burn_in=int(key("burn_in")),
        max_tries=float(key("max_tries")), learn_proposal=learn,
        learn_every=int(key("learn_every")), learn_Rminus1_max=30.0,
        record_draws=record_draws)
    assert chain.logpost == pytest.approx(float(key("logpost0")), rel=1e-13)
    n_rows = key("data").shape[0]
    while len(chain.rows) < n_rows:
        chain.step()
    return chain


@pytest.mark.parametrize("name,learn", [
    ("quick_nolearn", False), ("quick_learn", True), ("fixed3_T1", False),
    ("fixed3_T2", True), ("d30_covmat", False), ("d4_K2", True)])
def test_g6_chain_traces(golden, name, learn):
    """Tier A: identical accept/reject sequence (integer weights) and values to a few ulp
    over the whole reference trace, including covariance learning."""
    g = golden("g6_traces")
    chain = replay(g, name, learn)
    data = g[f"{name}__data"]
    cols = [str(c) for c in g[f"{name}__columns"]]
    d = chain.d
    rows = np.array(chain.rows)
    assert chain.n_steps == int(g[f"{name}__n_steps_raw"])
    assert np.array_equal(rows[:, 0], data[:, cols.index("weight")])
    np.testing.assert_allclose(rows[:, 2:2 + d], data[:, 2:2 + d], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(rows[:, 1], data[:, cols.index("minuslogpost")],
                               rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(rows[:, 2 + d], data[:, cols.index("minuslogprior")],
                               rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(rows[:, 3 + d], data[:, cols.index("chi2")],
                               rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(chain.x, g[f"{name}__final_x"], rtol=1e-12)
    assert chain.weight == int(g[f"{name}__final_weight"])
    # G7 (single-chain split mode): progress table and learned covariances
    prog = np.array(chain.progress)
    n_prog = len(g[f"{name}__progress_N"])
    assert len(prog) == n_prog
    if n_prog:
        np.testing.assert_allclose(prog[:, 0], g[f"{name}__progress_N"])
        np.testing.assert_allclose(prog[:, 1], g[f"{name}__progress_acc"], rtol=1e-13)
        np.testing.assert_allclose(prog[:, 2], g[f"{name}__progress_Rminus1"], rtol=1e-8)
    ref_learned = g[f"{name}__learned_covs"]
    assert len(chain.learned) == len(ref_learned)
    for a, b in zip(chain.learned, ref_learned):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-18)


@pytest.mark.parametrize("name,learn", [
    ("over_thin", False), ("over_nothin", True), ("blocks_1d", False), ("drag", False),
    ("drag_learn", True)])
def test_g10_blocked_oversampled_dragged_traces(golden, name, learn):
    """Tier A for (f)1: blocks, oversampling, thinned output and dragging replay the
    reference's chains with identical weights (proposal.py:96-260, mcmc.py:320-410, 564-668)."""
    g = golden("g10_blocked")
    key = lambda k: g[f"{name}__{k}"]  # noqa: E731
    rng = np.random.Generator(np.random.PCG64())
    rng.bit_generator.state = json.loads(str(key("rng_state")))
    prior = R.Prior(kinds=[0] * 5, a=[0.0] * 5, b=[1.0] * 5)
    target = R.GaussianMixtureTarget(g["means"], g["covs"])
    sizes, flat = key("block_sizes"), key("block_params").tolist()
    blocks = [flat[sum(sizes[:i]):sum(sizes[:i + 1])] for i in range(len(sizes))]
    chain = R.BlockedRefChain(
        prior, target, key("cov0"), key("x0"), rng, blocks,
        oversampling=key("oversampling").tolist(), drag_last_slow=int(key("drag_last_slow")),
        drag_interp_steps=int(key("drag_interp_steps")), output_thin=int(key("output_thin")),
        temperature=float(key("temperature")), proposal_scale=float(key("proposal_scale")),
        burn_in=int(key("burn_in")), max_tries=float(key("max_tries")), learn_proposal=learn,
        learn_every=int(key("learn_every")), learn_Rminus1_max=30.0)
    assert chain.cycle_length == int(key("cycle_length"))
    data = key("data")
    cols = [str(c) for c in key("columns")]
    while len(chain.rows) < len(data):
        chain.step()
    rows = np.array(chain.rows)
    assert chain.n_steps == int(key("n_steps_raw"))
    assert np.array_equal(rows[:, 0], data[:, cols.index("weight")])
    np.testing.assert_allclose(rows[:, 2:7], data[:, 2:7], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(rows[:, 1], data[:, cols.index("minuslogpost")],
                               rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(chain.x, key("final_x"), rtol=1e-12)
    assert chain.weight == int(key("final_weight"))
    ref_learned = key("learned_covs")
    assert len(chain.learned) == len(ref_learned)
    for a, b in zip(chain.learned, ref_learned):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-18)


def test_g7_multichain_rminus1(golden):
    g = golden("g7_multichain")
    cols = [str(c) for c in g["columns"]]
    Ns, means, covs, accs = [], [], [], []
    for i in range(6):
        data = g[f"chain{i}"]
        n = len(data)
        first = int(n / 2)
        w = data[first:, cols.index("weight")]
        xs = data[first:, 2:5]
        Ns.append(n)
        means.append(R.weighted_mean(xs, w))
        covs.append(R.weighted_cov(xs, w))
        accs.append((n - first) / w.sum())
    np.testing.assert_allclose(means, g["means"], rtol=1e-14)
    np.testing.assert_allclose(covs, g["covs"], rtol=1e-12)
    np.testing.assert_allclose(accs, g["acceptance_rates"], rtol=1e-14)
    Rm1, mean_of_covs = R.rminus1_of_means(np.array(Ns, float), np.array(means),
                                           np.array(covs))
    assert Rm1 == pytest.approx(float(g["Rminus1"]), rel=1e-10)
    np.testing.assert_allclose(mean_of_covs, g["new_proposal_cov"], rtol=1e-13)
    assert np.average(accs, weights=Ns) == pytest.approx(float(g["progress_acc"]))


def test_g9_initial_covmat(golden):
    g = golden("g9_initial_covmat")
    order, kind, full = g["order"], g["kind"], g["full_cov"]
    names = [f"a_{i}" for i in order]
    sig = np.sqrt(np.diag(full))
    proposal = [sig[i] if kind[i] == 1 else None for i in order]
    ref_var = np.array([(2 * sig[i]) ** 2 if kind[i] == 2 else np.nan for i in order])
    prior_var = np.array([(2 * sig[i]) ** 2 if kind[i] == 3 else 1000.0 ** 2
                          for i in order])
    cov, where_nan = R.initial_proposal_covmat(
        names, g["reduced"], [f"a_{i}" for i in g["i_cov"]], proposal, ref_var, prior_var)
    np.testing.assert_allclose(cov, g["got"], rtol=1e-14)
    np.testing.assert_allclose(cov, g["expected"], rtol=1e-12)
    assert where_nan.sum() == 30


def test_haar_so_n_is_special_orthogonal():
    rng = np.random.default_rng(3)
    for n in (2, 3, 30):
        H = R.haar_so_n(n, rng)
        np.testing.assert_allclose(H @ H.T, np.eye(n), atol=1e-13)
        assert np.linalg.det(H) == pytest.approx(1.0, abs=1e-12)


def test_g2_haar_matrix_from_the_numpy_stream(golden):
    """G2 (a3): from the same PCG64 state, haar_so_n returns the reference's random_SO_N matrix
    and leaves the stream at the same position."""
    g = golden("g2_g8_haar_chainstats")
    for n in (2, 3, 5, 30):
        rng = np.random.Generator(np.random.PCG64())
        rng.bit_generator.state = json.loads(str(g[f"haar_state_{n}"]))
        H = R.haar_so_n(n, rng)
        np.testing.assert_allclose(H, g[f"haar_{n}"], rtol=0, atol=4e-16)
        assert rng.standard_normal() == float(g[f"haar_next_normal_{n}"])
        np.testing.assert_allclose(H @ H.T, np.eye(n), atol=1e-14)
        assert np.linalg.det(H) == pytest.approx(1.0, abs=1e-12)
