"""Bit-exact parity AT THE BENCHMARK GEOMETRY ("Next round" 1a and 2).

`python bench.py` runs BASELINE configs[1] as 65 536 walkers, R-1 groups of 256 walkers
(= one workgroup), one Haar basis per 4 096 walkers, 1 200 fused Metropolis steps per launch,
incremental evaluation with the carried log-likelihood, re-anchored at every launch, and the
proposal refreshed by learn checkpoints between launches (mcmc.py:545-562, 670-748, 1009-1023).
The other parity tests stop at 2 048 walkers; here the SAME launches are compared with the C
oracle walker by walker -- x, y, logpost, logprior, loglike, weight, prior_rej, accept counts --
so workgroup / basis-group indexing, Philox counters and the direction staging are certified at
4 096 groups x 1 200 steps, not only the moments.

`walker_offset = 458 752` is rank 7's shard of BASELINE configs[2] (8 x 65 536 walkers): its
Philox counters and basis groups [112, 128) -- what the driver's 8-GPU run executes on the last
device -- against the oracle's walkers [458 752, 524 288).

The oracle needs about 0.6 s per launch on the 128 threads of the GPU box (orc_run cuts the wide
basis groups over its threads)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cobaya_amd import engine as E  # noqa: E402
from oracle import cbind as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "targets.npz")


def _u64(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    bad = (_u64(a) != _u64(b)) if a.dtype == np.float64 else (a != b)
    if bad.any():
        idx = np.argwhere(bad)[:4].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} values differ, first at {idx}")


def _bench_target(d):
    """bench.py's `target(d)` (same arrays: the golden d = 30 target, the seeded d = 100 one)."""
    g = np.load(GOLDEN)
    if f"mean_d{d}" in g:
        return g[f"mean_d{d}"], g[f"cov_d{d}"]
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d))
    s = 10 ** rng.uniform(-2, np.log10(0.05), size=d)
    c = A @ A.T / d + np.eye(d)
    c = c / np.sqrt(np.outer(np.diag(c), np.diag(c))) * np.outer(s, s)
    return np.full(d, 0.5), c


def _pair(d, W, gs, bgs, offset, seed=1, K=1):
    mean, cov = _bench_target(d)
    eng = E.Engine(d, W, group_size=gs, seed=seed, incremental=True, basis_group_size=bgs,
                   walker_offset=offset)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    if K == 1:
        means, covs, weights = mean, cov, None
        eng.set_target_gaussian_mixture([mean], [cov])
    else:   # bench.py's mixture variants: the other modes a sigma or so away, the same covariance
        r0 = np.random.default_rng(11)
        means = [mean] + [np.clip(mean + r0.normal(size=d) * np.sqrt(np.diag(cov)), 0.05, 0.95)
                          for _ in range(K - 1)]
        covs, weights = [cov] * K, [1.0 / K] * K
        eng.set_target_gaussian_mixture(means, covs, weights)
    eng.set_proposal_cov(cov)
    prob = O.Problem(d, [0] * d, [0.0] * d, [1.0] * d, means=means, covs=covs, weights=weights,
                     T=eng.get_proposal_transform(), group_size=bgs, seed=seed,
                     derived=eng.derived_constants(), incremental=True,
                     carry_modes=eng.carries_modes())
    rng = np.random.default_rng(1 + offset)
    x0 = np.clip(mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov)), 1e-6, 1 - 1e-6)
    eng.set_state(x0)
    return eng, prob, O.State(prob, x0), mean, cov


def _compare(eng, st, what):
    s = eng.get_full_state()
    _same(s["x"], st.x, what + ": x")
    _same(s["y"], st.y, what + ": carried whitened residual y")
    _same(s["logpost"], st.logpost, what + ": logpost")
    _same(s["logprior"], st.logprior, what + ": logprior")
    _same(s["loglike"], st.loglike, what + ": carried loglike")
    _same(s["weight"], st.weight, what + ": weight")
    _same(s["prior_rej"], st.prior_rej, what + ": prior_rej")
    _same(s["n_accept"], st.n_accept, what + ": per-walker accept counts")
    assert int(s["step"]) == st.step
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    return s


@pytest.mark.parametrize("offset", [0, 7 * 65536], ids=["rank0", "rank7-of-configs2"])
def test_config2_bench_launches_bit_exact(offset):
    d, W, gs, bgs, spl = 30, 65536, 256, 4096, 1200
    threads = O.max_threads()
    eng, prob, st, mean, cov = _pair(d, W, gs, bgs, offset)
    assert prob.refresh_every == spl           # a bench launch = one re-anchoring interval
    _compare(eng, st, "initial evaluation")
    for launch in range(3):
        # (the last call holds TWO launches: their directions are formed as one set and the
        # second launch refreshes y = L^-1 (x - mu) itself, inside step_inc_kernel)
        n = spl if launch < 2 else 2 * spl
        eng.step(n)
        eng.accumulate_moments()               # (the snapshot kernels between two launches)
        eng.sync()
        st.run(n, walker0=offset, n_threads=threads)
        s = _compare(eng, st, f"launch {launch}")
        kernel = eng.last_step_kernel()
        assert "step_inc_kernel" in kernel, kernel
        if launch < 2:
            # what a learn checkpoint does between two launches (mcmc.py:1009-1023): the
            # proposal becomes the covariance of the samples; here of the current ensemble
            learned = np.cov(st.x.T)
            eng.set_proposal_cov(learned)
            prob.set_T(eng.get_proposal_transform())
    acc = st.n_accept.sum() / (W * st.step)
    assert 0.15 < acc < 0.5
    # the ensemble the three launches leave samples the target (north star: 1 % needs 1e6
    # accepted samples; one snapshot of 65 536 walkers gives sigma / 256 on a mean)
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(s["x"].mean(0) - mean) / sig) < 5 / np.sqrt(W) * 1.5
    eng.close()


def test_config4_d100_bench_launch_bit_exact():
    """BASELINE configs[3]: d = 100, 65 536 walkers, one Haar basis per 16 384 walkers, one
    bench step = 4 000 Metropolis steps (the engine cuts it into several kernel launches where
    the directions exceed their buffer)."""
    d, W, gs, bgs, spl = 100, 65536, 256, 16384, 4000
    eng, prob, st, mean, cov = _pair(d, W, gs, bgs, 0)
    assert prob.refresh_every == spl
    eng.step(spl)
    eng.sync()
    st.run(spl, n_threads=O.max_threads())
    _compare(eng, st, "d = 100 launch")
    assert "step_inc_kernel" in eng.last_step_kernel()
    eng.close()


@pytest.mark.parametrize("K", [2, 3])
def test_mixture_bench_launch_bit_exact_on_two_lanes(K):
    """The mixture variants of bench.py (gaussian_mixture.py:156-163 with K modes at d = 30, 65 536
    walkers): at this size the engine takes step_duo_mix_kernel (two lanes per walker, round 6) by
    itself -- one bench launch of 1 200 steps and a second one behind a refreshed proposal, walker
    by walker against the C oracle, the carried mode log-densities included."""
    d, W, gs, bgs, spl = 30, 65536, 256, 1024, 1200
    eng, prob, st, mean, cov = _pair(d, W, gs, bgs, 0, K=K)
    threads = O.max_threads()
    for launch in range(2):
        eng.step(spl)
        eng.sync()
        st.run(spl, n_threads=threads)
        _compare(eng, st, f"K = {K}, launch {launch}")
        _same(eng.get_full_state()["amode"], st.amode, "carried mode log-densities")
        assert "step_duo_mix_kernel" in eng.last_step_kernel(), eng.last_step_kernel()
        if launch == 0:
            eng.set_proposal_cov(np.cov(st.x.T))
            prob.set_T(eng.get_proposal_transform())
    eng.close()


@pytest.mark.parametrize("W,want", [(32768, "step_inc_mix_kernel"), (49152, "step_duo_mix_kernel")])
def test_the_launcher_takes_two_lanes_from_the_measured_size_on(W, want):
    """capi.hip: kDuoMinWalkers = 49 152 (profiles/r06_duo.txt): below, a two-mode mixture runs with four
    lanes per walker, from there on with two -- either way bit for bit the oracle's walkers."""
    d, gs, bgs = 30, 256, 1024
    eng, prob, st, mean, cov = _pair(d, W, gs, bgs, 0, K=2)
    for n in (1, 75):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=O.max_threads())
        _compare(eng, st, f"{W} walkers")
    assert want in eng.last_step_kernel(), eng.last_step_kernel()
    eng.close()
