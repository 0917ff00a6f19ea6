"""GPU parity (Tier B): the HIP path, called through the C ABI, against the CPU oracle
(oracle/mcmc_oracle.c) on the same seeded inputs -- BIT-EXACT for the walker state, the
log-posterior values, integer weights, accept counts and emitted rows; and against the golden
vectors generated from the reference (G4/G5) for the batch evaluator."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cobaya_amd import engine as E  # noqa: E402
from oracle import cbind as O  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    bad = bits(a) != bits(b)
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} values differ; first "
                           f"{a[bad][:3]} vs {b[bad][:3]}")


def random_target(d, K, rng, spread=0.05):
    means = rng.uniform(0.35, 0.65, size=(K, d))
    covs = []
    for _ in range(K):
        A = rng.normal(size=(d, d))
        c = A @ A.T / d + np.eye(d) * 0.5
        s = rng.uniform(0.5, 1.5, size=d) * spread
        covs.append(c * np.outer(s, s))
    return means, np.array(covs)


def make_pair(d, W, gs, K=1, kinds=None, a=None, b=None, periodic=None, seed=7, T=1.0,
              burn_in=0, cap=0, weights=None, normalized=True, rng=None, walker_offset=0,
              max_tries=None, blocks=None, over=None, drag_last_slow=-1, drag_steps=0,
              own_constants=False, incremental=False, shared_basis=True, basis_group_size=None):
    """own_constants=False hands the oracle the constants the engine derived on the host (T,
    L^-1, log-normalisations), so that the comparison isolates the KERNELS, bit for bit;
    own_constants=True lets the oracle derive them itself with the numpy recipe (the
    reference's arithmetic) -- see test_steps_with_the_oracles_own_constants."""
    rng = rng or np.random.default_rng(100 + d)
    kinds = [0] * d if kinds is None else kinds
    a = [0.0] * d if a is None else a
    b = [1.0] * d if b is None else b
    eng = E.Engine(d, W, group_size=gs, seed=seed, temperature=T, burn_in=burn_in,
                   emit_capacity=cap, walker_offset=walker_offset, max_tries=max_tries,
                   incremental=incremental, shared_basis=shared_basis,
                   basis_group_size=basis_group_size)
    eng.set_prior(kinds, a, b, periodic)
    if K == 0:
        eng.set_target_one()
        means = covs = None
    else:
        means, covs = random_target(d, K, rng)
        if K == 1 and not normalized:
            eng.set_target_gaussian(means[0], covs[0], normalized=False)
        else:
            eng.set_target_gaussian_mixture(means, covs, weights)
    pcov = (covs[0] if K else np.diag(np.full(d, 0.01))) * T
    if blocks is not None:
        eng.set_blocking(blocks, over, drag_last_slow, drag_steps)
    eng.set_proposal_cov(pcov)
    if blocks is not None:  # the engine's transform is the oracle's recipe in sorted order
        np.testing.assert_allclose(eng.get_proposal_transform(),
                                   O.blocked_transform(pcov, blocks, 2.4), rtol=1e-12, atol=1e-15)
    if own_constants:
        T_orc = (O.blocked_transform(pcov, blocks, 2.4) if blocks is not None
                 else O.proposal_transform(pcov, 2.4))
    else:
        T_orc = eng.get_proposal_transform()
    prob = O.Problem(d, kinds, a, b, periodic=periodic, means=means, covs=covs,
                     weights=weights, normalized=normalized, T=T_orc,
                     blocks=blocks, oversampling=over, drag_last_slow=drag_last_slow,
                     drag_steps=drag_steps,
                     group_size=(basis_group_size or gs) if (shared_basis or d == 1) else 1,
                     seed=seed,
                     temperature=T, max_tries=max_tries,
                     derived=None if own_constants else eng.derived_constants(),
                     incremental=incremental,
                     # (which mixtures carry the log-density of every mode is the engine's rule)
                     carry_modes=eng.carries_modes(), carry_periodic=eng.carries_periodic())
    m0 = means[0] if K else np.full(d, 0.5)
    s0 = np.sqrt(np.diag(covs[0])) if K else np.full(d, 0.1)
    x0 = np.clip(m0 + rng.normal(size=(W, d)) * s0, 1e-3, 1 - 1e-3)
    # (intervals narrower than [0, 1]: the walkers start inside them)
    av, bv = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    uni = np.asarray(kinds) == 0
    if periodic is not None:   # (a periodic parameter is wrapped into its interval below)
        uni &= np.asarray(periodic) == 0
    x0[:, uni] = np.clip(x0[:, uni], np.maximum(1e-3, av + 1e-3 * (bv - av))[uni],
                         np.minimum(1 - 1e-3, bv - 1e-3 * (bv - av))[uni])
    if kinds is not None:
        for i, k in enumerate(kinds):
            if k == 1:
                x0[:, i] = a[i] + rng.normal(size=W) * b[i]
    if periodic is not None:   # a periodic parameter starts inside its interval
        for i, per in enumerate(periodic):
            if per and (a[i], b[i]) != (0.0, 1.0):
                x0[:, i] = a[i] + (x0[:, i] - a[i]) % (b[i] - a[i])
    eng.set_state(x0)
    st = O.State(prob, x0, burn_in=burn_in, row_cap=cap)
    return eng, prob, st


def compare_state(eng, st):
    s = eng.get_state()
    assert_bit_equal(s["x"], st.x, "x")
    assert_bit_equal(s["logpost"], st.logpost, "logpost")
    assert_bit_equal(s["logprior"], st.logprior, "logprior")
    assert_bit_equal(s["loglike"], st.loglike, "loglike")
    assert np.array_equal(s["weight"], st.weight)
    if eng.incremental and eng.carries_modes() and st.step > 0:
        assert st.p.c.carry_modes == 1
        assert_bit_equal(eng.get_full_state()["amode"], st.amode, "carried mode log-densities")


def test_library_reports_gfx950():
    lib = E.load_library()
    assert b"gfx950" in lib.mcmc_hip_version()


def test_initial_evaluation_bit_exact():
    eng, prob, st = make_pair(30, 256, 64)
    compare_state(eng, st)


@pytest.mark.parametrize("d,W,gs,K,steps", [
    (30, 512, 64, 1, 95), (2, 256, 64, 1, 41), (3, 256, 128, 1, 50), (30, 512, 256, 1, 61),
    (4, 256, 64, 2, 60), (30, 256, 64, 3, 45), (5, 256, 64, 0, 40), (1, 256, 64, 1, 30),
    (27, 256, 64, 1, 60), (32, 256, 64, 1, 40),
    # d >= 14 with W % 256 == 0 runs the two-waves-per-walker-set kernel (every row split);
    # smaller d and other widths run the one-wave kernel
    (8, 256, 64, 1, 50), (13, 256, 128, 1, 45), (14, 256, 64, 1, 45), (16, 512, 256, 1, 40),
    (17, 256, 128, 1, 40), (21, 256, 64, 1, 50),
    (24, 256, 256, 1, 55), (30, 192, 64, 1, 70), (16, 64, 64, 1, 40),
    # more than 16 modes (kMaxModes 64 since round 5): the mode log-densities of a workgroup in LDS
    (6, 128, 64, 24, 40), (12, 256, 128, 64, 30)])
def test_steps_bit_exact(d, W, gs, K, steps):
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.2, 0.8] if K == 2 else None)
    # several launches that start and stop mid-cycle
    for n in (1, steps // 3, steps - steps // 3 - 1):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
    c = eng.counters()
    assert c["steps"] == steps and c["accepted"] == int(st.n_accept.sum())
    assert 0.05 < c["accepted"] / (W * steps) < 0.9


@pytest.mark.parametrize("d,W,gs,steps", [(33, 256, 64, 70), (48, 256, 128, 60),
                                          (50, 256, 64, 60), (56, 512, 128, 60),
                                          (70, 256, 64, 75), (88, 256, 128, 50),
                                          (93, 256, 64, 40), (96, 512, 64, 40),
                                          (64, 512, 64, 70), (100, 256, 64, 130),
                                          (112, 256, 128, 40), (120, 256, 64, 30),
                                          (128, 512, 128, 35), (100, 512, 256, 30),
                                          (128, 256, 256, 20)])
def test_big_dimension_steps_bit_exact(d, W, gs, steps):
    """32 < d <= 112 (BASELINE config 4 is d = 100): the column-sweep kernels against the
    oracle, bit for bit, across launches that start and stop mid-cycle."""
    eng, prob, st = make_pair(d, W, gs, rng=np.random.default_rng(1000 + d))
    compare_state(eng, st)
    for n in (1, steps // 2, steps - steps // 2 - 1):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
    c = eng.counters()
    assert c["accepted"] == int(st.n_accept.sum())
    assert 0.05 < c["accepted"] / (W * steps) < 0.9
    shift = st.x.mean(0)
    eng2, prob2, st2 = make_pair(d, W, gs, rng=np.random.default_rng(1000 + d))
    eng2.set_moment_shift(shift)
    eng2.step(d)
    eng2.accumulate_moments()
    st2.run(d, n_threads=8)
    gsum, S = O.moments(st2.x, gs, shift=shift)
    n, g_gs, g_S = eng2.read_moments()
    assert_bit_equal(g_gs, gsum, "group sums")
    assert_bit_equal(g_S, S, "pooled second moments")


@pytest.mark.parametrize("d,W,gs", [(40, 256, 64), (100, 512, 128), (112, 256, 128)])
def test_big_dimension_normal_priors_bit_exact(d, W, gs):
    """d > 32 with normal priors on the matrix-core kernel: the prior sum is formed as four
    interleaved chains (dimension i in lane class i mod 4), like chi2."""
    rng = np.random.default_rng(900 + d)
    kinds = (rng.random(d) < 0.6).astype(int).tolist()
    a = [0.5 if k else 0.0 for k in kinds]
    b = [float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds]
    eng, prob, st = make_pair(d, W, gs, kinds=kinds, a=a, b=b, T=1.5 if d == 40 else 1.0)
    compare_state(eng, st)
    for n in (3, d + 5, 17):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)


@pytest.mark.parametrize("d,W,gs,normal", [(33, 256, 128, False), (34, 512, 256, True),
                                            (35, 256, 64, False), (36, 512, 128, True),
                                            (37, 256, 256, False), (38, 256, 128, True),
                                            (39, 512, 256, False), (40, 512, 128, True),
                                            (40, 256, 256, False), (43, 256, 256, True),
                                            (44, 512, 128, False), (47, 512, 256, False),
                                            (48, 256, 256, True), (48, 512, 256, False),
                                            (49, 256, 256, False), (50, 512, 256, True),
                                            (52, 256, 256, False), (53, 512, 256, True),
                                            (55, 256, 256, False), (56, 512, 256, True),
                                            (56, 256, 256, False)])
def test_two_wave_kernel_above_32_dimensions_bit_exact(d, W, gs, normal):
    """32 < d <= 56 on whole 256-walker workgroups: the two-wave step kernel (compiled per
    dimension) with the d > 32 sums -- four interleaved chi2 chains handed from one wave to the
    other, four chains of normal-prior terms -- on the d > 32 layout of V.  The same launches on
    the matrix-core kernel (MCMC_HIP_NO_PAIR_BIG) give the same bits."""
    rng = np.random.default_rng(3300 + d)
    kw = {}
    if normal:
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kinds[d - 1] = 1                       # a normal prior beyond bit 31 of the mask
        kw = dict(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    eng, prob, st = make_pair(d, W, gs, rng=np.random.default_rng(d), T=1.0 if d % 2 else 1.7, **kw)
    compare_state(eng, st)
    for n in (1, d + 3, 2 * d - 1, 9):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    os.environ["MCMC_HIP_NO_PAIR_BIG"] = "1"
    try:
        eng2, prob2, st2 = make_pair(d, W, gs, rng=np.random.default_rng(d),
                                     T=1.0 if d % 2 else 1.7, **kw)
        eng2.step(3 * d + 12)
        eng2.sync()
    finally:
        del os.environ["MCMC_HIP_NO_PAIR_BIG"]
    compare_state(eng2, st)


@pytest.mark.parametrize("d,W,gs,K,case", [
    (40, 256, 64, 2, "mixture"), (64, 128, 64, 3, "mixture"), (36, 128, 128, 0, "one"),
    (50, 128, 64, 1, "periodic+normal"), (100, 256, 128, 1, "periodic"),
    (40, 64, 64, 1, "normal, odd ensemble"), (120, 128, 64, 1, "d > 112, odd ensemble"),
    (48, 256, 256, 2, "mixture+periodic+normal+T"), (40, 128, 64, 33, "mixture")])
def test_big_dimension_general_kernel_bit_exact(d, W, gs, K, case):
    """d > 32 outside the specialised kernels -- mixtures, `one`, periodic parameters, normal
    priors or d > 112 on ensembles that are not whole 256-walker workgroups -- runs on the
    general kernel (general_kernels.hip), bit for bit like the oracle."""
    rng = np.random.default_rng(7000 + d)
    kw = {}
    if "normal" in case:
        kinds = (rng.random(d) < 0.4).astype(int).tolist()
        kinds[d - 1] = 1
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    if "periodic" in case:
        per = [0] * d
        for i in (1, 33, d - 2):
            if not kw.get("kinds", [0] * d)[i]:
                per[i] = 1
        kw["periodic"] = per
    if "+T" in case:
        kw["T"] = 1.8
    eng, prob, st = make_pair(d, W, gs, K=K, rng=np.random.default_rng(d), burn_in=2, **kw)
    compare_state(eng, st)
    for n in (1, d + 2, 17):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
    assert eng.counters()["accepted"] == int(st.n_accept.sum())


def test_big_dimension_emitted_rows_bit_exact():
    eng, prob, st = make_pair(40, 128, 64, burn_in=1, cap=48)
    for n in (30, 45):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        rows, ref = eng.drain_samples(), st.drain()
        assert rows.shape == ref.shape and len(rows) > 128
        assert_bit_equal(rows, ref, "rows")
    assert eng.counters()["dropped_rows"] == 0
    with pytest.raises(E.EngineError):
        E.Engine(129, 64)


def test_general_priors_periodic_temperature_bit_exact():
    d = 6
    kinds = [0, 1, 0, 1, 0, 0]
    a = [0.0, 0.5, 0.0, 0.45, -1.0, 0.0]
    b = [1.0, 0.2, 1.0, 0.02, 2.0, 1.0]
    periodic = [0, 0, 1, 0, 0, 0]
    eng, prob, st = make_pair(d, 256, 64, kinds=kinds, a=a, b=b, periodic=periodic, T=2.0,
                              burn_in=3)
    eng.step(77)
    eng.sync()
    st.run(77, n_threads=4)
    compare_state(eng, st)


def test_emitted_rows_and_burn_in_bit_exact():
    eng, prob, st = make_pair(3, 128, 64, burn_in=2, cap=64)
    for n in (40, 37):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=2)
        rows, ref = eng.drain_samples(), st.drain()
        assert rows.shape == ref.shape and len(rows) > 128
        assert_bit_equal(rows, ref, "rows")
    assert eng.counters()["dropped_rows"] == 0
    # weights are multiplicities: per walker they sum to the steps spent at emitted points
    assert rows[:, 1].min() >= 1


@pytest.mark.parametrize("d,W,gs,K,blocks,over,steps", [
    # a one-parameter block (RandProposer1D variates) -> general kernel
    (7, 256, 64, 1, [[5, 0], [3], [1, 6, 2, 4]], [1, 2, 3], 100),
    # hot kernels: two waves per walker set (W % 256 == 0) and one wave (W = 192)
    (30, 512, 256, 1, [list(range(10, 30)), list(range(10))], [1, 3], 130),
    (12, 192, 64, 1, [[11, 3, 7, 1, 9, 5], [0, 2, 4, 6, 8, 10]], [2, 1], 75),
    # mixture target, three blocks, permuted order
    (5, 256, 128, 2, [[4, 1], [0, 3], [2]], [1, 1, 5], 80),
    # blocks without oversampling still change the proposal (block-wise directions)
    (8, 256, 64, 1, [[0, 1, 2], [3, 4, 5, 6, 7]], [1, 1], 60)])
def test_blocked_oversampled_steps_bit_exact(d, W, gs, K, blocks, over, steps):
    """(f)1: parameter blocks with oversampling (proposal.py:96-224): the slot shuffle, the
    per-block Haar bases and the steps equal the oracle's bit for bit."""
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.3, 0.7] if K == 2 else None,
                              blocks=blocks, over=over)
    assert eng.cycle_length() == prob.cycle_length() == sum(
        o * len(b) for o, b in zip(over, blocks))
    for n in (1, steps // 3, steps - steps // 3 - 1):   # launches start and stop mid-cycle
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
    c = eng.counters()
    assert c["steps"] == steps and c["accepted"] == int(st.n_accept.sum())


@pytest.mark.parametrize("d,W,gs,K,blocks,last_slow,n_drag,steps,extra", [
    (5, 256, 64, 1, [[0, 1], [2, 3, 4]], 0, 6, 40, {}),
    # one-parameter blocks on both sides, mixture target, two slow blocks
    (7, 256, 128, 2, [[5], [0, 3], [1], [6, 2, 4]], 1, 3, 45, {}),
    (30, 256, 256, 1, [list(range(10)), list(range(10, 30))], 0, 4, 25, {}),
    # normal priors, a periodic parameter, temperature, burn-in
    (6, 128, 64, 1, [[0, 1, 2], [3, 4, 5]], 0, 5, 40,
     dict(kinds=[0, 1, 0, 1, 0, 0], a=[0.0, 0.5, 0.0, 0.5, 0.0, 0.0],
          b=[1.0, 0.2, 1.0, 0.3, 1.0, 1.0], periodic=[0, 0, 1, 0, 0, 1], T=1.7, burn_in=3))])
def test_dragging_steps_bit_exact(d, W, gs, K, blocks, last_slow, n_drag, steps, extra):
    """(f)1: the dragging step (mcmc.py:564-668) against the oracle's drag_core, bit for bit:
    slow and fast direction sequences, interpolation steps, final averaged test, bookkeeping."""
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.4, 0.6] if K == 2 else None,
                              blocks=blocks, over=[1] * (last_slow + 1) + [2] * (len(blocks) - last_slow - 1),
                              drag_last_slow=last_slow, drag_steps=n_drag, **extra)
    n_slow = sum(len(b) for b in blocks[:last_slow + 1])
    assert eng.cycle_length() == prob.cycle_length(1) == n_slow
    for n in (1, steps // 3, steps - steps // 3 - 1):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
    c = eng.counters()
    assert c["steps"] == steps and c["accepted"] == int(st.n_accept.sum())
    assert c["accepted"] > 0.05 * W * steps


@pytest.mark.parametrize("d,W,gs,K,blocks,last_slow,n_drag,extra", [
    (5, 256, 64, 1, [[0, 1], [2, 3, 4]], 0, 6, dict(burn_in=2)),
    (7, 256, 128, 2, [[5], [0, 3], [1], [6, 2, 4]], 1, 3, {}),
    (6, 128, 64, 1, [[0, 1, 2], [3, 4, 5]], 0, 5,
     dict(kinds=[0, 1, 0, 1, 0, 0], a=[0.0, 0.5, 0.0, 0.5, 0.0, 0.0],
          b=[1.0, 0.2, 1.0, 0.3, 1.0, 1.0], periodic=[0, 0, 1, 0, 0, 1], T=1.7))])
def test_dragging_steps_emit_rows_bit_exact(d, W, gs, K, blocks, last_slow, n_drag, extra):
    """VERDICT r3 missing 3: every dragging step ends in process_accept_or_reject
    (mcmc.py:656-668), so with `emit_capacity > 0` the point a walker leaves goes out with its
    weight -- the rows of drag_kernel equal the oracle's (drag_core -> commit) bit for bit,
    burn-in included, and per walker the weights of the emitted rows plus the open weight add
    up to the steps taken."""
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.4, 0.6] if K == 2 else None,
                              blocks=blocks, over=[1] * (last_slow + 1) + [2] * (len(blocks) - last_slow - 1),
                              drag_last_slow=last_slow, drag_steps=n_drag, cap=48, **extra)
    total = np.zeros(W)
    for n in (1, 30, 17):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
        rows, ref = eng.drain_samples(), st.drain()
        assert rows.shape == ref.shape
        assert_bit_equal(rows, ref, "rows")
        np.add.at(total, rows[:, 0].astype(int), rows[:, 1])
    assert len(rows) > W // 4 and eng.counters()["dropped_rows"] == 0
    # a walker's emitted weights + the weight of its open point = the initial 1 + the 48 steps,
    # minus the weights of the points dropped instead of emitted: the initial point (a fresh run
    # never stores it, mcmc.py:265 `+ int(resuming is False)`) and those left during the burn-in
    full = eng.get_full_state()
    spent = total + full["weight"]
    moved = full["n_accept"] > 0
    assert np.all(spent[~moved] == 49) and np.all(spent[moved] <= 48) and np.all(spent >= 1)
    if not extra.get("burn_in"):
        assert np.max(spent[moved]) == 48      # (a walker whose first step was accepted)


@pytest.mark.parametrize("d,W,gs,K,blocks,over,extra", [
    (40, 128, 64, 1, [list(range(25)), list(range(25, 40))], [1, 3], {}),
    # a one-parameter block, a mixture, a periodic parameter, normal priors, emitted rows
    (36, 128, 64, 2, [list(range(5, 36)), [2], [0, 1, 3, 4]], [1, 2, 4],
     dict(kinds=[0] * 30 + [1] * 6, a=[0.0] * 30 + [0.5] * 6, b=[1.0] * 30 + [0.3] * 6,
          periodic=[0, 0, 0, 1] + [0] * 32, cap=40, burn_in=2)),
    (100, 64, 64, 1, [list(range(60)), list(range(60, 100))], [1, 2], {})])
def test_blocked_steps_above_d32_from_scratch_bit_exact(d, W, gs, K, blocks, over, extra):
    """VERDICT r3 missing 4: parameter blocks / oversampling from scratch above d = 32
    (step_general_kernel reading the blocked directions): bit for bit the oracle's."""
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.3, 0.7] if K == 2 else None,
                              blocks=blocks, over=over, **extra)
    for n in (1, 21, 18):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
        if extra.get("cap"):
            assert_bit_equal(eng.drain_samples(), st.drain(), "rows")
    assert eng.last_step_kernel().startswith("mcmc::step_general_kernel")
    assert eng.counters()["accepted"] == int(st.n_accept.sum()) > 0


@pytest.mark.parametrize("d,W,gs,K,blocks,last_slow,n_drag,extra", [
    (40, 128, 64, 1, [list(range(12)), list(range(12, 40))], 0, 4, {}),
    (36, 128, 64, 2, [[35], list(range(5, 35)), [2], [0, 1, 3, 4]], 1, 3,
     dict(kinds=[0] * 30 + [1] * 6, a=[0.0] * 30 + [0.5] * 6, b=[1.0] * 30 + [0.3] * 6,
          periodic=[0, 0, 0, 1] + [0] * 16 + [1] + [0] * 15, cap=24, burn_in=1, T=1.3)),
    (72, 64, 64, 1, [list(range(30)), list(range(30, 72))], 0, 2, dict(cap=12))])
def test_dragging_above_d32_from_scratch_bit_exact(d, W, gs, K, blocks, last_slow, n_drag, extra):
    """VERDICT r3 missing 4: the dragging step from scratch above d = 32 (drag_general_kernel):
    mixtures, periodic parameters (the DELTA of an interpolation step is wrapped, mcmc.py:606),
    one-parameter blocks, emitted rows -- the oracle's drag_core bit for bit."""
    eng, prob, st = make_pair(d, W, gs, K=K, weights=[0.4, 0.6] if K == 2 else None,
                              blocks=blocks, over=[1] * (last_slow + 1) + [2] * (len(blocks) - last_slow - 1),
                              drag_last_slow=last_slow, drag_steps=n_drag, **extra)
    for n in (1, 11, 9):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
        if extra.get("cap"):
            assert_bit_equal(eng.drain_samples(), st.drain(), "rows")
    assert eng.last_step_kernel().startswith("mcmc::drag_general_kernel")
    assert eng.counters()["accepted"] == int(st.n_accept.sum()) > 0


def test_blocking_errors_are_loud():
    eng = E.Engine(4, 64, group_size=64, seed=1)
    with pytest.raises(E.EngineError, match="do not contain all"):
        eng.set_blocking([[0, 1], [1, 3]], [1, 2])
    with pytest.raises(E.EngineError, match="Oversampling factors"):
        eng.set_blocking([[0, 1], [2, 3]], [1, 0])


def test_paired_kernel_with_temperature_bit_exact():
    """Hot variant, two waves per walker set, T != 1 (the division by T stays in)."""
    eng, prob, st = make_pair(30, 512, 128, T=2.5)
    for n in (7, 40, 33):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)


def test_walker_offset_shards_reproduce_the_whole():
    """Multi-GPU sharding: walkers [256, 512) run as a shard with
    walker_offset=256 are bit-identical to the same walkers inside a 512-walker ensemble."""
    full, _, _ = make_pair(8, 512, 64, rng=np.random.default_rng(5))
    x0 = full.get_state()["x"]
    full.step(50)
    full.sync()
    part, _, _ = make_pair(8, 256, 64, rng=np.random.default_rng(5), walker_offset=256)
    part.set_state(x0[256:])
    part.step(50)
    part.sync()
    assert_bit_equal(part.get_state()["x"], full.get_state()["x"][256:], "shard")


def test_moments_bit_exact():
    eng, prob, st = make_pair(30, 512, 64)
    shift = st.x.mean(0)
    eng.set_moment_shift(shift)
    gs = S = None
    for _ in range(3):
        eng.step(30)
        eng.accumulate_moments()
        st.run(30, n_threads=4)
        gs, S = O.moments(st.x, 64, shift=shift, group_sum=gs, pooled=S)
    n, g_gs, g_S = eng.read_moments(reset=True)
    assert n == 3
    assert_bit_equal(g_gs, gs, "group sums")
    assert_bit_equal(g_S, S, "pooled second moments")
    n, g_gs, g_S = eng.read_moments()
    assert n == 0 and not g_gs.any() and not g_S.any()


def test_evaluator_against_reference_goldens(golden):
    """model.logposterior parity: device evaluator vs values produced by the reference."""
    g = golden("g5_loglike")
    for tag in ("gm_d2_K1", "gm_d3_K3", "gm_d4_K2", "gm_d30_K1", "gm_d30_K3", "gm_d100_K1"):
        means = g[tag + "_means"]
        K, d = means.shape
        eng = E.Engine(d, 64, group_size=64)
        eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
        eng.set_target_gaussian_mixture(means, g[tag + "_covs"],
                                        g[tag + "_weights"] if K > 1 else None)
        lp, ll, der = eng.evaluate(g[tag + "_points"], derived=True)
        assert np.all(lp == 0.0)
        np.testing.assert_allclose(ll, g[tag + "_loglike"], rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(der, g[tag + "_derived"], rtol=1e-9, atol=1e-10)
    for tag in ("gauss_d3_norm1", "gauss_d27_norm1", "gauss_d27_norm0"):
        mean = g[tag + "_mean"]
        d = len(mean)
        eng = E.Engine(d, 64, group_size=64)
        eng.set_prior([0] * d, [-10.0] * d, [10.0] * d)
        eng.set_target_gaussian(mean, g[tag + "_cov"], normalized=tag.endswith("1"))
        lp, ll = eng.evaluate(g[tag + "_points"])
        np.testing.assert_allclose(ll, g[tag + "_loglike"], rtol=1e-12, atol=1e-11)
    g4 = golden("g4_prior")
    kinds = g4["kinds"]
    a = np.where(kinds == 0, g4["bounds"][:, 0], g4["loc"])
    b = np.where(kinds == 0, g4["bounds"][:, 1], g4["scale"])
    eng = E.Engine(5, 64, group_size=64)
    eng.set_prior(kinds, a, b, g4["periodic"])
    eng.set_target_one()
    lp, ll = eng.evaluate(g4["points"])
    ref = g4["logprior"]
    assert np.array_equal(np.isinf(lp), np.isinf(ref))
    m = ~np.isinf(ref)
    np.testing.assert_allclose(lp[m], ref[m], rtol=4e-16)


def test_errors_are_loud():
    with pytest.raises(E.EngineError):
        E.Engine(30, 100, group_size=64)  # not a multiple of the group size
    eng = E.Engine(3, 64)
    eng.set_prior([0] * 3, [0.0] * 3, [1.0] * 3)
    with pytest.raises(E.NotPositiveDefinite):
        eng.set_target_gaussian_mixture([[0.5] * 3], [np.array([[1, 2, 0], [2, 1, 0],
                                                                 [0, 0, 1.0]])])
    eng.set_target_one()
    with pytest.raises(E.NotPositiveDefinite):
        eng.set_proposal_cov(np.array([[1, 2, 0], [2, 1, 0], [0, 0, 1.0]]))
    with pytest.raises(E.EngineError):
        eng.step(1)  # no state yet
    with pytest.raises(E.EngineError):
        eng.set_state(np.full((64, 3), 2.0))  # outside the prior: non-finite posterior


def test_stuck_chain_is_reported():
    """mcmc.py:717-743: a proposal far too wide trips max_tries."""
    d = 3
    eng = E.Engine(d, 64, max_tries=20)
    eng.set_prior([0] * d, [-50.0] * d, [50.0] * d)
    eng.set_target_gaussian_mixture([[0.0] * d], [np.eye(d) * 1e-6])
    eng.set_proposal_cov(np.eye(d) * 100.0)
    eng.set_state(np.zeros((64, d)))
    eng.step(400)
    with pytest.raises(E.ChainStuck):
        eng.sync()


@pytest.mark.parametrize("mode", ["full", "incremental", "incremental-1024", "incremental-4096"])
def test_posterior_moments_full_size(mode):
    """Tier C at BASELINE config-2 size: 65 536 walkers, d=30 target of the golden fixture;
    mean within 1% of sigma and covariance within 1% (north star) after >= 1e6 accepted --
    with every trial evaluated from scratch, with incremental evaluation, with basis groups of
    1 024 walkers and with the 4 096 the sampler picks at this size (what bench.py runs: 16
    Haar bases for the ensemble)."""
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden",
                                           "targets.npz"))
    mean, cov = g["mean_d30"], g["cov_d30"]
    d, W = 30, 65536
    eng = E.Engine(d, W, group_size=64, seed=1, incremental=mode != "full",
                   basis_group_size=int(mode.split("-")[1]) if "-" in mode else None)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    rng = np.random.default_rng(1)
    x0 = mean + rng.standard_normal((W, d)) * np.sqrt(np.diag(cov))
    eng.set_state(np.clip(x0, 1e-6, 1 - 1e-6))
    eng.step(40 * d)  # burn-in from the (narrower than posterior) ref pdf
    eng.set_moment_shift(mean)
    for _ in range(24):
        eng.step(4 * d)
        eng.accumulate_moments()
    eng.sync()
    n, gs, S = eng.read_moments()
    N = n * W
    m = gs.sum(0) / N
    c = S / N - np.outer(m, m)
    acc = eng.counters()["accepted"]
    assert acc >= 1e6
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(m) / sig) < 0.01           # shift = true mean
    corr_err = (c - cov) / np.outer(sig, sig)
    assert np.max(np.abs(corr_err)) < 0.01
    kl = 0.5 * (np.trace(np.linalg.solve(c, cov)) + m @ np.linalg.solve(c, m) - d
                + np.linalg.slogdet(c)[1] - np.linalg.slogdet(cov)[1])
    assert kl < 0.07 and kl < 1e-3


def test_config5_shape_d27_bit_exact():
    """d = 27, `gaussian` (normalized) likelihood, 6 uniform + 21 normal priors: the GENERAL
    kernel variant against the oracle."""
    d = 27
    rng = np.random.default_rng(27)
    kinds = [0] * 6 + [1] * 21
    a = [0.0] * 6 + list(rng.uniform(0.45, 0.55, 21))
    b = [1.0] * 6 + list(rng.uniform(0.05, 0.2, 21))
    # W = 256: the two-wave kernel with normal priors; W = 192: the general one-wave kernel;
    # T != 1 and mixed positions of the normal priors as well
    for W, T, kk in ((256, 1.0, kinds), (192, 1.0, kinds), (512, 1.3, kinds[::-1])):
        aa = a if kk is kinds else a[::-1]
        bb = b if kk is kinds else b[::-1]
        eng, prob, st = make_pair(d, W, 64, kinds=kk, a=aa, b=bb, T=T,
                                  rng=np.random.default_rng(27))
        for n in (30, 51):
            eng.step(n)
            eng.sync()
            st.run(n, n_threads=4)
            compare_state(eng, st)


# ------------------------------------------------------------------ host-derived constants
# The bit-exact tests above give the oracle the constants the engine derived on the host, so
# they compare kernels, not the host algebra.  The three tests below pin that algebra on its
# own: against the REFERENCE's golden G1, against the numpy recipe, and end to end.
def _g1_blocks(g, name):
    flat, lens = g[name + "_blocks"], g[name + "_blocklens"]
    blocks, k = [], 0
    for n in lens:
        blocks.append([int(i) for i in flat[k:k + n]])
        k += n
    return blocks


@pytest.mark.parametrize("name", ["d2", "d3", "d30", "d100", "d5_2blocks", "d5_3blocks"])
def test_proposal_transform_matches_reference_golden_g1(golden, name):
    """a1: mcmc_hip_set_proposal_cov -> mcmc_hip_get_proposal_transform against the transforms
    `BlockedProposer.set_covariance` produced in the reference (proposal.py:226-260,
    tools.py:761-788; tests/golden/make_golden.py g1_transforms), one-block and blocked form.
    The engine keeps ONE d x d factor in sorted order; the reference's per-block matrix is its
    slice T[j_b:, j_b:j_b+n_b]."""
    g = golden("g1_transforms")
    cov, blocks = g[name + "_cov"], _g1_blocks(g, name)
    d = len(cov)
    eng = E.Engine(d, 256, group_size=64, proposal_scale=1.0)
    eng.set_prior([0] * d, [-100.0] * d, [100.0] * d)
    eng.set_target_one()
    if len(blocks) > 1:
        eng.set_blocking(blocks, [1] * len(blocks))
    eng.set_proposal_cov(cov)
    T = eng.get_proposal_transform()
    assert np.array_equal(np.triu(T, 1), np.zeros_like(T))
    j = 0
    for b, blk in enumerate(blocks):
        ref = g[f"{name}_transform{b}"]
        np.testing.assert_allclose(T[j:, j:j + len(blk)], ref, rtol=1e-14,
                                   atol=1e-15 * np.abs(ref).max())
        j += len(blk)
    np.testing.assert_array_equal(eng.get_proposal_cov(), cov)
    eng.close()


@pytest.mark.parametrize("d", [30, 100])
def test_derived_constants_match_the_numpy_recipe(golden, d):
    """L^-1, the log-normalisation, -log(scale) - log(2 pi)/2 and the uniform log-volume that
    the host side of the library derives, against numpy (the reference's own tools:
    functions.py:81-89 inverse_cholesky, tools.py:720-729 _fast_norm_logpdf, prior.py:514-533),
    on the BASELINE targets.  Scalars to 4 ulp; the triangular inverse entrywise to a few
    ulp of its row scale (two different but backward-stable algorithms)."""
    t = golden("targets")
    mean, cov = t[f"mean_d{d}"], t[f"cov_d{d}"]
    rng = np.random.default_rng(d)
    kinds = (rng.random(d) < 0.4).astype(int)
    a = np.where(kinds == 1, 0.5, -0.25)
    b = np.where(kinds == 1, rng.uniform(0.05, 3.0, d), 1.5)
    eng = E.Engine(d, 256, group_size=64)
    eng.set_prior(kinds, a, b)
    eng.set_target_gaussian_mixture([mean, mean + 0.01], [cov, 2.0 * cov], [0.25, 0.75])
    dc = eng.derived_constants()
    ulp = np.finfo(float).eps
    uni = kinds == 0
    # (a sum of n logs: sequential in the library, pairwise in numpy -- n/2 ulp apart at most)
    assert abs(dc["uniform_logp"] + np.sum(np.log(b[uni] - a[uni]))) <= (4 + uni.sum() / 2) * ulp * abs(
        dc["uniform_logp"])
    mls = -np.log(b[kinds == 1]) - np.log(2 * np.pi) / 2
    np.testing.assert_allclose(dc["mls"][kinds == 1], mls, rtol=4 * ulp, atol=4 * ulp)
    for k, c in enumerate((cov, 2.0 * cov)):
        L = np.linalg.cholesky(c)
        Linv = np.linalg.inv(L)
        scale = np.abs(Linv).max(axis=1, keepdims=True)
        assert np.max(np.abs(dc["Linv"][k] - Linv) / scale) < 64 * d * ulp
        assert np.array_equal(np.triu(dc["Linv"][k], 1), np.zeros((d, d)))
        cn = d * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(L)))
        assert abs(dc["cnorm"][k] - cn) <= 4 * ulp * abs(cn)
    np.testing.assert_allclose(dc["weight"], [0.25, 0.75], rtol=2 * ulp)
    eng.close()


@pytest.mark.parametrize("d,W,gs,K,steps,kw", [
    (8, 256, 64, 1, 40, {}),                                    # one-wave kernel
    (30, 512, 256, 1, 60, {}),                                  # two-wave kernel (config 2)
    (4, 256, 64, 2, 40, {"weights": [0.2, 0.8]}),               # mixture
    (40, 256, 256, 1, 50, {}),                                  # two-wave kernel, d > 32
    (100, 256, 64, 1, 40, {}),                                  # matrix-core kernel (config 4)
    (40, 256, 64, 2, 30, {"weights": [0.5, 0.5]}),              # general kernel
    (9, 256, 64, 1, 40, {"blocks": [[0, 1, 2, 3], [4, 5, 6, 7, 8]], "over": [1, 3]}),
])
def test_steps_with_the_oracles_own_constants(d, W, gs, K, steps, kw):
    """One case per kernel family with NOTHING handed over: the oracle derives T, L^-1 and the
    normalisations itself with numpy (the reference's arithmetic), the engine with its host
    C++.  The constants differ by rounding (see the two tests above), so the trajectories agree
    to rounding as long as no accept decision sits within rounding of its threshold -- at
    these sizes none does: the integer state (weights, accept counts) must be IDENTICAL and the
    positions equal to 1e-11 of the prior width."""
    eng, prob, st = make_pair(d, W, gs, K=K, own_constants=True, **kw)
    eng.step(steps)
    eng.sync()
    st.run(steps, n_threads=8)
    s = eng.get_state()
    assert np.array_equal(s["weight"], st.weight)
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    np.testing.assert_allclose(s["x"], st.x, rtol=0, atol=1e-11)
    np.testing.assert_allclose(s["logpost"], st.logpost, rtol=1e-10, atol=1e-9)
    eng.close()


@pytest.mark.parametrize("incremental,bgs", [(False, None), (True, None), (True, 1024),
                                             (True, 8192)])   # 8192: ONE basis for all
def test_walkers_of_a_group_are_independent_chains(incremental, bgs):
    """The walkers of a group share the Haar basis of every cycle but draw their own sign,
    radial distance and accept variate: given the bases each walker's kernel is symmetric and
    pi-invariant, so at stationarity the walkers are independent and the variance of the
    ensemble mean is sigma^2 / W.  (With a shared sign -- the round-1 specification -- the
    walkers of a group drifted together and this ratio was ~ group_size / 4.)"""
    t = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    mean, cov = t["mean_d30"], t["cov_d30"]
    d, W, gs = 30, 8192, 256
    eng = E.Engine(d, W, group_size=gs, seed=5, incremental=incremental, basis_group_size=bgs)
    eng.set_prior([0] * d, [0.0] * d, [1.0] * d)
    eng.set_target_gaussian_mixture([mean], [cov])
    eng.set_proposal_cov(cov)
    rng = np.random.default_rng(1)
    x0 = np.clip(rng.multivariate_normal(mean, cov, size=W), 1e-9, 1 - 1e-9)
    eng.set_state(x0)
    eng.step(20 * d)
    ms = []
    for _ in range(300):
        eng.step(3 * d)
        ms.append(eng.get_state()["x"].mean(0))
    ratio = np.var(ms, axis=0) / (np.diag(cov) / W)
    assert 0.8 < ratio.mean() < 1.25 and ratio.max() < 1.8, ratio
    eng.close()


# ------------------------------------------------------------------ incremental evaluation
@pytest.mark.parametrize("d,W,gs,normal,T", [
    (2, 256, 64, False, 1.0), (3, 128, 64, True, 1.0), (5, 256, 128, False, 2.0),
    (8, 256, 64, False, 1.0), (13, 192, 64, True, 1.0), (30, 512, 256, False, 1.0),
    (27, 256, 64, True, 1.0), (32, 256, 64, False, 1.0), (33, 256, 64, False, 1.0),
    (48, 256, 128, True, 1.5), (64, 256, 64, False, 1.0), (100, 256, 64, False, 1.0),
    (100, 512, 256, True, 1.0), (112, 128, 64, False, 1.0), (128, 256, 128, False, 1.0),
    # general bounds at the top of the range: one wave per SIMD, everything in registers
    (128, 128, 64, True, 1.0), (124, 128, 64, "bounds differ", 1.0), (128, 128, 64, "bounds differ", 2.0),
    (120, 128, 64, True, 1.0),
    # two waves per SIMD with single-precision copies of the bounds in registers (dq = 14..22)
    (56, 128, 64, "bounds differ", 1.0), (80, 256, 128, "bounds differ", 1.0), (84, 128, 64, True, 1.0),
    (88, 128, 64, "bounds differ", 2.0), (60, 128, 64, True, 1.0),
    # ... and bounds the walkers DO reach (the exact comparisons behind the single-precision test)
    (64, 256, 64, "tight", 1.0), (80, 128, 64, "tight", 1.0), (30, 256, 64, "tight", 1.0),
    (100, 128, 64, "tight", 1.0), (128, 128, 64, "tight", 1.0)])
def test_incremental_steps_bit_exact(d, W, gs, normal, T):
    """MCMC_HIP_FLAG_INCREMENTAL (incremental_kernels.hip) against the oracle's incremental mode
    (oracle/mcmc_oracle.c: step_core_inc, orc_whiten, orc_whiten_directions): positions, the
    carried whitened residual, log-posterior values, weights and accept counts BIT FOR BIT,
    over launches that end mid-cycle, off the four-step variate blocks, and across the refresh
    at 40 d steps."""
    kw = {}
    if normal == "bounds differ":
        kw = dict(a=[-0.25 * (i % 3) for i in range(d)], b=[1.0 + 0.5 * (i % 2) for i in range(d)])
    elif normal == "tight":   # every fifth parameter on an interval a few sigma wide
        kw = dict(a=[0.44 if i % 5 == 0 else 0.0 for i in range(d)],
                  b=[0.56 if i % 5 == 0 else 1.0 + 0.25 * (i % 2) for i in range(d)])
    elif normal:
        rng = np.random.default_rng(7000 + d)
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kinds[d - 1] = 1
        kw = dict(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    eng, prob, st = make_pair(d, W, gs, T=T, incremental=True,
                              rng=np.random.default_rng(6000 + d), **kw)
    compare_state(eng, st)
    R = 40 * d
    for n in (1, 2, 7, d + 3, R - (d + 13) - 1, 5, 2 * d + 1):   # crosses step R after launch 5
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residual")
    assert st.step > R
    c = eng.counters()
    assert c["steps"] == st.step and c["accepted"] == int(st.n_accept.sum())
    # ("tight": a fifth of the parameters on intervals a few sigma wide -- most trials leave them)
    assert (0.003 if normal == "tight" else 0.03) < c["accepted"] / (W * st.step) < 0.9
    assert "step_inc_kernel" in eng.last_step_kernel()
    if normal == "tight":   # (trials did leave the support: the exact path has run)
        assert int(st.prior_rej.sum()) > 0 or c["accepted"] / (W * st.step) < 0.28
    eng.close()


@pytest.mark.parametrize("d,W,gs,kw", [
    (30, 512, 64, {}), (12, 256, 64, {"blocks": [[0, 1, 2], [3], list(range(4, 12))],
                                      "over": [1, 2, 3]}),
    (9, 256, 64, {"blocks": [[0, 1, 2, 3], [4, 5, 6, 7, 8]], "over": [1, 1],
                  "drag_last_slow": 0, "drag_steps": 3})])
def test_directions_computed_ahead_never_change_the_results(d, W, gs, kw):
    """The directions of the next launch are computed on a second stream while a step kernel
    runs (capi.hip, DirSet), on the guess that the next call is like this one.  Whatever comes
    instead -- the same call (the set is used), another length, a new proposal covariance, a
    restored state -- the states are those of the oracle, bit for bit: a set computed ahead is
    dropped when it does not fit."""
    eng, prob, st = make_pair(d, W, gs, incremental=True, **kw)

    def go(n):
        eng.step(n)
        st.run(n, n_threads=8)

    def check():
        eng.sync()
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residual")

    for _ in range(3):          # equal calls: the second and third use the set computed ahead
        go(17)
    check()
    go(5)                       # another length: recomputed in line
    go(17)
    check()
    # a proposal refresh between two equal calls (what a learn checkpoint does): the set computed
    # ahead with the old transform must not be used
    cov2 = eng.get_proposal_cov() * 1.7 + np.diag(np.full(d, 1e-5))
    eng.set_proposal_cov(cov2)
    prob.set_T(eng.get_proposal_transform())
    go(17)
    go(17)
    check()
    # a restored state: the step counter jumps back
    snap = eng.get_full_state()
    snap_o = {k: getattr(st, k).copy() for k in ("x", "y", "logpost", "logprior", "loglike",
                                                 "weight", "prior_rej", "burn_left", "n_accept")}
    step_o = st.step
    go(17)
    go(17)
    eng.set_full_state(snap)
    for k, v in snap_o.items():
        getattr(st, k)[...] = v
    st.step = step_o
    go(17)
    check()
    go(2 * 40 * eng.cycle_length() + 3)     # across refreshes: several launches in one call
    check()
    eng.close()


@pytest.mark.parametrize("which,bound", [(0, 24), (1, 28)])
def test_short_variates_are_redrawn_in_their_lowest_bin(which, bound):
    """VERDICT r3 item 9 (law hardening): a paired variate whose 24-bit (radial) / 28-bit (accept)
    uniform falls into the lowest bin is redrawn there at full width (det_math.h `pair_tail`,
    oracle `pair_tail`): the exponential laws keep their exact tails beyond 24 ln 2 / 28 ln 2.
    The oracle finds such a draw (2^-24 / 2^-28 per step and walker); an ensemble that contains
    that walker runs through that step on the device, bit for bit the oracle's."""
    seed = 7
    hit = O.find_short_tail(seed, 0, 1 << 17, 0, 4096, which)
    assert hit is not None
    gid, step = hit
    r, Ea = O.pair_variates(seed, gid, step)
    # |r| = E on the exponential branch, sqrt(2 E) on the chi(2) branch (proposal.py:79-82)
    assert (max(abs(r), r * r / 2.0) if which == 0 else Ea) > bound * np.log(2.0)
    off = gid - gid % 64
    eng, prob, st = make_pair(4, 64, 64, seed=seed, incremental=True, walker_offset=off)
    for n in (step - 3, 8):          # the second launch crosses the redrawn step
        eng.step(n)
        eng.sync()
        st.run(n, walker0=off, n_threads=4)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residual")


def test_incremental_mode_refuses_what_it_does_not_cover():
    with pytest.raises(E.EngineError, match="incremental"):
        E.Engine(1, 256, group_size=64, incremental=True)
    emi = E.Engine(4, 256, group_size=64, incremental=True, emit_capacity=8)
    emi.set_prior([0] * 4, [0.0] * 4, [1.0] * 4)
    m2, c2 = random_target(4, 1, np.random.default_rng(0))   # rows are emitted by Metropolis steps
    emi.set_target_gaussian_mixture(m2, c2)
    emi.set_blocking([[0, 1], [2, 3]], [1, 1], 0, 3)         # ... not by INCREMENTAL dragging steps
    emi.set_proposal_cov(c2[0])
    emi.set_state(np.full((256, 4), 0.5))
    with pytest.raises(E.EngineError, match="dragging with emitted rows"):
        emi.step(2)
    emi.close()
    eng = E.Engine(128, 256, group_size=64, incremental=True)
    eng.set_prior([0] * 128, [0.0] * 128, [1.0] * 128)
    m, c = random_target(128, 16, np.random.default_rng(0))  # 16 modes at d = 128: 0.5 MB per wave
    eng.set_target_gaussian_mixture(m, c)
    eng.set_proposal_cov(c[0])
    eng.set_state(np.full((256, 128), 0.5))
    with pytest.raises(E.EngineError, match="do not fit the LDS"):
        eng.step(3)
    eng.close()


@pytest.mark.parametrize("d,W,gs,kw", [
    (30, 256, 64, dict()),                                   # MODE 0: one box
    (30, 512, 256, dict(burn_in=3, T=2.0)),
    (27, 256, 64, dict(kinds=[0] * 6 + [1] * 21, a=[0.0] * 6 + [0.5] * 21, b=[1.0] * 6 + [0.3] * 21)),
    (9, 256, 64, dict(a=[0.0] * 4 + [-1.0] * 5, b=[1.0] * 4 + [2.0] * 5)),   # MODE 1: own bounds
    (100, 256, 64, dict()),                                  # two waves per SIMD, read-ahead
    (12, 256, 64, dict(blocks=[[0, 1, 2, 3, 4], [5, 6, 7, 8, 9, 10, 11]], over=[1, 3])),
    # what step_inc_kernel<.., EMIT> leaves out: the general kernels emit at run time --
    (30, 256, 64, dict(K=2, weights=[0.3, 0.7])),            # a mixture (register planes)
    (40, 128, 64, dict(K=6, burn_in=2)),
    (9, 256, 64, dict(blocks=[[4], [0, 1, 2, 3], [5, 6, 7, 8]], over=[1, 2, 2])),   # a 1-D block
    (9, 256, 64, dict(periodic=[0, 0, 1, 0, 0, 0, 0, 0, 0], a=[0.0, 0.0, 0.42] + [0.0] * 6,
                      b=[1.0, 1.0, 0.58] + [1.0] * 6)),      # a periodic parameter (residuals in LDS)
    (30, 128, 64, dict(K=3, periodic=[1] + [0] * 29, a=[0.42] + [0.0] * 29, b=[0.58] + [1.0] * 29))])
def test_incremental_emitted_rows_bit_exact(d, W, gs, kw):
    """`emit: chains` on the incremental path (missing 5): every accepted step past
    the burn-in stores the point it leaves with its weight (mcmc.py:691-707,
    collection.py:402-427); rows, counters and state bit for bit the oracle's, through both
    drains (the copying one and the pinned zero-copy view)."""
    cap = 80
    eng, prob, st = make_pair(d, W, gs, incremental=True, cap=cap, **kw)
    total = 0
    for i, n in enumerate((1, 40, 37)):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        ref = st.drain()
        rows = eng.drain_samples() if i % 2 == 0 else np.array(eng.drain_samples_view())
        assert rows.shape == ref.shape
        assert_bit_equal(rows, ref, "rows")
        total += len(rows)
        compare_state(eng, st)
    assert total > W and eng.counters()["dropped_rows"] == 0
    assert "emit" in eng.last_step_kernel()
    eng.close()


@pytest.mark.parametrize("d,W,gs,thin,kw", [
    (30, 256, 64, 3, {}), (8, 512, 128, 7, dict(burn_in=3, T=2.0)), (100, 128, 64, 2, {}),
    (27, 256, 64, 5, dict(kinds=[0] * 6 + [1] * 21, a=[0.0] * 6 + [0.5] * 21, b=[1.0] * 6 + [0.3] * 21)),
    (12, 256, 64, 4, dict(blocks=[[0, 1, 2, 3, 4], [5, 6, 7, 8, 9, 10, 11]], over=[1, 3])),
    # round 6: the general incremental kernels thin too -- mixtures (register planes), a periodic
    # parameter, a mixture with one, eight modes (LDS state), a block of ONE parameter
    (9, 256, 64, 3, dict(K=2, weights=[0.3, 0.7])),
    (30, 256, 64, 5, dict(K=3)),
    (10, 256, 64, 4, dict(periodic=[1] + [0] * 9)),
    (12, 256, 64, 3, dict(K=2, periodic=[0, 1] + [0] * 10)),
    (16, 128, 64, 2, dict(K=8)),
    (8, 256, 64, 3, dict(blocks=[[0], [1, 2, 3, 4, 5, 6, 7]], over=[1, 2]))])
def test_rows_thinned_on_the_device_bit_exact(d, W, gs, thin, kw):
    """mcmc_hip_set_emit_thin (round 5): step_inc_kernel<.., emit> thins the rows where they are
    produced -- OneSamplePoint.add_to_collection with output_thin (collection.py:1373-1383): the
    weights of a walker add up, a row of weight sum // thin goes out when the sum reaches thin, the
    remainder is carried (and is part of the full state).  Rows, remainders and state bit for bit
    the oracle's; the chain itself is the unthinned one."""
    cap = 80
    eng, prob, st = make_pair(d, W, gs, incremental=True, cap=cap, **kw)
    eng.set_emit_thin(thin)
    st.thin = st.c.thin = thin
    total = weight = 0
    for i, n in enumerate((1, 40, 37, 25)):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        ref = st.drain()
        rows = eng.drain_samples() if i % 2 == 0 else np.array(eng.drain_samples_view())
        assert rows.shape == ref.shape
        assert_bit_equal(rows, ref, "rows")
        total += len(rows)
        weight += int(rows[:, 1].sum()) if len(rows) else 0
        compare_state(eng, st)
        assert np.array_equal(eng.get_thin_carry(), st.thin_acc)
    assert total > W // 2 and eng.counters()["dropped_rows"] == 0
    assert "emit" in eng.last_step_kernel()
    # the carried remainders travel with the full state
    full = eng.get_full_state()
    assert np.array_equal(full["thin_carry"], st.thin_acc)
    eng.set_thin_carry(np.zeros(W, np.int32))
    eng.set_full_state(full)
    assert np.array_equal(eng.get_thin_carry(), st.thin_acc)
    eng.close()


def test_device_thinning_is_refused_where_the_from_scratch_kernels_emit():
    eng, prob, st = make_pair(9, 256, 64, incremental=False, cap=40)
    with pytest.raises(E.EngineError, match="thin on the host"):
        eng.set_emit_thin(3)
    eng.close()
    eng, prob, st = make_pair(9, 256, 64, incremental=True)
    with pytest.raises(E.EngineError, match="emit_capacity"):
        eng.set_emit_thin(3)
    eng.close()


def test_pinned_drain_views_stay_valid_for_the_ring():
    """drain_samples_view hands out library-owned pinned memory: a view is valid until its slot
    comes round again (drain_slots - 1 further drains)."""
    eng, prob, st = make_pair(8, 256, 64, incremental=True, cap=40)
    eng.set_drain_slots(3)
    kept = []
    for _ in range(5):
        eng.step(30)
        v = eng.drain_samples_view()
        assert not v.flags.writeable and len(v) > 0
        kept.append((v, np.array(v)))
        for view, copy in kept[-2:]:        # the last two drains are still readable in place
            assert np.array_equal(view, copy)
    eng.close()


def test_incremental_resume_carries_the_whitened_residual():
    """get_full_state / set_full_state include y: a fresh engine continues bit-identically."""
    eng, prob, st = make_pair(30, 256, 64, incremental=True)
    eng.step(137)
    full = eng.get_full_state()
    eng.step(211)
    ref = eng.get_full_state()
    eng2, _, _ = make_pair(30, 256, 64, incremental=True)
    eng2.set_full_state(full)
    eng2.step(211)
    got = eng2.get_full_state()
    for k in ("x", "y", "logpost", "weight", "n_accept"):
        assert_bit_equal(np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64), k)
    eng.close(), eng2.close()


# ------------------------------------------------------------------ a Haar basis per walker
@pytest.mark.parametrize("d,W,gs,K,extra", [
    (2, 128, 64, 1, {}), (3, 128, 64, 2, {"weights": [0.3, 0.7]}), (8, 192, 64, 1, {}),
    (30, 128, 64, 1, {}), (30, 256, 256, 1, {"T": 2.0}), (5, 128, 64, 0, {}),
    (40, 128, 64, 1, {}), (100, 64, 64, 1, {})])
def test_own_basis_steps_bit_exact(d, W, gs, K, extra):
    """`shared_basis: False` (MCMC_HIP_FLAG_OWN_BASIS): every walker draws its own Haar basis per
    cycle, as every chain of the reference does (proposal.py:59-69).  Against the oracle run
    with one-walker groups -- the basis stream of "group" = global walker id."""
    eng, prob, st = make_pair(d, W, gs, K=K, shared_basis=False, **extra)
    compare_state(eng, st)
    for n in (1, d + 2, 2 * d + 1):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    # (round 6: the tuned d <= 32 step kernel with every walker's own columns from HBM; the general
    # kernel above that)
    assert ("own basis" if d <= 32 else "step_general_kernel") in eng.last_step_kernel()
    # the R-1 groups are still the walker groups
    shift = st.x.mean(0)
    eng.set_moment_shift(shift)
    eng.accumulate_moments()
    gsum, S = O.moments(st.x, gs, shift=shift)
    n, g_gs, g_S = eng.read_moments()
    assert_bit_equal(g_gs, gsum, "group sums")
    eng.close()


def test_own_basis_periodic_rows_and_offsets_bit_exact():
    d, W, gs = 6, 128, 64
    per = [0, 1, 0, 0, 1, 0]
    eng, prob, st = make_pair(d, W, gs, periodic=per, shared_basis=False, cap=40, burn_in=3,
                              walker_offset=4096)
    for n in (9, 20):
        eng.step(n)
        st.run(n, walker0=4096, n_threads=4)
        compare_state(eng, st)
    rows_e, rows_o = eng.drain_samples(), st.drain()
    rows_o[:, 0] += 4096
    assert_bit_equal(rows_e, rows_o, "emitted rows")
    eng.close()


@pytest.mark.parametrize("d,W,gs,K,normal,T", [
    (4, 256, 64, 2, False, 1.0), (30, 256, 64, 2, False, 1.0), (30, 256, 256, 3, False, 1.0),
    (9, 128, 64, 4, True, 1.0), (64, 128, 64, 2, False, 2.0), (27, 256, 128, 2, True, 1.0),
    (30, 256, 64, 2, "box off the origin", 1.0), (30, 256, 64, 4, "bounds differ", 1.0),
    (30, 256, 64, 5, False, 1.0), (28, 128, 64, 6, True, 1.7), (32, 128, 64, 5, True, 1.0), (6, 256, 128, 6, False, 1.0),
    (27, 128, 64, 5, "bounds differ", 1.0),
    # round 6: x in LDS where that buys step_inc_mix_kernel a second wave per SIMD (three modes at
    # d = 49 .. 64, four at d = 41 .. 48)
    (52, 128, 64, 3, False, 1.0), (64, 128, 64, 3, True, 1.0), (44, 128, 64, 4, "bounds differ", 1.0),
    (48, 256, 128, 4, False, 1.7), (60, 64, 64, 3, "box off the origin", 1.0)])
def test_incremental_mixture_steps_bit_exact(d, W, gs, K, normal, T):
    _mixture_case(d, W, gs, K, normal, T, "step_inc_mix_kernel")


@pytest.mark.parametrize("d,W,gs,K,normal,T", [
    (30, 256, 128, 2, False, 1.0), (30, 512, 256, 3, False, 1.0), (24, 256, 128, 4, False, 1.0),
    (16, 256, 128, 4, True, 1.0), (27, 256, 128, 2, True, 1.7), (32, 256, 128, 3, "bounds differ", 1.0),
    (30, 256, 128, 2, "box off the origin", 1.0), (5, 128, 128, 2, False, 1.0), (12, 256, 256, 3, True, 2.0),
    (20, 128, 128, 4, "bounds differ", 1.0), (26, 256, 128, 3, True, 1.0), (2, 128, 128, 4, False, 1.0),
    # two modes above d = 32 (x in LDS; from d = 41 on the v plane is read again at the commit)
    (36, 256, 128, 2, False, 1.0), (40, 128, 128, 2, True, 1.0), (44, 256, 128, 2, "bounds differ", 2.0),
    (48, 128, 128, 2, False, 1.0), (33, 128, 128, 2, "box off the origin", 1.0)])
def test_two_lane_mixture_steps_bit_exact(d, W, gs, K, normal, T, monkeypatch):
    """step_duo_mix_kernel (incremental_duo.hip, round 6): the mixture step with TWO lanes per walker
    -- the layout large ensembles run on (65 536 walkers: tests/test_gpu_bench_geometry.py), forced
    here for small ones (MCMC_HIP_DUO=1) -- bit for bit against the same oracle as the four-lane
    kernel: two and three modes up to d = 32 (x in LDS from d = 25 on with three), four up to
    d = 24; one box, a box off the origin, per-parameter bounds, normal priors, T != 1."""
    monkeypatch.setenv("MCMC_HIP_DUO", "1")
    _mixture_case(d, W, gs, K, normal, T, "step_duo_mix_kernel")


def _mixture_case(d, W, gs, K, normal, T, kernel):
    """Mixtures of 2..4 modes (5 and 6 up to d = 32) in incremental mode (step_inc_mix_kernel): a carried residual and
    a whitened direction per mode, the log-sum-exp of eval_point -- bit for bit against the
    oracle, across the refresh at 40 d steps.  With one box for all dimensions the kernel takes
    the support test on the extremes of the trial (its padded dimensions rest at the middle of
    the box: the box of one case does not contain 0), with per-dimension bounds on lane masks."""
    kw = {}
    if normal == "box off the origin":
        kw = dict(a=[2.0 ** -12] * d, b=[1.5] * d)
    elif normal == "bounds differ":
        kw = dict(a=[-0.25 * (i % 3) for i in range(d)], b=[1.0 + 0.5 * (i % 2) for i in range(d)])
    elif normal:
        rng = np.random.default_rng(7100 + d)
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kw = dict(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    w = np.random.default_rng(d).uniform(0.5, 1.5, K)
    eng, prob, st = make_pair(d, W, gs, K=K, T=T, incremental=True, weights=(w / w.sum()).tolist(),
                              rng=np.random.default_rng(6100 + d), **kw)
    compare_state(eng, st)
    R = 40 * d
    for n in (1, 6, d + 3, R - (d + 10) - 2, 9, d + 1):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
    assert st.step > R and kernel in eng.last_step_kernel()
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    assert eng.carries_modes()
    # the carried a_k are part of the state (mcmc_hip_set_mode_logdensities): handed back with the
    # full state the run continues bit for bit; without them they are re-anchored on y at the next
    # launch -- the same values to rounding
    full = eng.get_full_state()
    assert full["amode"].shape == (W, K)
    eng.set_full_state(full)
    eng.step(d + 5)
    eng.sync()
    st.run(d + 5, n_threads=8)
    compare_state(eng, st)
    full = eng.get_full_state()
    del full["amode"]
    eng.set_full_state(full)
    eng.step(3)
    eng.sync()
    st.run(3, n_threads=8)
    s2 = eng.get_full_state()
    np.testing.assert_allclose(s2["logpost"], st.logpost, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(s2["amode"], st.amode, rtol=1e-12, atol=1e-10)
    eng.close()


@pytest.mark.parametrize("d,W,gs,K,per,extra", [
    (4, 256, 64, 7, [], {}),                                   # more than six modes
    (30, 256, 64, 8, [], {"T": 1.7, "burn_in": 2}),
    (36, 128, 64, 5, [], {}),                                  # five above d = 32
    (10, 128, 64, 16, [], {}),                                 # kMaxModes
    (80, 128, 64, 2, [], {}),                                  # a mixture above d = 64
    (100, 128, 64, 3, [], {"normal": True}),
    (30, 256, 128, 3, [0, 7, 29], {}),                         # periodic parameters and a mixture
    (27, 128, 64, 5, [3], {"normal": True}),
    (40, 128, 64, 1, list(range(0, 40, 2)), {}),               # more than 16 periodic parameters
    (128, 128, 64, 1, list(range(0, 128)), {}),                # ... every one of 128
    (9, 128, 64, 6, [4], {"blocks": [[4], [0, 1, 2, 3], [5, 6, 7, 8]], "over": [1, 2, 2]}),
    (30, 1024, 256, 8, [2], {"bgs": 1024}),
    (9, 128, 64, 7, [], {"blocks": [[4], [0, 1, 2, 3], [5, 6, 7, 8]], "over": [1, 2, 2]}),
    (40, 128, 64, 12, [], {}),                                 # 16 register planes, 12 live
    (6, 128, 64, 24, [], {}),                                  # more than 16 modes: residuals in LDS
    (90, 128, 64, 7, [], {"normal": True})])                   # 8 planes at dq = 23
def test_incremental_general_kernel_steps_bit_exact(d, W, gs, K, per, extra):
    """What the tuned incremental kernels leave out -- more than four modes, mixtures above d = 64,
    periodic parameters with a mixture, more than 16 periodic parameters -- on the general
    incremental kernel (step_inc_any_kernel, residuals in LDS): bit for bit against the oracle's
    step_core_inc, carried residuals included, across the refresh at 40 cycle lengths."""
    extra = dict(extra)
    periodic = [int(i in per) for i in range(d)]
    a = [0.42 if p else 0.0 for p in periodic]
    b = [0.58 if p else 1.0 for p in periodic]
    kinds = [0] * d
    if extra.pop("normal", False):
        rng = np.random.default_rng(7400 + d)
        kinds = [int(not p and rng.random() < 0.5) for p in periodic]
        a = [0.5 if k else v for k, v in zip(kinds, a)]
        b = [float(rng.uniform(0.1, 0.4)) if k else v for k, v in zip(kinds, b)]
    kw = dict(kinds=kinds, a=a, b=b, incremental=True, rng=np.random.default_rng(6400 + d + K))
    if per:
        kw["periodic"] = periodic
    if K > 1:
        wm = np.random.default_rng(d + K).uniform(0.5, 1.5, K)
        kw["weights"] = (wm / wm.sum()).tolist()
    if "bgs" in extra:
        kw["basis_group_size"] = extra.pop("bgs")
    eng, prob, st = make_pair(d, W, gs, K=K, **kw, **extra)
    L = eng.cycle_length()
    compare_state(eng, st)
    wraps = 0
    for n in (1, 6, L + 3, 40 * L - (L + 10) - 2, 9, L + 1):
        before = st.x.copy()
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
        if per:
            wraps += int(np.sum(np.abs(st.x - before)[:, per] > 0.08))
    # (what fits the register file, a few periodic parameters included: step_inc_regs_kernel;
    # 128 periodic parameters: their columns of L^-1 do not fit beside it -- residuals in LDS)
    want = "step_inc_any_kernel" if (len(per) > 100 or K > 16) else "step_inc_regs_kernel"
    assert st.step > 40 * L and want in eng.last_step_kernel(), eng.last_step_kernel()
    assert eng.counters()["accepted"] == int(st.n_accept.sum()) > 0
    if per:
        assert wraps > 20
        x = eng.get_full_state()["x"]
        assert np.all((x[:, per] >= 0.42) & (x[:, per] <= 0.58))
    eng.close()


@pytest.mark.parametrize("d,W,gs,per,extra", [
    (6, 256, 64, [0, 2, 5], {}),
    (30, 256, 128, [0, 7, 29], {"T": 1.7, "burn_in": 2}),
    (27, 128, 64, [3], {"normal": True}),
    (9, 128, 64, [4], {"blocks": [[4], [0, 1, 2, 3], [5, 6, 7, 8]], "over": [1, 2, 2]}),
    (33, 128, 64, [32, 1], {}),
    (56, 128, 64, [55], {}), (64, 128, 64, [3, 40], {}),       # dq = 14..20: single-precision bounds
    (80, 128, 64, [0, 41, 79], {"normal": True}), (72, 256, 128, [5, 6, 7, 8], {}),
    (40, 128, 64, list(range(0, 40, 3)), {}),                  # 14 periodic parameters
    (100, 128, 64, list(range(3, 100, 7)), {"normal": True}),  # 14 at d = 100, carried log-prior
    (100, 128, 64, [5, 50, 99], {})])
def test_incremental_periodic_steps_bit_exact(d, W, gs, per, extra):
    """Periodic parameters in incremental mode (step_inc_kernel<.., periodic>; oracle step_core_inc):
    the coordinate is the wrapped one at every step, and a wrap that changes the winding number
    moves the carried residual by the wrap times a column of L^-1.  The periodic intervals are
    a few sigma wide around the mode, so that walkers cross the seam all the time."""
    extra = dict(extra)
    periodic = [int(i in per) for i in range(d)]
    a = [0.42 if p else 0.0 for p in periodic]
    b = [0.58 if p else 1.0 for p in periodic]
    kinds = [0] * d
    if extra.pop("normal", False):
        rng = np.random.default_rng(7300 + d)
        kinds = [int(not p and rng.random() < 0.5) for p in periodic]
        a = [0.5 if k else v for k, v in zip(kinds, a)]
        b = [float(rng.uniform(0.1, 0.4)) if k else v for k, v in zip(kinds, b)]
    eng, prob, st = make_pair(d, W, gs, kinds=kinds, a=a, b=b, periodic=periodic,
                              incremental=True, rng=np.random.default_rng(6300 + d), **extra)
    L = eng.cycle_length()
    compare_state(eng, st)
    wraps = 0
    for n in (1, 6, L + 3, 40 * L - (L + 10) - 2, 9, L + 1):
        before = st.x.copy()
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
        wraps += int(np.sum(np.abs(st.x - before)[:, per] > 0.08))
    assert st.step > 40 * L and "step_inc_kernel" in eng.last_step_kernel() and "periodic" in eng.last_step_kernel()
    assert wraps > 20 and eng.counters()["accepted"] == int(st.n_accept.sum())
    x = eng.get_full_state()["x"]
    assert np.all((x[:, per] >= 0.42) & (x[:, per] <= 0.58))
    eng.close()


@pytest.mark.parametrize("d,W,gs,K,blocks,over", [
    (9, 256, 64, 1, [[0, 1, 2, 3], [4, 5, 6, 7, 8]], [1, 3]),
    (30, 256, 256, 1, [list(range(10)), list(range(10, 30))], [1, 2]),
    (8, 128, 64, 2, [[6, 1, 3], [0, 2], [4, 5, 7]], [1, 2, 4]),
    (32, 256, 128, 1, [list(range(0, 32, 2)), list(range(1, 32, 2))], [2, 5]),
    # d > 32: the general blocked-direction kernel (blocks of <= 32 parameters project with one
    # chain, larger ones with four)
    (40, 128, 64, 1, [list(range(20)), list(range(20, 40))], [1, 2]),
    (33, 128, 64, 1, [[32, 0, 5], list(range(1, 5)) + list(range(6, 20)), list(range(20, 32))],
     [1, 2, 3]),
    (100, 128, 64, 1, [list(range(0, 100, 5)) + list(range(1, 100, 5)),
                       [i for i in range(100) if i % 5 >= 2]], [1, 2]),
    (48, 128, 64, 2, [list(range(40)), list(range(40, 48))], [1, 4]),
    # one-parameter blocks (RandProposer1D variates on their columns): the 1-D variants
    (4, 256, 64, 1, [[0], [1, 2, 3]], [1, 2]),
    (27, 256, 128, 1, [list(range(6)), list(range(6, 26)), [26]], [1, 2, 4]),
    (7, 128, 64, 1, [[3], [0], [1, 2, 4, 5, 6]], [1, 1, 3]),
    (40, 128, 64, 1, [[39], list(range(20)), list(range(20, 39))], [1, 1, 2]),
    (100, 128, 64, 1, [list(range(50)), [99], list(range(50, 99))], [1, 2, 2]),
    # ... and under a mixture (the flag of the column is a run-time one in step_inc_mix_kernel)
    (9, 128, 64, 2, [[4], [0, 1, 2, 3], [5, 6, 7, 8]], [1, 2, 2]),
    (30, 128, 64, 2, [list(range(10)), [29], list(range(10, 29))], [1, 1, 2]),
    # every block has one parameter: every step draws the 1-D variates
    (3, 128, 64, 1, [[2], [0], [1]], [1, 2, 3]),
    (2, 64, 64, 1, [[1], [0]], [1, 1])])
def test_incremental_blocked_oversampled_steps_bit_exact(d, W, gs, K, blocks, over):
    """Parameter blocks with oversampling (proposal.py:96-260) in incremental mode: a cycle has
    L = sum_b oversample_b n_b columns, each with its whitened image; the refresh falls every
    40 L steps.  Columns of one-parameter blocks draw the RandProposer1D variates
    (proposal.py:85-93) of the un-paired stream."""
    kw = {"weights": [0.3, 0.7]} if K == 2 else {}
    eng, prob, st = make_pair(d, W, gs, K=K, blocks=blocks, over=over, incremental=True, **kw)
    L = eng.cycle_length()
    assert L == sum(o * len(b) for o, b in zip(over, blocks)) and prob.refresh_every == 40 * L
    compare_state(eng, st)
    for n in (1, L + 3, 40 * L - (L + 4) - 3, 7, 2 * L):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
    assert st.step > 40 * L and "step_inc" in eng.last_step_kernel()
    if K == 1:
        assert ("1-D blocks" in eng.last_step_kernel()) == (min(len(b) for b in blocks) == 1)
    eng.close()


@pytest.mark.parametrize("d,W,gs,blocks,last_slow,n_drag,extra", [
    (5, 256, 64, [[0, 1], [2, 3, 4]], 0, 6, {}),
    (30, 256, 256, [list(range(10)), list(range(10, 30))], 0, 4, {}),
    (9, 128, 64, [[0, 1, 2], [3, 4], [5, 6, 7, 8]], 1, 9, {"T": 1.7, "burn_in": 3}),
    # the config-5 shape: 6 slow + 21 fast parameters, normal priors on the fast ones
    (27, 256, 128, [list(range(6)), list(range(6, 27))], 0, 7,
     dict(kinds=[0] * 6 + [1] * 21, a=[0.0] * 6 + [0.5] * 21, b=[1.0] * 6 + [0.25] * 21)),
    # d > 32: the general blocked-direction kernel feeds the slow and the fast sequence
    (40, 128, 64, [list(range(12)), list(range(12, 40))], 0, 5, {}),
    (100, 128, 64, [list(range(30)), list(range(30, 100))], 0, 3, {}),
    # one-parameter blocks among the slow and among the fast ones
    (8, 256, 64, [[0], [1, 2, 3], [4, 5, 6, 7]], 1, 5, {}),
    (27, 256, 128, [list(range(6)), [26], list(range(6, 26))], 0, 7, {"T": 1.3}),
    (40, 128, 64, [[7], list(range(7)), [39], list(range(8, 39))], 1, 4, {}),
    # the fast block is ONE parameter; normal priors and a temperature with 1-D columns
    (6, 128, 64, [[0, 1, 2, 3, 4], [5]], 0, 3, {}),
    (9, 128, 64, [[4], [0, 1, 2, 3], [5, 6, 7, 8]], 1, 2,
     dict(kinds=[0, 1, 0, 1, 1, 0, 0, 1, 0], a=[0, .5, 0, .5, .5, 0, 0, .5, 0],
          b=[1, .2, 1, .3, .25, 1, 1, .2, 1], T=1.6))])
def test_incremental_dragging_steps_bit_exact(d, W, gs, blocks, last_slow, n_drag, extra):
    """The dragging step (mcmc.py:564-668) in incremental mode (drag_inc_kernel against the
    oracle's drag_core_inc): every one of its 1 + 2 n evaluations is O(d), the whitened
    residuals of the start and end points are dragged along; at every d <= 128."""
    eng, prob, st = make_pair(d, W, gs, blocks=blocks,
                              over=[1] * (last_slow + 1) + [2] * (len(blocks) - last_slow - 1),
                              drag_last_slow=last_slow, drag_steps=n_drag, incremental=True, **extra)
    n_slow = sum(len(b) for b in blocks[:last_slow + 1])
    assert eng.cycle_length() == n_slow and prob.refresh_every == 40 * n_slow
    compare_state(eng, st)
    for n in (1, n_slow + 2, 40 * n_slow - (n_slow + 3) - 2, 6, n_slow):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residual")
    assert st.step > 40 * n_slow and "drag_inc_kernel" in eng.last_step_kernel()
    c = eng.counters()
    assert c["accepted"] == int(st.n_accept.sum()) and c["accepted"] > 0.03 * W * st.step
    eng.close()


@pytest.mark.parametrize("d,W,gs,bgs,kw", [
    (30, 2048, 256, 1024, {}), (8, 512, 64, 256, {}), (30, 1024, 64, 512, {"walker_offset": 4096}),
    (9, 512, 64, 512, {"blocks": [[0, 1, 2, 3], [4, 5, 6, 7, 8]], "over": [1, 3]}),
    (12, 512, 128, 256, {"blocks": [[0, 1, 2], [3, 4, 5, 6, 7, 8, 9, 10, 11]], "over": [1, 1],
                         "drag_last_slow": 0, "drag_steps": 4}),
    (100, 512, 64, 256, {})])
def test_wide_basis_groups_bit_exact(d, W, gs, bgs, kw):
    """`basis_group_size`: the walkers sharing one Haar basis may be a multiple of the R-1
    group (incremental mode): the basis stream is indexed by the WIDE group, the moments keep
    the groups of group_size walkers."""
    eng, prob, st = make_pair(d, W, gs, incremental=True, basis_group_size=bgs, **kw)
    w0 = kw.get("walker_offset", 0)
    for n in (3, 2 * d + 1, 11):
        eng.step(n)
        eng.sync()
        st.run(n, walker0=w0, n_threads=8)
        compare_state(eng, st)
    shift = st.x.mean(0)
    eng.set_moment_shift(shift)
    eng.accumulate_moments()
    gsum, S = O.moments(st.x, gs, shift=shift)
    n, g_gs, g_S = eng.read_moments()
    assert g_gs.shape == (W // gs, d)
    assert_bit_equal(g_gs, gsum, "group sums")
    assert_bit_equal(g_S, S, "pooled second moments")
    with pytest.raises(E.EngineError, match="incremental"):
        E.Engine(4, 512, group_size=64, basis_group_size=256)
    eng.close()


@pytest.mark.parametrize("d,W,gs,blocks", [(30, 1024, 64, None), (3, 512, 128, None),
                                            (100, 512, 64, None),
                                            (12, 512, 64, [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9, 10, 11]])])
def test_device_checkpoint_matches_the_host_arithmetic(d, W, gs, blocks):
    """Row N2: the learn / convergence checkpoint on the device -- window sums over
    the ring of intervals, R-1 of the means (mcmc.py:856-889) and the refreshed transform
    (proposal.py:226-260) -- against the host routines on the same statistics
    (`gelman_rubin`, golden G7's arithmetic, and `set_proposal_cov`)."""
    eng, prob, st = make_pair(d, W, gs, incremental=True, blocks=blocks,
                              over=[1, 2] if blocks else None)
    eng.set_moment_shift(np.full(d, 0.5))
    eng.checkpoint_set_ring()
    T0 = eng.get_proposal_transform()
    intervals, acc_last, steps_last = [], 0, 0
    for k, (n_launch, window, learn) in enumerate([(3, 1, False), (2, 2, False), (4, 2, True),
                                                   (3, 4, True)]):
        for _ in range(n_launch):
            eng.step(2 * d)
            eng.accumulate_moments()
        eng.request_moments()
        lo, hi = (0.0, np.inf) if learn else (np.inf, -np.inf)
        n_win = sum(iv[0] for iv in intervals[len(intervals) + 1 - window:]) + n_launch
        steps = eng.counters()["steps"]
        ptr, n = eng.checkpoint_begin(window, n_win, steps - steps_last)
        assert n == 5 + 2 * d * d + d and ptr != 0
        eng.checkpoint_solve(lo, hi)
        eng.step(d)                                   # (work queued behind the checkpoint)
        n_snap, gsum, S, c = eng.fetch_moments()
        assert n_snap == n_launch
        intervals.append((n_snap, gsum, S))
        dev = eng.checkpoint_fetch()
        # the host's version of the same checkpoint (sampler.check_convergence_and_learn_proposal)
        ivs = intervals[-window:]
        n_tot = sum(iv[0] for iv in ivs)
        g_sum, S_sum = sum(iv[1] for iv in ivs), sum(iv[2] for iv in ivs)
        N_c = float(n_tot * gs)
        means = g_sum / N_c
        mm = means.T @ means
        R, mean_of_covs = E.gelman_rubin(float(W // gs), N_c * (W // gs), S_sum - N_c * mm,
                                         means.sum(0), mm)
        assert dev["status"] == 0 and dev["n_chains"] == W // gs and dev["sum_N"] == N_c * (W // gs)
        assert dev["accepted"] == c["accepted"] and dev["d_accepted"] == c["accepted"] - acc_last
        assert dev["d_steps"] == (steps - steps_last) * W
        np.testing.assert_allclose(dev["mean_of_covs"], mean_of_covs, rtol=1e-9, atol=1e-18)
        np.testing.assert_allclose(dev["Rminus1_groups"], R, rtol=1e-8)
        T1 = eng.get_proposal_transform()
        if learn:
            assert dev["refreshed"]
            ref = E.Engine(d, gs, group_size=gs, incremental=True)
            ref.set_prior([0] * d, [0.0] * d, [1.0] * d)
            if blocks:
                ref.set_blocking(blocks, [1, 2])
            ref.set_proposal_cov(dev["mean_of_covs"])
            # from the device's own mean of covariances the transform is the host's, bit for bit
            assert_bit_equal(T1, ref.get_proposal_transform(), "T")
            assert_bit_equal(eng.get_proposal_cov(), dev["mean_of_covs"], "cov")
            ref.close()
        else:
            assert not dev["refreshed"]
            assert_bit_equal(T1, T0, "T untouched")
        T0 = T1
        acc_last, steps_last = c["accepted"], steps + d
        steps_last = steps
    eng.close()


def test_device_checkpoint_payload_read_out():
    """`device_checkpoint: reduce`: checkpoint_begin -> checkpoint_request_payload -> [a launch] ->
    checkpoint_fetch_payload hands the host the statistics the all-reduce carries (window sums over
    the device ring, group means, sum of outer products: mcmc.py:791-793's gather) --
    equal to the host path's payload from the same moments (counts exactly, sums to rounding: the
    device adds the groups one by one, numpy in BLAS order)."""
    d, W, gs = 12, 1024, 64
    eng, prob, st = make_pair(d, W, gs, incremental=True)
    eng.set_moment_shift(np.full(d, 0.5))
    eng.checkpoint_set_ring()
    intervals, acc_last, steps_last = [], 0, 0
    with pytest.raises(E.EngineError, match="checkpoint_begin must precede"):
        eng.checkpoint_request_payload()
    for n_launch, window in [(3, 1), (2, 2), (4, 2), (1, 3)]:
        for _ in range(n_launch):
            eng.step(2 * d)
            eng.accumulate_moments()
        eng.request_moments()
        n_win = sum(iv[0] for iv in intervals[len(intervals) + 1 - window:]) + n_launch
        steps = eng.counters()["steps"]
        ptr, n = eng.checkpoint_begin(window, n_win, steps - steps_last)
        assert n == 5 + 2 * d * d + d
        eng.checkpoint_request_payload()
        with pytest.raises(E.EngineError, match="no device checkpoint is pending"):
            eng.checkpoint_fetch()
        eng.step(d)                                   # (work queued behind the read-out)
        n_snap, gsum, S, c = eng.fetch_moments()
        intervals.append((n_snap, gsum, S))
        got = eng.checkpoint_fetch_payload()
        with pytest.raises(E.EngineError, match="no payload read-out is pending"):
            eng.checkpoint_fetch_payload()
        ivs = intervals[-window:]
        g_sum, S_sum = sum(iv[1] for iv in ivs), sum(iv[2] for iv in ivs)
        N_c = float(sum(iv[0] for iv in ivs) * gs)
        means = g_sum / N_c
        mm = means.T @ means
        G = W // gs
        assert got[0] == G and got[1] == N_c * G and got[2] == c["accepted"] - acc_last
        assert got[3] == (steps - steps_last) * W and got[4] == c["accepted"]
        scale = np.abs(S_sum).max()
        np.testing.assert_allclose(got[5:5 + d * d].reshape(d, d), S_sum - N_c * mm, rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(got[5 + d * d:5 + d * d + d], means.sum(0), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(got[5 + d * d + d:].reshape(d, d), mm, rtol=1e-12, atol=1e-16)
        # the host's solve of the device's payload = the host path's, to rounding
        R_dev, cov_dev = E.gelman_rubin(got[0], got[1], got[5:5 + d * d].reshape(d, d),
                                        got[5 + d * d:5 + d * d + d], got[5 + d * d + d:].reshape(d, d))
        R_host, cov_host = E.gelman_rubin(float(G), N_c * G, S_sum - N_c * mm, means.sum(0), mm)
        np.testing.assert_allclose(R_dev, R_host, rtol=1e-8)
        np.testing.assert_allclose(cov_dev, cov_host, rtol=1e-9, atol=1e-18)
        acc_last, steps_last = c["accepted"], steps
    eng.close()


@pytest.mark.parametrize("d,W,gs,kw", [
    (2, 256, 64, {}), (7, 256, 64, {}), (30, 512, 256, {"basis_group_size": 512}),
    (48, 256, 64, {}), (52, 256, 64, {}), (56, 256, 128, {}), (100, 256, 64, {}),
    (104, 128, 64, {}), (128, 128, 64, {}),
    # normal priors (MODE 2), bounds that differ (MODE 1), a temperature
    (27, 256, 64, dict(kinds=[0] * 6 + [1] * 21, a=[0.0] * 6 + [0.5] * 21, b=[1.0] * 6 + [0.25] * 21)),
    (13, 256, 64, dict(a=[-0.25 * (i % 3) for i in range(13)], b=[1.0 + 0.5 * (i % 2) for i in range(13)],
                       T=1.4)),
    # parameter blocks with a one-parameter block; emitted rows
    (12, 256, 64, {"blocks": [[0, 1, 2], [3], list(range(4, 12))], "over": [1, 2, 3]}),
    (30, 256, 64, {"cap": 400})])
def test_calls_of_several_launches_refresh_y_inside_the_step_kernel(d, W, gs, kw):
    """Round 5 (capi.hip plan_span, step_inc_kernel `anchor & 2`): a call that spans several refresh
    intervals forms its directions as ONE set and every launch after the first refreshes
    y = L^-1 (x - mu) itself -- deviations and eight-column tiles of L^-1 through LDS, the same
    ascending chains as whiten_state_kernel -- before it re-anchors the carried log-likelihood.
    Every padded dimension dq = 1 ... 32 class, every prior mode, blocks and emitted rows: bit for
    bit the oracle's, the carried residual included."""
    kw = dict(kw)
    cap = kw.pop("cap", 0)
    eng, prob, st = make_pair(d, W, gs, incremental=True, cap=cap, max_tries=1e9, **kw)
    R = prob.refresh_every
    for n in (3, 3 * R + 5, 2 * R):      # launches cut inside the call at the multiples of R
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        if cap:
            assert_bit_equal(eng.drain_samples(), st.drain(), "rows")
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residual")
    assert "step_inc_kernel" in eng.last_step_kernel()
    assert eng.counters()["accepted"] == int(st.n_accept.sum())
    eng.close()
