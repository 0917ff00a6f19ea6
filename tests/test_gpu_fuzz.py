"""GPU parity fuzz (Tier B): randomly drawn problem shapes -- dimension, ensemble and group
size, number of modes, prior kinds, periodic parameters, temperature, burn-in, parameter
blocks with oversampling, dragging -- stepped in uneven launches and compared bit for bit
with the oracle.  The draws are seeded: the cases are the same on every run."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_parity import compare_state, make_pair  # noqa: E402


def draw_case(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(2, 33))
    gs = int(rng.choice([64, 128, 256]))
    W = gs * int(rng.integers(1, 4)) if rng.random() < 0.5 else 256 * int(rng.integers(1, 3))
    if W % gs:
        W = gs * max(1, W // gs)
    K = int(rng.choice([1, 1, 1, 2, 3]))
    kw = {}
    if rng.random() < 0.4:   # some normal priors
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.05, 0.3)) if k else 1.0 for k in kinds])
    if rng.random() < 0.3:   # periodic among the uniform ones
        kinds = kw.get("kinds", [0] * d)
        kw["periodic"] = [int(k == 0 and rng.random() < 0.3) for k in kinds]
    if rng.random() < 0.3:
        kw["T"] = float(rng.choice([1.5, 2.0, 3.0]))
    if rng.random() < 0.3:
        kw["burn_in"] = int(rng.integers(1, 5))
    if K > 1:
        w = rng.uniform(0.2, 1.0, K)
        kw["weights"] = (w / w.sum()).tolist()
    if rng.random() < 0.5 and d >= 3:   # parameter blocks
        perm = rng.permutation(d).tolist()
        nb = int(rng.integers(2, min(4, d) + 1))
        cuts = sorted(rng.choice(np.arange(1, d), size=nb - 1, replace=False).tolist())
        blocks = [perm[a:b] for a, b in zip([0] + cuts, cuts + [d])]
        over = sorted(int(v) for v in rng.integers(1, 4, size=nb))
        kw.update(blocks=blocks, over=over)
        if rng.random() < 0.4:
            kw.update(drag_last_slow=int(rng.integers(0, nb - 1)), drag_steps=int(rng.integers(2, 6)))
            kw["over"] = [1] * nb
    steps = [int(v) for v in rng.integers(1, 25, size=3)]
    return d, W, gs, K, kw, steps


# MCMC_FUZZ_CASES=400 widens the hunt (the default keeps the suite short)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_CASES", "40")))))
def test_random_shapes_bit_exact(seed):
    from cobaya_amd.engine import EngineError
    d, W, gs, K, kw, steps = draw_case(1000 + seed)
    eng, prob, st = make_pair(d, W, gs, K=K, **kw)
    for n in steps:
        try:
            eng.step(n)
        except EngineError as e:   # a long cycle of a wide problem with small groups
            if "KiB of LDS per group" in str(e):
                pytest.skip(str(e))
            raise
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
    c = eng.counters()
    assert c["steps"] == sum(steps) and c["accepted"] == int(st.n_accept.sum())


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_BIG_CASES", "20")))))
def test_random_big_dimensions_bit_exact(seed):
    """d = 33 .. 128: the two-wave kernel (d <= 56), the matrix-core kernel (ensembles that are
    a multiple of 256), the column-sweep fallback (other sizes) and the general kernel
    (mixtures, `one`, periodic parameters, odd ensembles with normal priors or d > 112); any
    group size, launches that stop mid-cycle."""
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.integers(33, 129))
    gs = int(rng.choice([64, 128, 256]))
    W = (int(rng.choice([256, 512])) if rng.random() < 0.6 else gs * int(rng.integers(1, 4)))
    if W % gs:
        W = gs * max(1, W // gs)
    kw = {}
    if rng.random() < 0.3:
        kw["T"] = 2.0
    if rng.random() < 0.3:
        kw["burn_in"] = 2
    kinds = [0] * d
    if rng.random() < 0.5:
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    if rng.random() < 0.2:
        kw["periodic"] = [int(not k and rng.random() < 0.1) for k in kinds]
    if rng.random() < 0.25:
        kw["K"] = int(rng.choice([0, 2, 3]))
    eng, prob, st = make_pair(d, W, gs, **kw)
    for n in (int(rng.integers(1, 10)), int(rng.integers(5, d)), int(rng.integers(1, 30))):
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)


def draw_incremental_case(seed):
    """A random problem incremental evaluation serves: d = 2..128, 1..4 modes (d <= 64 above one
    mode), uniform / normal priors, temperature, burn-in, parameter blocks (of any size for one
    mode, of >= 2 parameters for a mixture) with oversampling, or dragging (one mode)."""
    rng = np.random.default_rng(seed)
    d = int(rng.choice([int(rng.integers(2, 33)), int(rng.integers(33, 65)), int(rng.integers(65, 129))],
                       p=[0.6, 0.25, 0.15]))
    gs = int(rng.choice([64, 128, 256]))
    W = gs * int(rng.integers(1, 4))
    K = int(rng.choice([1, 1, 2, 3, 4])) if d <= 64 else 1
    kw = {}
    if rng.random() < 0.4:
        kinds = (rng.random(d) < 0.5).astype(int).tolist()
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.05, 0.3)) if k else 1.0 for k in kinds])
    elif rng.random() < 0.3:   # uniform priors on different intervals (MODE 1)
        kw.update(a=[float(v) for v in rng.uniform(-0.5, 0.0, d)],
                  b=[float(v) for v in rng.uniform(1.0, 1.5, d)])
    if rng.random() < 0.3:
        kw["T"] = float(rng.choice([1.5, 2.0]))
    if rng.random() < 0.3:
        kw["burn_in"] = int(rng.integers(1, 4))
    if K > 1:
        w = rng.uniform(0.2, 1.0, K)
        kw["weights"] = (w / w.sum()).tolist()
    L = d
    if rng.random() < 0.5 and d >= 4:
        perm = rng.permutation(d).tolist()
        nb = int(rng.integers(2, min(4, d // 2) + 1))
        cuts = sorted(rng.choice(np.arange(1, d), size=nb - 1, replace=False).tolist())
        blocks = [perm[a:b] for a, b in zip([0] + cuts, cuts + [d])]
        # (one-parameter blocks too, also under a mixture)
        over = sorted(int(v) for v in rng.integers(1, 4, size=nb))
        kw.update(blocks=blocks, over=over)
        L = sum(o * len(b) for o, b in zip(over, blocks))
        if K == 1 and rng.random() < 0.5:
            last = int(rng.integers(0, nb - 1))
            kw.update(drag_last_slow=last, drag_steps=int(rng.integers(2, 6)), over=[1] * nb)
            L = sum(len(b) for b in blocks[:last + 1])
    # MCMC_FUZZ_INC_PERIODIC=1 (developer switch, off in the committed runs: the periodic kernel
    # has its own cases in test_gpu_parity.py): one to three periodic parameters, a few sigma
    # wide around the mode, on shapes the periodic kernel serves (one mode, no dragging).  Drawn
    # from a generator of its own, so that the shapes above stay what they are.
    if os.environ.get("MCMC_FUZZ_INC_PERIODIC") and K == 1 and "drag_last_slow" not in kw:
        r2 = np.random.default_rng(77000 + seed)
        kinds = kw.get("kinds", [0] * d)
        a = list(kw.get("a", [0.0] * d))
        b = list(kw.get("b", [1.0] * d))
        per = [0] * d
        for i in r2.choice(d, size=int(r2.integers(1, min(3, d) + 1)), replace=False):
            if not kinds[i]:
                per[int(i)] = 1
                a[int(i)], b[int(i)] = 0.42, 0.58
        if any(per):
            kw.update(kinds=kinds, a=a, b=b, periodic=per)
    steps = [int(rng.integers(1, 12)), int(rng.integers(1, L + 5)), 40 * L - int(rng.integers(0, 20)),
             int(rng.integers(1, 30))]
    return d, W, gs, K, kw, steps


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_INC_CASES", "40")))))
def test_random_incremental_shapes_bit_exact(seed):
    """Incremental evaluation on randomly drawn shapes (plain, mixture, blocked, dragging kernels;
    every MODE; launches that end anywhere and cross the refresh at 40 cycle lengths): state,
    carried residuals, log-posterior, weights and counts bit for bit against the oracle."""
    from tests.test_gpu_parity import assert_bit_equal
    d, W, gs, K, kw, steps = draw_incremental_case(9000 + seed)
    eng, prob, st = make_pair(d, W, gs, K=K, incremental=True, rng=np.random.default_rng(seed), **kw)
    compare_state(eng, st)
    for n in steps:
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
    c = eng.counters()
    assert c["steps"] == sum(steps) and c["accepted"] == int(st.n_accept.sum())
    assert "inc" in eng.last_step_kernel()
    eng.close()


def draw_general_case(seed):
    """A random shape only the general incremental kernels serve (incremental_any.hip): more than
    four modes, a mixture above d = 64, periodic parameters beside a mixture, more than 16
    periodic parameters -- with normal priors, temperature, burn-in, blocks (one-parameter blocks
    too), and emitted rows on half of the cases."""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    if kind == 0:      # many modes
        d, K = int(rng.integers(2, 41)), int(rng.integers(5, 17))
        while ((d + 3) // 4) * (K + 1) > 208:
            K -= 1
    elif kind == 1:    # a mixture above d = 64
        d, K = int(rng.integers(65, 129)), int(rng.integers(2, 5))
    elif kind == 2:    # periodic parameters beside a mixture
        d, K = int(rng.integers(3, 65)), int(rng.integers(2, 7))
    else:              # many periodic parameters, one mode
        d, K = int(rng.integers(12, 100)), 1
    gs = int(rng.choice([64, 128]))
    W = gs * int(rng.integers(1, 3))
    kw = {}
    kinds = [0] * d
    a, b = [0.0] * d, [1.0] * d
    if rng.random() < 0.4:
        kinds = (rng.random(d) < 0.4).astype(int).tolist()
        a = [0.5 if k else 0.0 for k in kinds]
        b = [float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds]
    per = [0] * d
    if kind >= 2:
        n_per = int(rng.integers(1, 4)) if kind == 2 else int(rng.integers(9, min(d, 40)))
        for i in rng.choice(d, size=min(n_per, d), replace=False):
            if not kinds[int(i)]:
                per[int(i)] = 1
                a[int(i)], b[int(i)] = 0.42, 0.58
    kw.update(kinds=kinds, a=a, b=b)
    if any(per):
        kw["periodic"] = per
    if rng.random() < 0.3:
        kw["T"] = float(rng.choice([1.5, 2.0]))
    if rng.random() < 0.3:
        kw["burn_in"] = int(rng.integers(1, 4))
    if K > 1:
        w = rng.uniform(0.2, 1.0, K)
        kw["weights"] = (w / w.sum()).tolist()
    L = d
    if rng.random() < 0.4 and d >= 4:
        perm = rng.permutation(d).tolist()
        nb = int(rng.integers(2, min(4, d // 2) + 1))
        cuts = sorted(rng.choice(np.arange(1, d), size=nb - 1, replace=False).tolist())
        blocks = [perm[p:q] for p, q in zip([0] + cuts, cuts + [d])]
        over = sorted(int(v) for v in rng.integers(1, 4, size=nb))
        kw.update(blocks=blocks, over=over)
        L = sum(o * len(bl) for o, bl in zip(over, blocks))
    if rng.random() < 0.5:
        kw["cap"] = 64
    steps = [int(rng.integers(1, 12)), int(rng.integers(1, 40)), int(rng.integers(1, 20))]
    if L <= 60:   # (short cycles: across the refresh at 40 cycle lengths)
        steps[1] = 40 * L - int(rng.integers(0, 20))
    return d, W, gs, K, kw, steps


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_GENERAL_CASES", "32")))))
def test_random_general_incremental_shapes_bit_exact(seed):
    """The general incremental kernels (register planes with and without periodic parameters, the
    LDS-state kernel; snapshots or emitted rows) on randomly drawn shapes, bit for bit against the
    oracle: state, carried residuals, counters and, where rows are emitted, the rows."""
    from tests.test_gpu_parity import assert_bit_equal
    d, W, gs, K, kw, steps = draw_general_case(12000 + seed)
    cap = kw.get("cap", 0)
    eng, prob, st = make_pair(d, W, gs, K=K, incremental=True, rng=np.random.default_rng(seed), **kw)
    compare_state(eng, st)
    for n in steps:
        for n1 in ([n] if not cap else [min(cap, n)] * (n // min(cap, n)) + ([n % cap] if n % min(cap, n) else [])):
            eng.step(n1)
            eng.sync()
            st.run(n1, n_threads=8)
            if cap:
                assert_bit_equal(eng.drain_samples(), st.drain(), "rows")
        compare_state(eng, st)
        assert_bit_equal(eng.get_full_state()["y"], st.y, "carried whitened residuals")
    c = eng.counters()
    assert c["steps"] == sum(steps) and c["accepted"] == int(st.n_accept.sum())
    name = eng.last_step_kernel()
    # (a draw whose periodic parameters mostly fell on normal priors is left with at most 16:
    # the periodic kernel's, unless rows are emitted;
    # or none: the tuned mixture kernel's)
    general = "step_inc_regs_kernel" in name or "step_inc_any_kernel" in name
    tuned = not cap and (("step_inc_kernel" in name and "periodic" in name) or "step_inc_mix_kernel" in name)
    assert general or tuned, name
    assert ("emit" in name) == bool(cap), name
    eng.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_GENERAL_CASES", "24")))))
def test_random_blocked_and_dragging_shapes_above_d32_bit_exact(seed):
    """Round 4: from-scratch parameter blocks / oversampling / dragging at 32 < d <= 128
    (step_general_kernel with blocks, drag_general_kernel) -- random block structures incl.
    one-parameter blocks, 0..3 modes, normal priors, periodic parameters, temperature, burn-in,
    emitted rows -- and rows out of the d <= 32 dragging kernel; uneven launches; state AND
    drained rows bit for bit the oracle's."""
    from tests.test_gpu_parity import assert_bit_equal
    rng = np.random.default_rng(9000 + seed)
    small = rng.random() < 0.25                     # d <= 32: drag_kernel's rows
    d = int(rng.integers(4, 33)) if small else int(rng.integers(33, 129))
    gs = int(rng.choice([64, 128]))
    W = gs * int(rng.integers(1, 3))
    K = int(rng.choice([1, 1, 2, 3, 0])) if d <= 64 else int(rng.choice([1, 1, 2]))
    kw = {}
    kinds = [0] * d
    if rng.random() < 0.4:
        kinds = (rng.random(d) < 0.4).astype(int).tolist()
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.3)) if k else 1.0 for k in kinds])
    if rng.random() < 0.35:
        kw["periodic"] = [int(k == 0 and rng.random() < 0.08) for k in kinds]
    if rng.random() < 0.3:
        kw["T"] = float(rng.choice([1.5, 2.0]))
    if rng.random() < 0.3:
        kw["burn_in"] = int(rng.integers(1, 4))
    if K > 1:
        w = rng.uniform(0.2, 1.0, K)
        kw["weights"] = (w / w.sum()).tolist()
    perm = rng.permutation(d).tolist()
    nb = int(rng.integers(2, 5))
    cuts = sorted(rng.choice(np.arange(1, d), size=nb - 1, replace=False).tolist())
    if rng.random() < 0.4:                          # force a one-parameter block
        cuts[0] = 1
        cuts = sorted(set(cuts))
        nb = len(cuts) + 1
    blocks = [perm[a:b] for a, b in zip([0] + cuts, cuts + [d])]
    drag = small or rng.random() < 0.5
    if drag:
        kw.update(blocks=blocks, over=[1] * nb, drag_last_slow=int(rng.integers(0, nb - 1)),
                  drag_steps=int(rng.integers(2, 5)))
    else:
        kw.update(blocks=blocks, over=sorted(int(v) for v in rng.integers(1, 4, size=nb)))
    cap = int(rng.choice([0, 40]))
    if small:
        cap = 40
    steps = [int(v) for v in rng.integers(1, 14, size=3)]
    eng, prob, st = make_pair(d, W, gs, K=K, cap=cap, **kw)
    for n in steps:
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=4)
        compare_state(eng, st)
        if cap:
            assert_bit_equal(eng.drain_samples(), st.drain(), "rows")
    kern = eng.last_step_kernel()
    if not small:
        assert ("drag_general_kernel" if drag else "step_general_kernel") in kern, kern
    c = eng.counters()
    assert c["steps"] == sum(steps) and c["accepted"] == int(st.n_accept.sum())


def draw_two_lane_case(seed):
    """A random shape step_duo_mix_kernel serves (kernels.h: duo_serves): 2..4 modes with
    2 dq K <= 48, whole workgroups of 128 walkers inside a basis group; uniform priors on one box
    or on boxes that differ, normal priors, T != 1, burn-in, parameter blocks of >= 2 parameters
    with oversampling."""
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 5))
    dq_max = min(12, 48 // (2 * K))
    d = int(rng.integers(2, 4 * dq_max + 1))
    gs = int(rng.choice([128, 256]))
    W = gs * int(rng.integers(1, 3))
    kw = {}
    u = rng.random()
    if u < 0.3:       # normal priors on some parameters
        kinds = (rng.random(d) < 0.4).astype(int).tolist()
        kw.update(kinds=kinds, a=[0.5 if k else 0.0 for k in kinds],
                  b=[float(rng.uniform(0.1, 0.4)) if k else 1.0 for k in kinds])
    elif u < 0.55:    # boxes that differ
        kw.update(a=[-0.25 * (i % 3) for i in range(d)], b=[1.0 + 0.5 * (i % 2) for i in range(d)])
    elif u < 0.7:     # tight boxes the walkers do reach (the exact comparisons behind the high-word test)
        kw.update(a=[0.0] * d, b=[0.62] * d)
    if rng.random() < 0.3:
        kw["T"] = float(rng.choice([1.5, 2.0]))
    if rng.random() < 0.3:
        kw["burn_in"] = int(rng.integers(1, 4))
    w = rng.uniform(0.2, 1.0, K)
    kw["weights"] = (w / w.sum()).tolist()
    L = d
    if rng.random() < 0.35 and d >= 6:
        perm = rng.permutation(d).tolist()
        nb = int(rng.integers(2, 4))
        sizes = [2] * nb
        for _ in range(d - 2 * nb):
            sizes[int(rng.integers(0, nb))] += 1
        cuts = np.cumsum(sizes).tolist()
        blocks = [perm[p:q] for p, q in zip([0] + cuts[:-1], cuts)]
        over = sorted(int(v) for v in rng.integers(1, 4, size=nb))
        kw.update(blocks=blocks, over=over)
        L = sum(o * len(bl) for o, bl in zip(over, blocks))
    steps = [int(rng.integers(1, 12)), 40 * L - int(rng.integers(0, 20)), int(rng.integers(1, 20)),
             int(rng.integers(1, 3 * L))]
    return d, W, gs, K, kw, steps


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MCMC_FUZZ_DUO_CASES", "24")))))
def test_random_two_lane_mixture_shapes_bit_exact(seed, monkeypatch):
    """step_duo_mix_kernel (incremental_duo.hip; two lanes per walker, forced with MCMC_HIP_DUO=1 at
    these small ensembles) on randomly drawn shapes, bit for bit against the oracle: state, carried
    residuals and mode log-densities, counters -- across the refresh at 40 cycle lengths."""
    from tests.test_gpu_parity import assert_bit_equal
    monkeypatch.setenv("MCMC_HIP_DUO", "1")
    d, W, gs, K, kw, steps = draw_two_lane_case(15000 + seed)
    eng, prob, st = make_pair(d, W, gs, K=K, incremental=True, rng=np.random.default_rng(seed), **kw)
    compare_state(eng, st)
    for n in steps:
        eng.step(n)
        eng.sync()
        st.run(n, n_threads=8)
        compare_state(eng, st)
        full = eng.get_full_state()
        assert_bit_equal(full["y"], st.y, "carried whitened residuals")
        assert_bit_equal(full["amode"], st.amode, "carried mode log-densities")
    c = eng.counters()
    assert c["steps"] == sum(steps) and c["accepted"] == int(st.n_accept.sum())
    assert "step_duo_mix_kernel" in eng.last_step_kernel(), eng.last_step_kernel()
    eng.close()
