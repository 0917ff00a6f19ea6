"""The N > 1 path on CPU: world_size 2, `gloo` backend, 127.0.0.1 rendezvous.  Each rank owns
half of the walker groups; the checkpoint's single all-reduce of pooled sufficient statistics
must give every rank the R-1 and the learned covariance that the reference arithmetic
(oracle/ref_numpy.rminus1_of_means, pinned to golden G7) gives for ALL groups together."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_numpy as R

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_checkpoint_allreduce_world_size_2(tmp_path):
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"),
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-3000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    # expected, from the raw synthetic samples of both ranks, with the reference arithmetic
    sys.path.insert(0, HERE)
    import _gloo_worker as w
    chains = np.concatenate([w.synthetic(r) for r in range(2)], axis=1)  # [snap, 2G, gs, d]
    m = chains.shape[1]
    flat = chains.transpose(1, 0, 2, 3).reshape(m, -1, w.D)
    means = flat.mean(1)
    covs = np.array([np.cov(c.T, ddof=0) for c in flat])
    Ns = np.full(m, flat.shape[1], dtype=float)
    Rref, Wref = R.rminus1_of_means(Ns, means, covs)
    Rref *= w.GS   # the sampler quotes R-1 per walker: a chain of the statistic is a group
    for r in res:
        assert abs(r["Rminus1"] - Rref) <= 1e-9 * Rref
        np.testing.assert_allclose(r["new_cov"], Wref, rtol=1e-10)
        assert abs(r["acc"] - 3000 / (50 * 128 * 2)) < 1e-12 and r["N"] == 3000
        np.testing.assert_allclose(r["buf"], np.arange(6).reshape(2, 3) * 3.0)
    assert res[0]["Rminus1"] == res[1]["Rminus1"] and res[0]["new_cov"] == res[1]["new_cov"]
    assert res[1]["gathered"] is None and len(res[0]["gathered"]) == 2


def _run_ranks(tmp_path, world, *args):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_run_worker.py"),
                                       str(tmp_path), *args], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()[-3000:]
    return [json.load(open(tmp_path / f"run_rank{r}.json")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_run_loop_on_several_ranks(tmp_path, world):
    """The whole `run()` loop on 2, 4 and 8 ranks (oracle-backed engine double, each rank its
    shard of the walkers -- BASELINE configs[2]'s layout): the checkpoints are processed two
    launches after their request (the multi-process default, sampler.advance), every rank sees
    the same R-1, acceptance rate and learned covariance at every checkpoint, and the ranks
    really hold different walkers."""
    res = _run_ranks(tmp_path, world)
    a = res[0]
    assert all(r["size"] == world and r["lag"] == 2 for r in res)
    assert [r["walker_offset"] for r in res] == [128 * k for k in range(world)]
    assert len(a["Rminus1"]) >= 3
    for b in res[1:]:
        assert a["steps"] == b["steps"]
        for k in ("Rminus1", "N", "acc", "cov"):
            assert a[k] == b[k], k                 # one all-reduce: bit-identical on every rank
    assert len({r["xsum"] for r in res}) == world
    for r in res:
        req = [n for what, n in r["log"] if what == "request"]
        ref = [n for what, n in r["log"] if what == "refresh"][1:]
        assert len(ref) >= 3 and all(y - x == 2 for x, y in zip(req, ref)), r["log"]
    exts = [".checkpoint", ".covmat", ".progress"] + [f".{k}.{e}" for k in range(1, world + 1)
                                                      for e in ("txt", "state.npz")]
    for ext in exts:
        assert os.path.exists(str(tmp_path / "run") + ext), ext


@pytest.mark.parametrize("world", [4, 8])
def test_shards_equal_the_slices_of_one_ensemble(tmp_path, world):
    """"walker w -> GPU w // (W/G) ... results independent of G".  With the proposal
    fixed, rank r's 128 walkers after the whole run ARE walkers [128 r, 128 (r + 1)) of a
    single-process ensemble of 128 x world walkers, bit for bit (initial points and Philox
    streams are keyed by GLOBAL group / walker ids); and the R-1 every rank forms from the
    all-reduced statistics is the single process's to rounding (the summation order differs)."""
    from cobaya_amd.model import ProblemSpec
    from tests.test_host_logic import QUICK
    from tests.test_sampler_on_oracle import OnOracle
    res = _run_ranks(tmp_path, world, "nolearn")
    one = OnOracle({"seed": 5, "n_walkers": 128 * world, "group_size": 64, "steps_per_launch": 40,
                    "max_samples": 30000 * world, "Rminus1_stop": 0.0, "learn_every": "40d",
                    "learn_proposal": False, "checkpoint_lag": 2}, ProblemSpec.from_info(QUICK))
    one.run()
    st = one.engine.get_full_state()
    assert one.n_steps_raw == res[0]["steps"]
    for r in range(world):
        z = np.load(tmp_path / f"state_rank{r}.npz")
        sl = slice(128 * r, 128 * (r + 1))
        for k in ("x", "logpost", "weight", "n_accept"):
            assert np.array_equal(z[k], st[k][sl]), (r, k)
    prog = one.progress
    assert [int(v) for v in prog["N"]] == res[0]["N"]
    np.testing.assert_allclose(prog["Rminus1"].to_numpy(float), res[0]["Rminus1"], rtol=1e-9)
    np.testing.assert_allclose(prog["acceptance_rate"].to_numpy(float), res[0]["acc"], rtol=1e-13)


@pytest.mark.parametrize("mode", ["id", "init"])
def test_rccl_fallback_does_not_hang(mode):
    """ADVICE r4 (medium + low): asked for RCCL where it cannot be had, every rank must come out
    of `dist.init_from_env` on the gloo stand-in -- (id) rank 0 fails BEFORE the id broadcast
    (it now always broadcasts: the id or the reason); (init) one rank sits in ncclCommInitRank
    past the deadline and joins late: its communicator is destroyed by the helper thread, never
    published behind the main thread's back.  Both used to leave the ranks in mismatched
    collectives."""
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCMC_HIP_RCCL_TIMEOUT="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_rccl_fallback_worker.py"),
                                       mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    try:
        for p in procs:
            out, _ = p.communicate(timeout=120)
            assert p.returncode == 0, out.decode()[-3000:]
            assert b" ok after " in out
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_config2_full_size_on_eight_ranks(tmp_path):
    """BASELINE configs[2] at its FULL size -- 8 ranks x 65 536 walkers of the d = 30 target,
    R-1 groups of 256, one Haar basis per 4 096 walkers -- through the whole run loop on the
    oracle-backed engine, with learn checkpoints and their all-reduce (mcmc.py:791-793, 1005-1007,
    1021).  Shortened in TIME only (40-step launches, a checkpoint per d accepted steps): >= 2
    checkpoints are processed, every rank forms the same R-1 / acceptance / learned covariance
    from the reduced statistics, the shards start at 65 536 k, and rank 7's shard after the run
    IS walkers [458 752, 524 288) of the ensemble: an independent replay of those walkers on the
    oracle, fed the logged proposal refreshes, gives the same state bit for bit."""
    from oracle import cbind as O
    import bench
    W, world = 65536, 8
    res = _run_ranks(tmp_path, world, "config2", str(int(0.3 * 8 * W * 330)))
    a = res[0]
    assert [r["walker_offset"] for r in res] == [W * k for k in range(world)]
    assert all(r["size"] == world and r["lag"] == 2 for r in res)
    assert len(a["Rminus1"]) >= 2 and all(np.isfinite(a["Rminus1"]))
    for b in res[1:]:
        assert a["steps"] == b["steps"]
        for k in ("Rminus1", "N", "acc", "cov"):
            assert a[k] == b[k], k
    assert all(0.15 < x < 0.5 for x in a["acc"])
    # the accepted total of the last checkpoint is the sum over all 524 288 walkers
    assert a["N"][-1] > 0.15 * world * W * 200
    # independent replay of the last rank's shard
    ref = np.load(tmp_path / "refreshes_last_rank.npz")
    assert len(ref["launches"]) >= 3          # the initial covmat + >= 2 learned proposals
    mean, cov = bench.target(30)
    d = 30
    x0 = np.load(tmp_path / "x0_last_rank.npy")
    prob = O.Problem(d, [0] * d, [0.0] * d, [1.0] * d, means=mean, covs=cov,
                     T=O.proposal_transform(ref["covs"][0], 2.4), group_size=4096, seed=1,
                     incremental=True)
    st = O.State(prob, x0)
    n_launch = a["steps"] // 40
    for k in range(n_launch):
        for j in np.nonzero(ref["launches"] == k)[0]:
            if k or j:
                prob.set_T(O.proposal_transform(ref["covs"][j], 2.4))
        st.run(40, walker0=7 * W, n_threads=O.max_threads())
    z = np.load(tmp_path / "state_rank7.npz")
    assert np.array_equal(z["x"].view(np.uint64), st.x.view(np.uint64))
    assert np.array_equal(z["logpost"].view(np.uint64), st.logpost.view(np.uint64))
    assert np.array_equal(z["weight"], st.weight) and np.array_equal(z["n_accept"], st.n_accept)
