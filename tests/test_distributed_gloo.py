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


def test_run_loop_world_size_2(tmp_path):
    """The whole `run()` loop on two ranks (oracle-backed engine double, each rank its shard
    of the walkers): the checkpoints are processed two launches after their request (the
    multi-process default, sampler.advance), every rank sees the same R-1, acceptance rate
    and learned covariance at every checkpoint, and the ranks really hold different walkers."""
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_run_worker.py"),
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]
    a, b = (json.load(open(tmp_path / f"run_rank{r}.json")) for r in range(2))
    assert a["size"] == b["size"] == 2 and a["lag"] == b["lag"] == 2
    assert (a["walker_offset"], b["walker_offset"]) == (0, 128)
    assert len(a["Rminus1"]) >= 3 and a["steps"] == b["steps"]
    for k in ("Rminus1", "N", "acc", "cov"):
        assert a[k] == b[k], k                 # one all-reduce: bit-identical on every rank
    assert a["xsum"] != b["xsum"]
    for r in (a, b):
        req = [n for what, n in r["log"] if what == "request"]
        ref = [n for what, n in r["log"] if what == "refresh"][1:]
        assert len(ref) >= 3 and all(y - x == 2 for x, y in zip(req, ref)), r["log"]
    for ext in (".checkpoint", ".covmat", ".progress", ".1.txt", ".2.txt", ".1.state.npz",
                ".2.state.npz"):
        assert os.path.exists(str(tmp_path / "run") + ext), ext
