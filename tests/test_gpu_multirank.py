"""BASELINE configs[2]'s machinery on real devices: N ranks, one per GPU, joined by the
library's own RCCL communicator (include/mcmc_hip.h `mcmc_hip_comm_*`), the checkpoint's
all-reduce queued in place on the engine's stream.  The multi-rank tests run for every
2 <= N <= number of visible GPUs and skip on a one-GPU box (RCCL refuses two ranks on one
device) -- there the launcher itself is still exercised with the gloo stand-in."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, world, mode):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_rccl_ranks_worker.py"),
                                       str(tmp_path), mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()[-3000:]
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


WORLDS = [n for n in (2, 4, 8) if n <= max(n_gpus(), 1)] or [2]


@pytest.mark.parametrize("world", WORLDS)
def test_ranks_over_rccl_agree_and_learn(tmp_path, world):
    """Every rank forms the same R-1 and learns the same proposal from the in-stream
    all-reduce (bit-identical: one reduction, every rank solves the same numbers)."""
    if n_gpus() < world:
        pytest.skip(f"{world} ranks need {world} GPUs (RCCL: one device per rank); {n_gpus()} visible")
    res = run_ranks(tmp_path, world, "learn")
    a = res[0]
    for r, b in enumerate(res):
        c = b["collective"]
        assert (c["backend"], c["world_size"], c["nranks_seen"]) == ("nccl", world, world)
        assert c["library"].startswith("libmcmc_hip.so (RCCL ")
        assert b["device_checkpoint"] and b["comm_attached"] and b["device"] == r
        assert b["walker_offset"] == 2048 * r and b["steps"] == a["steps"]
        assert b["progress"] == a["progress"] and b["proposal_cov"] == a["proposal_cov"]
        assert 0 < b["allreduce_in_stream_us"] < 5000
    assert len(a["progress"]) >= 3
    t = np.load(os.path.join(HERE, "golden", "targets.npz"))
    cov, got = t["cov_d30"], np.array(a["proposal_cov"])
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(got - cov) / np.outer(sig, sig)) < 0.15     # the learned proposal


@pytest.mark.parametrize("world", WORLDS)
def test_shards_on_several_gpus_equal_the_slices_of_one_ensemble(tmp_path, world):
    """rank r's walkers ARE walkers [2048 r, 2048 (r + 1)) of a single-GPU ensemble of
    2048 x world walkers, bit for bit (proposal fixed), and the R-1 of the all-reduced statistics
    is the single ensemble's to rounding."""
    if n_gpus() < world:
        pytest.skip(f"{world} ranks need {world} GPUs; {n_gpus()} visible")
    sys.path.insert(0, HERE)
    import _rccl_ranks_worker as w
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    res = run_ranks(tmp_path, world, "nolearn")
    assert res[0]["checkpoint_mode"] == "reduce" and res[0]["checkpoint_lag"] == 2   # the defaults for N > 1
    opts = dict(w.options(world, False, walkers=2048 * world), checkpoint_lag=res[0]["checkpoint_lag"],
                device_checkpoint=res[0]["checkpoint_mode"])
    one = MCMCHip(opts, ProblemSpec.from_info(w.problem()))
    one.run()
    st = one.engine.get_full_state()
    assert one.n_steps_raw == res[0]["steps"]
    for r in range(world):
        z = np.load(tmp_path / f"state_rank{r}.npz")
        sl = slice(2048 * r, 2048 * (r + 1))
        for k in ("x", "logpost", "weight", "n_accept"):
            assert np.array_equal(z[k], st[k][sl]), (r, k)
    prog = one.progress[["N", "acceptance_rate", "Rminus1"]].to_numpy()
    got = np.array(res[0]["progress"])
    assert np.array_equal(prog[:, 0], got[:, 0])
    np.testing.assert_allclose(got[:, 1:], prog[:, 1:], rtol=1e-8)
    one.close()


def bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env,
                         capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns the two ranks itself
    (VERDICT r3 item 1a).  With two GPUs visible the collective is the library's RCCL; on this
    one-GPU box the ranks share cuda:0 and fall back to the gloo stand-in -- and the line says so."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = bench("--gpus", "2", "--steps", "6", "--warmup", "6", "--walkers", "16384",
                "--steps-per-launch", "900", env=env)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    c = res["collective"]
    assert c["world_size"] == 2 and c["nranks_seen"] == 2
    assert c["backend"] == ("nccl" if n_gpus() >= 2 else "gloo")
    assert len(c["per_rank_step_kernel_ms"]) == 2 and min(c["per_rank_step_kernel_ms"]) > 0
    assert res["config"]["evals_per_step"] == 2 * 16384 * 900
    assert res["config"]["checkpoint_on"].startswith("device (window sums" if n_gpus() >= 2 else "host")
    assert res["config"]["learn_checkpoints_in_timed_region"] >= 1
    assert res["value"] > 1e8 and res["cpu_baseline"] is None
    if n_gpus() >= 2:
        assert 0 < c["checkpoint_allreduce_in_stream_us"] < 5000


def test_bench_line_through_the_launcher_matches_the_direct_one():
    """N = 1 through the same entry point: `--gpus 1` runs in this process (no ranks spawned)."""
    res = bench("--gpus", "1", "--steps", "8", "--warmup", "4", "--no-variants", "--no-cpu-baseline")
    assert res["n_gpus"] == 1 and res["collective"]["backend"] is None
    assert res["config"]["checkpoint_on"] == "host" and res["value"] > 3e10


def test_a_refused_rccl_communicator_falls_back_to_gloo_on_every_rank():
    """Robustness of the N > 1 start-up: two ranks forced onto RCCL on ONE device -- RCCL refuses
    (duplicate GPU) --; every rank learns of it through the bootstrap group, the job continues on
    the gloo stand-in and the bench line says so (`collective.rccl_error`).  With two GPUs this
    scenario cannot be staged: skipped."""
    if n_gpus() >= 2:
        pytest.skip("needs ranks that share a device")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MCMC_HIP_BACKEND"] = "rccl"
    res = bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--walkers", "16384",
                "--steps-per-launch", "600", "--cross-check-seconds", "0", env=env)
    c = res["collective"]
    assert res["n_gpus"] == 2 and c["backend"] == "gloo" and c["nranks_seen"] == 2
    assert "rccl_error" in c and "ncclCommInitRank" in c["rccl_error"]
    assert res["config"]["checkpoint_on"] == "host" and res["value"] > 1e8
