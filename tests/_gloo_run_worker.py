"""Worker of tests/test_distributed_gloo.py::test_run_loop_world_size_2: one rank of a
world_size-2 `gloo` group running the WHOLE sampler loop (`MCMCHip.run`: fused launches,
moment snapshots, the checkpoint with its all-reduce processed `checkpoint_lag` launches
after its request, proposal refresh) with the ctypes seam served by the oracle-backed double
(tests/oracle_engine.py), every rank owning its shard of the walkers."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cobaya_amd import dist  # noqa: E402
from cobaya_amd.model import ProblemSpec  # noqa: E402
from cobaya_amd.sampler import MCMCHip  # noqa: E402
from tests.oracle_engine import OracleEngine  # noqa: E402
from tests.test_host_logic import QUICK  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_from_env(backend="gloo")
    log = []

    class Spy(OracleEngine):
        launches = 0

        def step(self, n):
            self.launches += 1
            return super().step(n)

        def request_moments(self):
            log.append(["request", self.launches])
            return super().request_moments()

        def set_proposal_cov(self, cov):
            log.append(["refresh", self.launches])
            return super().set_proposal_cov(cov)

    class S(MCMCHip):
        _engine_factory = staticmethod(Spy)

    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    learn = mode != "nolearn"
    covs = []
    if mode == "config2":
        # BASELINE configs[2] at FULL size: 8 x 65 536 walkers of the d = 30 target, R-1 groups of
        # 256, one Haar basis per 4 096 walkers (the sampler's own choice at this size), learn
        # checkpoints with the all-reduce -- shortened in TIME only (launches of 40 steps, a
        # checkpoint every d accepted steps per chain) so that the CPU oracle finishes in seconds
        import bench
        mean, cov = bench.target(30)
        info = bench.make_info(30, mean, cov, 65536, 256, 40)
        opts = dict(info["sampler"]["mcmc_hip"], seed=1, learn_every="1d", max_samples=int(sys.argv[3]),
                    learn_proposal_Rminus1_max=1e9, max_rows=0)
        spec = ProblemSpec.from_info(info)
        set_cov = Spy.set_proposal_cov

        def logged_set_cov(self, c):
            covs.append((self.launches, np.array(c)))
            return set_cov(self, c)
        Spy.set_proposal_cov = logged_set_cov
        s = S(opts, spec)
        assert (s.group_size, s.basis_group_size, s.incremental) == (256, 4096, True)
        if dist.rank() == dist.size() - 1:
            np.save(os.path.join(out_dir, "x0_last_rank.npy"), s.engine.get_full_state()["x"])
    else:
        s = S({"seed": 5, "n_walkers": 128, "group_size": 64, "steps_per_launch": 40,
               "max_samples": 30000 * dist.size(), "Rminus1_stop": 0.0, "learn_every": "40d",
               "learn_proposal": learn},
              ProblemSpec.from_info(QUICK), output=os.path.join(out_dir, "run"))
    s.run()
    if covs and dist.rank() == dist.size() - 1:
        np.savez(os.path.join(out_dir, "refreshes_last_rank.npz"),
                 launches=np.array([k for k, _ in covs]), covs=np.array([c for _, c in covs]))
    st = s.engine.get_full_state()
    np.savez(os.path.join(out_dir, f"state_rank{dist.rank()}.npz"), x=st["x"], logpost=st["logpost"],
             weight=st["weight"], n_accept=st["n_accept"])
    prog = s.progress
    res = {"rank": dist.rank(), "size": dist.size(), "lag": s._ckpt_lag, "log": log,
           "walker_offset": int(s.engine.walker_offset),
           "Rminus1": [float(v) for v in prog["Rminus1"]],
           "N": [int(v) for v in prog["N"]],
           "acc": [float(v) for v in prog["acceptance_rate"]],
           "cov": np.asarray(s.engine.get_proposal_cov()).tolist(),
           "xsum": float(np.sum(st["x"])), "steps": int(s.n_steps_raw)}
    with open(os.path.join(out_dir, f"run_rank{dist.rank()}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()


if __name__ == "__main__":
    main()
