"""Worker of tests/test_distributed_gloo.py: one rank of a world_size-2 `gloo` group running
the sampler's checkpoint (all-reduce of pooled sufficient statistics -> R-1 -> proposal
refresh) on a fake engine that serves synthetic ensemble statistics."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cobaya_amd import dist  # noqa: E402
from cobaya_amd.sampler import HIP_DEFAULTS, MCMC_DEFAULTS, MCMCHip  # noqa: E402
from cobaya_amd.model import ProblemSpec  # noqa: E402

D, G, GS, NSNAP = 4, 8, 16, 5


def synthetic(rank):
    rng = np.random.default_rng(100 + rank)
    A = np.random.default_rng(7).normal(size=(D, D))
    L = np.linalg.cholesky(A @ A.T / D + np.eye(D))
    offs = rng.normal(size=(G, D)) * 0.05
    return 2.0 + offs[None, :, None, :] + rng.normal(size=(NSNAP, G, GS, D)) @ L.T


class FakeEngine:
    def __init__(self, x, shift, rank):
        self.x, self.shift = x, shift
        self.group_size, self.G, self.W = GS, G, G * GS
        self.new_cov = None
        self.rank = rank

    def read_moments(self, reset=False):
        xc = self.x - self.shift
        gs = xc.sum(axis=(0, 2))
        S = np.einsum("sgwi,sgwj->ij", xc, xc)
        return NSNAP, gs, S

    def sync(self):
        pass

    def counters(self):
        return {"accepted": 1000 * (1 + self.rank), "steps": 50, "stuck": 0, "dropped_rows": 0}

    def set_proposal_cov(self, cov):
        self.new_cov = np.array(cov)


def main():
    out_dir = sys.argv[1]
    dist.init_from_env(backend="gloo")
    rank = dist.rank()
    assert dist.size() == 2
    spec = ProblemSpec.from_info({"likelihood": {"one": None},
                                  "params": {f"p{i}": {"prior": [-10, 10]} for i in range(D)}})
    s = MCMCHip.__new__(MCMCHip)
    for k, v in {**MCMC_DEFAULTS, **HIP_DEFAULTS}.items():
        setattr(s, k, v)
    s.spec = spec
    s.rank, s.size = rank, 2
    s._progress_rows = {}
    s.i_learn, s._intervals, s._dropped_snapshots = 1, [], 0
    s._acc_last = s._steps_last = 0
    s._accepted_total, s.converged, s.Rminus1_last = 0, False, np.inf
    s.learn_proposal_Rminus1_max = 30.0
    x = synthetic(rank)
    s.engine = FakeEngine(x, np.full(D, 2.0), rank)
    s.check_convergence_and_learn_proposal()
    # reduction helpers
    buf = np.arange(6, dtype=float).reshape(2, 3) * (rank + 1)
    dist.all_reduce_sum(buf)
    rows = dist.gather_rows(np.full((2, 2), float(rank)))
    prog = s.progress
    assert list(prog.columns) == ["N", "timestamp", "acceptance_rate", "Rminus1", "Rminus1_cl"]
    res = {"rank": rank, "Rminus1": float(prog.at[1, "Rminus1"]),
           "acc": float(prog.at[1, "acceptance_rate"]), "N": int(prog.at[1, "N"]),
           "new_cov": s.engine.new_cov.tolist(), "buf": buf.tolist(),
           "gathered": None if rows is None else [r.tolist() for r in rows]}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()


if __name__ == "__main__":
    main()
