"""Worker of tests/test_gpu_sampler.py::test_rccl_path_with_one_rank: creates the library's RCCL
communicator (mcmc_hip_comm_*, a world of one on cuda:0) and runs the sampler's collective layer
and full learn/convergence checkpoints of the real engine through it.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def run_checkpoints(use_group, device_checkpoint=False):
    from cobaya_amd import dist
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    if use_group == "torch":
        # the loading order of a multi-rank job (dist.init_from_env): PyTorch first -- its gloo
        # group is the bootstrap, and importing it maps the RCCL build that ships with it --, then
        # the library binds RCCL at run time and creates its communicator
        import torch
        import torch.distributed as td
        td.init_process_group("gloo", rank=0, world_size=1,
                              init_method=f"tcp://127.0.0.1:{int(sys.argv[2])}")
        assert torch.cuda.device_count() >= 1
        dist.init_native_comm(0, 1, 0)
    elif use_group:
        # the library's own RCCL communicator, a world of one on cuda:0 -- no PyTorch involved
        dist.init_native_comm(0, 1, 0)
        assert "torch" not in sys.modules
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "targets.npz"))
    mean, cov = t["mean_d30"], t["cov_d30"]
    names = [f"p{i}" for i in range(30)]
    info = {"likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov]}},
            "params": {n: {"prior": {"min": 0, "max": 1},
                           "ref": {"dist": "norm", "loc": float(mean[i]),
                                   "scale": float(np.sqrt(cov[i, i]))}}
                       for i, n in enumerate(names)}}
    s = MCMCHip({"seed": 9, "n_walkers": 2048, "group_size": 64, "steps_per_launch": "10d",
                 "learn_every": "10d", "max_samples": 6e6, "Rminus1_stop": 0.0,
                 "proposal_scale": 2.4, "device_checkpoint": device_checkpoint},
                ProblemSpec.from_info(info))
    s.run()
    out = {"collective": dist.describe(), "progress": s.progress[["N", "acceptance_rate",
                                                                  "Rminus1"]].to_numpy().tolist(),
           "proposal_cov": s.proposer.get_covariance().tolist(),
           "x_sum": float(s.engine.get_state()["x"].sum()),
           # R-1 of the bounds: its sums cross the communicator in stream order as well
           "Rminus1_cl": s._rminus1_of_bounds(s.proposer.get_covariance())}
    buf = np.arange(12, dtype=float).reshape(3, 4)
    out["allreduce_identity"] = bool(np.array_equal(dist.all_reduce_sum(buf.copy()), buf))
    s.close()
    if use_group:
        dist.shutdown()
    return out


if __name__ == "__main__":
    where = sys.argv[3] if len(sys.argv) > 3 else "host"
    group = "torch" if sys.argv[1] == "nccl+torch" else sys.argv[1] == "nccl"
    print("RESULT " + json.dumps(run_checkpoints(
        group, {"device": True, "reduce": "reduce", "host": False}[where])))
