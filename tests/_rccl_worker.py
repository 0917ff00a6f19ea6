"""Worker of tests/test_gpu_sampler.py::test_rccl_path_with_one_rank: initialises
torch.distributed with backend `nccl` (= RCCL on ROCm), world_size 1, on cuda:0 and runs the
sampler's collective layer and a full learn/convergence checkpoint of the real engine through
it.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def run_checkpoints(use_group, device_checkpoint=False):
    from cobaya_amd import dist
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    if use_group:
        import torch
        import torch.distributed as td
        torch.cuda.set_device(0)
        td.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2],
                              world_size=1, rank=0)
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
                 "proposal_scale": 2.4, "device_checkpoint": bool(device_checkpoint)},
                ProblemSpec.from_info(info))
    s.run()
    out = {"collective": dist.describe(), "progress": s.progress[["N", "acceptance_rate",
                                                                  "Rminus1"]].to_numpy().tolist(),
           "proposal_cov": s.proposer.get_covariance().tolist(),
           "x_sum": float(s.engine.get_state()["x"].sum())}
    buf = np.arange(12, dtype=float).reshape(3, 4)
    out["allreduce_identity"] = bool(np.array_equal(dist.all_reduce_sum(buf.copy()), buf))
    s.close()
    if use_group:
        td.destroy_process_group()
    return out


if __name__ == "__main__":
    print("RESULT " + json.dumps(run_checkpoints(sys.argv[1] == "nccl",
                                                 len(sys.argv) > 3 and sys.argv[3] == "device")))
