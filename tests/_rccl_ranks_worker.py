"""Worker of tests/test_gpu_multirank.py: one rank of an N-GPU job (RANK / WORLD_SIZE / MASTER_*
in the environment, one process per GPU).  `dist.init_from_env()` forms the library's RCCL
communicator (mcmc_hip_comm_*, bootstrapped over a gloo group); the sampler shards the walkers by
rank and runs with the device checkpoint (the default for N > 1, `reduce`: window sums, payload and
ncclAllReduce queued in place on the engine's stream by mcmc_hip_checkpoint_begin; the host solves
the reduced payload beside the next launch).  Writes one JSON + one npz per rank."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def problem():
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "targets.npz"))
    mean, cov = t["mean_d30"], t["cov_d30"]
    names = [f"p{i}" for i in range(30)]
    return {"likelihood": {"gaussian_mixture": {"means": [mean], "covs": [cov]}},
            "params": {n: {"prior": {"min": 0, "max": 1},
                           "ref": {"dist": "norm", "loc": float(mean[i]),
                                   "scale": float(np.sqrt(cov[i, i]))}}
                       for i, n in enumerate(names)}}


def options(world, learn, walkers=2048):
    return {"seed": 9, "n_walkers": walkers, "group_size": 64, "steps_per_launch": "10d",
            "learn_every": "10d", "max_samples": 2.5e6 * world, "Rminus1_stop": 0.0,
            "proposal_scale": 2.4, "learn_proposal": bool(learn)}


def main():
    out_dir, learn = sys.argv[1], sys.argv[2] == "learn"
    from cobaya_amd import dist
    from cobaya_amd.model import ProblemSpec
    from cobaya_amd.sampler import MCMCHip
    dist.init_from_env()
    s = MCMCHip(options(dist.size(), learn), ProblemSpec.from_info(problem()))
    s.run()
    st = s.engine.get_full_state()
    res = {"rank": dist.rank(), "collective": dist.describe(), "device_checkpoint": bool(s._device_ckpt),
           "checkpoint_mode": s.device_checkpoint, "checkpoint_lag": int(s._ckpt_lag),
           "comm_attached": bool(s.engine.comm_attached), "device": int(s.engine.cfg.device),
           "walker_offset": int(s.engine.walker_offset), "steps": int(s.n_steps_raw),
           "progress": s.progress[["N", "acceptance_rate", "Rminus1"]].to_numpy().tolist(),
           "proposal_cov": s.proposer.get_covariance().tolist(),
           "allreduce_in_stream_us": dist.native().time_allreduce(2 * 900 + 35, 20)}
    np.savez(os.path.join(out_dir, f"state_rank{dist.rank()}.npz"), x=st["x"], logpost=st["logpost"],
             weight=st["weight"], n_accept=st["n_accept"])
    with open(os.path.join(out_dir, f"rank{dist.rank()}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    s.close()
    dist.shutdown()


if __name__ == "__main__":
    main()
