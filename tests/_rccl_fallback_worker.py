"""Worker of tests/test_distributed_gloo.py::test_rccl_fallback_*: one rank of a `gloo` bootstrap
group asking for the RCCL backend on a box where the communicator cannot be made.  argv[1]:
"id" -- rank 0's ncclGetUniqueId raises (librccl not loadable); "init" -- the id exists but
ncclCommInitRank never returns on rank 1 (a peer that cannot reach the bootstrap interface) and
succeeds late, after the deadline.  Either way every rank must leave `init_from_env` with the
gloo stand-in and all-reduce correctly."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cobaya_amd import dist, engine  # noqa: E402

mode = sys.argv[1]
rank = int(os.environ["RANK"])
closed = []


class FakeComm:
    """Stands where cobaya_amd.engine.Communicator is: no GPU in the CPU container."""
    version = "fake"

    def __init__(self, ident, r, n, dev):
        assert ident == b"i" * 128
        self.rank, self.size, self.device = r, n, dev
        if mode == "init" and r == 1:
            time.sleep(6.0)          # joins long after the deadline (MCMC_HIP_RCCL_TIMEOUT=2)

    @staticmethod
    def unique_id():
        if mode == "id":
            raise OSError("librccl.so.1: cannot open shared object file")
        return b"i" * 128

    def close(self):
        closed.append(self.rank)


engine.Communicator = FakeComm
t0 = time.time()
dist.init_from_env("rccl")
took = time.time() - t0
assert dist.native() is None, "the job must have fallen back to the gloo stand-in"
assert dist.size() == 2 and dist.rank() == rank
buf = dist.all_reduce_sum(np.array([1.0 + rank, 10.0]))
assert buf.tolist() == [3.0, 20.0], buf
desc = dist.describe()
assert desc["backend"] == "gloo" and desc["nranks_seen"] == 2 and desc["rccl_error"]
if mode == "init":
    time.sleep(5.0)                  # the late communicator arrives ...
    assert dist.native() is None     # ... and is NOT published behind the main thread's back
    if rank == 1:
        assert closed == [1], closed  # the abandoned helper destroyed what it made
    else:
        assert closed == [0], closed  # rank 0's (successful) one was closed by the agreement
dist.barrier()
dist.shutdown()
print(f"rank {rank} ok after {took:.1f} s: {desc['rccl_error']}")
