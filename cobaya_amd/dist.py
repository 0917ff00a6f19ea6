"""Process-group plumbing: one process per GPU.

The DATA PATH collective -- ONE all-reduce(sum) of the pooled sufficient statistics per
learn/convergence checkpoint, replacing the mpi4py gather/broadcast of the reference's mcmc
path (mcmc.py:791-793, 914, 1005-1007, 1021; mpi.py:178-191) -- runs inside
libmcmc_hip.so: RCCL over xGMI through the library's own communicator
(`mcmc_hip_comm_*`, include/mcmc_hip.h), in place on the engine's stream for the device
checkpoint, staged through pinned memory for host buffers.  PyTorch is NOT on that path.

What is left to `torch.distributed` is the bootstrap -- a `gloo` group over
MASTER_ADDR/MASTER_PORT that carries the 128-byte RCCL id from rank 0 to the others and the
final host-side gather of sample rows -- and the CPU stand-in of the collective for the
world_size > 1 tests (`MCMC_HIP_BACKEND=gloo`, also what several ranks sharing ONE GPU must
use: RCCL refuses two ranks on a device).
"""
from __future__ import annotations

import os

import numpy as np

_comm = None        # cobaya_amd.engine.Communicator: the library's RCCL communicator
_rccl_error = None  # why the communicator could not be created (then: the gloo stand-in)
_own_group = False  # the torch.distributed group was created here (shutdown destroys it)


def _td():
    import torch.distributed as td
    return td


def is_initialized() -> bool:
    """A torch.distributed group exists (bootstrap / gloo stand-in)."""
    try:
        td = _td()
        return td.is_available() and td.is_initialized()
    except Exception:
        return False


def native():
    """The library's communicator, or None (single process, or the gloo stand-in)."""
    return _comm


def rank() -> int:
    if _comm is not None:
        return _comm.rank
    return _td().get_rank() if is_initialized() else 0


def size() -> int:
    if _comm is not None:
        return _comm.size
    return _td().get_world_size() if is_initialized() else 1


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def _device_count() -> int:
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def default_device() -> int:
    """HIP ordinal for this process: LOCAL_RANK, wrapped onto the visible devices (ranks may
    share a GPU in tests -- with the gloo stand-in only)."""
    if _comm is not None:
        return _comm.device
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "LOCAL_RANK" not in os.environ:
        return 0  # single process: no need to import torch at all
    n = _device_count()
    return local_rank() % n if n else local_rank()


def _exchange_id(r, n):
    """The 128-byte RCCL id from rank 0 to every rank through the bootstrap group.  Rank 0 ALWAYS
    broadcasts -- the id, or the reason it has none (librccl not loadable, ncclGetUniqueId
    failed) -- so that no rank is left waiting in the broadcast; every rank then raises the same
    error (a rank 0 that raised before the broadcast hung the whole job)."""
    from .engine import Communicator
    msg = [None]
    if r == 0:
        try:
            msg = [(Communicator.unique_id(), None)]
        except Exception as e:
            msg = [(None, f"{type(e).__name__}: {e}")]
    if n > 1:
        _td().broadcast_object_list(msg, src=0)
    ident, err = msg[0]
    if err is not None:
        raise RuntimeError(f"rank 0 could not create the RCCL id ({err})")
    return ident


def init_native_comm(rank_=None, size_=None, device=None):
    """Create the library's RCCL communicator (collective over all ranks).  The 128-byte id
    travels from rank 0 through the bootstrap group; a world of one needs no group at all."""
    global _comm
    from .engine import Communicator
    if _comm is not None:
        return _comm
    r = rank() if rank_ is None else int(rank_)
    n = size() if size_ is None else int(size_)
    dev = default_device() if device is None else int(device)
    _comm = Communicator(_exchange_id(r, n), r, n, dev)
    return _comm


def _create_comm_guarded(timeout_s):
    """The communicator with a deadline: ncclCommInitRank blocks until every rank has joined
    and cannot be cancelled -- a rank whose peers never arrive (a bootstrap interface RCCL cannot
    use, a rank that died) would hang the job.  The id is exchanged HERE, on the calling thread
    (so no bootstrap collective is ever pending on another thread when the agreement that follows
    runs on the same group); only ncclCommInitRank runs in a helper thread (ctypes releases the
    GIL).  Past the deadline this rank reports failure, and the agreement sends every rank to the
    gloo stand-in.  The communicator is PUBLISHED by the calling thread only: a helper that
    finishes after it was abandoned destroys what it made.  Returns (communicator or None, None
    or the reason)."""
    import threading
    from .engine import Communicator
    r, n, dev = rank(), size(), default_device()
    try:
        ident = _exchange_id(r, n)
    except Exception as e:
        return None, str(e)
    box, lock = {}, threading.Lock()

    def work():
        try:
            c = Communicator(ident, r, n, dev)
        except Exception as e:      # EngineError with RCCL's message
            with lock:
                box["err"] = f"{type(e).__name__}: {e}"
            return
        with lock:
            if box.get("abandoned"):
                c.close()           # nobody will ever use it: the job runs on gloo
            else:
                box["comm"] = c
    t = threading.Thread(target=work, name="mcmc_hip-rccl-init", daemon=True)
    t.start()
    t.join(timeout_s)
    with lock:
        if "comm" in box:
            return box["comm"], None
        if "err" in box:
            return None, box["err"]
        box["abandoned"] = True
    return None, (f"timed out after {timeout_s:g} s in ncclCommInitRank (MCMC_HIP_RCCL_TIMEOUT; "
                  f"NCCL_SOCKET_IFNAME selects the bootstrap interface)")


def init_from_env(backend=None):
    """From RANK/WORLD_SIZE/MASTER_* (torch.distributed.run or bench.py's own launcher) when
    WORLD_SIZE > 1; a no-op for single-process runs.  `backend` (or $MCMC_HIP_BACKEND):
    "rccl" (default wherever a GPU per rank is visible; "nccl" is accepted as its alias): the
    library's communicator; "gloo": the CPU stand-in."""
    global _own_group, _comm, _rccl_error
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or _comm is not None:
        return
    if backend is None:
        backend = os.environ.get("MCMC_HIP_BACKEND")
    if backend is None:
        # several ranks on ONE device cannot form an RCCL communicator
        backend = "rccl" if _device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo"
    if backend == "nccl":
        backend = "rccl"
    if backend not in ("rccl", "gloo"):
        raise ValueError(f"unknown collective backend {backend!r} (rccl or gloo)")
    if not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _td().init_process_group(backend="gloo")   # bootstrap + host-side row gather only
        _own_group = True
    if backend == "rccl":
        # Creation is collective; a rank that fails (no RCCL, a refused device, ...) must not
        # leave the others inside ncclCommInitRank with a different idea of the backend: every
        # rank reports, and unless ALL succeeded the job falls back -- loudly -- to the gloo
        # stand-in for the collective (the kernels are unaffected; `describe()` says what runs).
        comm, err = _create_comm_guarded(float(os.environ.get("MCMC_HIP_RCCL_TIMEOUT", "300")))
        import torch
        flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64)
        _td().all_reduce(flag, op=_td().ReduceOp.MIN)
        if float(flag[0]) >= 0.5:
            _comm = comm        # published here, by the thread that agreed on it
        else:
            if comm is not None:
                comm.close()
            _comm = None
            _rccl_error = err or "another rank could not create its RCCL communicator"
            import sys
            print(f"[mcmc_hip] WARNING: the RCCL communicator could not be created on every rank "
                  f"({_rccl_error}); the checkpoint's all-reduce uses the gloo stand-in",
                  file=sys.stderr)


def shutdown():
    """Destroy the communicator and a group created here (idempotent)."""
    global _comm, _own_group
    if _comm is not None:
        _comm.close()
        _comm = None
    if _own_group and is_initialized():
        _td().destroy_process_group()
    _own_group = False


def attach(engine):
    """Hand the communicator to an engine: its device checkpoint then all-reduces in place, in
    stream order (mcmc_hip_set_comm).  Nothing to do for one process or the gloo stand-in."""
    if _comm is not None and hasattr(engine, "set_comm"):
        engine.set_comm(_comm)
        return True
    return False


def _all_reduce(buf: np.ndarray, op: str) -> np.ndarray:
    if _comm is not None:
        return _comm.allreduce(buf, op)
    if not is_initialized():
        return buf
    import torch
    td = _td()
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    red = td.ReduceOp.SUM if op == "sum" else td.ReduceOp.MAX
    if str(td.get_backend()) == "nccl":     # a group the host program made with torch's RCCL
        g = t.cuda(local_rank())
        td.all_reduce(g, op=red)
        t = g.cpu()
    else:
        td.all_reduce(t, op=red)
    buf[...] = t.numpy().reshape(buf.shape)
    return buf


def all_reduce_sum(buf: np.ndarray) -> np.ndarray:
    """In-place sum over ranks of a float64 host buffer: RCCL through the library's
    communicator, gloo for the CPU stand-in, the identity without a group."""
    return _all_reduce(buf, "sum")


def all_reduce_max(buf: np.ndarray) -> np.ndarray:
    return _all_reduce(buf, "max")


def device_collective() -> bool:
    """True if an all-reduce can run on device memory in place: one process, or the library's
    communicator."""
    return _comm is not None or not is_initialized()


def all_reduce_sum_device(ptr: int, n: int, stream_handle: int):
    """In-place RCCL all-reduce(sum) of n float64 at the device pointer `ptr`, queued IN ORDER on
    the HIP stream `stream_handle`.  (The sampler does not call this for an engine the
    communicator is attached to: `mcmc_hip_checkpoint_begin` queues the reduction itself.)"""
    if _comm is None:
        if is_initialized():
            raise RuntimeError("all_reduce_sum_device needs the library's RCCL communicator")
        return
    _comm.allreduce_device(ptr, n, "sum", stream_handle)


def describe():
    """What the collective layer really is in this process, measured rather than assumed:
    backend, world size and the number of ranks an all-reduce of ones actually summed."""
    if _comm is None and not is_initialized():
        return {"backend": None, "world_size": 1, "nranks_seen": 1}
    seen = all_reduce_sum(np.ones(1))
    out = {"backend": "nccl" if _comm is not None else str(_td().get_backend()),
           "world_size": size(), "nranks_seen": int(round(float(seen[0])))}
    if _comm is not None:
        out["library"] = f"libmcmc_hip.so ({_comm.version}, in-stream ncclAllReduce)"
        out["bootstrap"] = "torch.distributed gloo" if is_initialized() else None
    if _rccl_error:
        out["rccl_error"] = _rccl_error
    return out


def barrier():
    if _comm is not None:
        if _comm.size > 1:
            _comm.allreduce(np.zeros(1), "sum")
    elif size() > 1:
        td = _td()
        if str(td.get_backend()) == "nccl":   # name the device: RCCL otherwise guesses (and warns)
            td.barrier(device_ids=[local_rank()])
        else:
            td.barrier()


def gather_rows(rows: np.ndarray):
    """All ranks' 2-d row blocks on rank 0 (host-side concatenation of per-GPU sample
    buffers: mcmc.py:1136-1183); None elsewhere.  Host data: through the bootstrap group."""
    if size() == 1:
        return [rows]
    out = [None] * size() if rank() == 0 else None
    _td().gather_object(rows, out, dst=0)
    return out
