"""Process-group plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for the CPU tests).  Replaces the mpi4py calls of the reference's
mcmc path: the only data-path collective is ONE all-reduce(sum) of the pooled
sufficient statistics per learn/convergence checkpoint (mcmc.py:791-793, 914, 1005, 1021).
"""
from __future__ import annotations

import os

import numpy as np


def _td():
    import torch.distributed as td
    return td


def is_initialized() -> bool:
    try:
        td = _td()
        return td.is_available() and td.is_initialized()
    except Exception:
        return False


def rank() -> int:
    return _td().get_rank() if is_initialized() else 0


def size() -> int:
    return _td().get_world_size() if is_initialized() else 1


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def default_device() -> int:
    """HIP ordinal for this process: LOCAL_RANK, wrapped onto the visible devices (ranks may
    share a GPU in tests)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "LOCAL_RANK" not in os.environ:
        return 0  # single process: no need to import torch at all
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return local_rank() % n if n else local_rank()


def init_from_env(backend=None):
    """Initialise the default group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run)
    when WORLD_SIZE > 1; a no-op for single-process runs."""
    if is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    import torch
    td = _td()
    if backend is None:
        # MCMC_HIP_BACKEND=gloo forces the CPU collective (e.g. several ranks sharing one GPU)
        backend = os.environ.get("MCMC_HIP_BACKEND") or (
            "nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    td.init_process_group(backend=backend)


def all_reduce_sum(buf: np.ndarray) -> np.ndarray:
    """In-place sum over ranks of a float64 host buffer; through the GPU (RCCL) when the
    group's backend is nccl, on the CPU for gloo.  Without a process group: the identity.
    A group of ONE rank still goes through its backend (so that a single-GPU test executes
    the very RCCL path an 8-GPU job takes)."""
    if not is_initialized():
        return buf
    import torch
    td = _td()
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    if td.get_backend() == "nccl":
        g = t.cuda(local_rank())
        td.all_reduce(g, op=td.ReduceOp.SUM)
        t = g.cpu()
    else:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    buf[...] = t.numpy().reshape(buf.shape)
    return buf


def device_collective() -> bool:
    """True if an all-reduce can run on device memory in place (no process group, or RCCL)."""
    return not is_initialized() or str(_td().get_backend()) == "nccl"


class _DeviceBuffer:
    """A device allocation of another library seen through `__cuda_array_interface__`."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def all_reduce_sum_device(ptr: int, n: int, stream_handle: int):
    """In-place RCCL all-reduce(sum) of n float64 at the device pointer `ptr`, queued IN ORDER on
    the HIP stream `stream_handle` (the engine's): what precedes it on that stream has written the
    buffer, what follows reads the reduced one -- no host bounce, no synchronisation.  Without a
    process group: nothing to do."""
    if not is_initialized():
        return
    import torch
    td = _td()
    if str(td.get_backend()) != "nccl":
        raise RuntimeError("all_reduce_sum_device needs the nccl (RCCL) backend")
    dev = torch.device("cuda", local_rank())
    t = torch.as_tensor(_DeviceBuffer(ptr, n), device=dev)
    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream_handle), device=dev)):
        td.all_reduce(t, op=td.ReduceOp.SUM)


def describe():
    """What the collective layer really is in this process, measured rather than assumed:
    backend, world size and the number of ranks an all-reduce of ones actually summed."""
    if not is_initialized():
        return {"backend": None, "world_size": 1, "nranks_seen": 1}
    seen = all_reduce_sum(np.ones(1))
    return {"backend": str(_td().get_backend()), "world_size": size(),
            "nranks_seen": int(round(float(seen[0])))}


def barrier():
    if size() > 1:
        td = _td()
        if td.get_backend() == "nccl":   # name the device: RCCL otherwise guesses (and warns)
            td.barrier(device_ids=[local_rank()])
        else:
            td.barrier()


def gather_rows(rows: np.ndarray):
    """All ranks' 2-d row blocks on rank 0 (host-side concatenation of per-GPU sample
    buffers: mcmc.py:1136-1183); None elsewhere."""
    if size() == 1:
        return [rows]
    out = [None] * size() if rank() == 0 else None
    _td().gather_object(rows, out, dst=0)
    return out
