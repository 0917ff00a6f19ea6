"""ctypes binding of libmcmc_hip.so (include/mcmc_hip.h) and a thin object wrapper.

The engine has NO CPU fallback: if the shared library is missing, or no gfx950 device is
usable, construction raises `EngineError` -- loudly, by design.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libmcmc_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)

OK, ERR_ARG, ERR_DEVICE, ERR_NOT_PD, ERR_STATE, ERR_STUCK = 0, -1, -2, -3, -4, -5


class EngineError(RuntimeError):
    """Any failure reported by libmcmc_hip (carries the C error code in `.code`)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class NotPositiveDefinite(EngineError):
    pass


class ChainStuck(EngineError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("n_walkers", C.c_int32), ("group_size", C.c_int32),
        ("device", C.c_int32), ("seed", C.c_uint64), ("walker_offset", C.c_uint32),
        ("burn_in", C.c_int32), ("temperature", C.c_double), ("proposal_scale", C.c_double),
        ("max_tries", C.c_double), ("emit_capacity", C.c_int32), ("flags", C.c_int32),
    ]


# every symbol include/mcmc_hip.h declares: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("mcmc_hip_version", C.c_char_p, []),
    ("mcmc_hip_last_error", C.c_char_p, [_H]),
    ("mcmc_hip_dim_supported", C.c_int, [C.c_int]),
    ("mcmc_hip_incremental_supported", C.c_int, [C.c_int32] * 6),
    ("mcmc_hip_create", C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    ("mcmc_hip_destroy", None, [_H]),
    ("mcmc_hip_set_prior", C.c_int, [_H, c_int32_p, c_double_p, c_double_p, c_int32_p]),
    ("mcmc_hip_set_target_gaussian_mixture", C.c_int,
     [_H, C.c_int32, c_double_p, c_double_p, c_double_p]),
    ("mcmc_hip_set_target_gaussian", C.c_int, [_H, c_double_p, c_double_p, C.c_int32]),
    ("mcmc_hip_set_target_one", C.c_int, [_H]),
    ("mcmc_hip_checkpoint_set_ring", C.c_int, [_H, C.c_int32, c_double_p, c_double_p, C.c_int32]),
    ("mcmc_hip_checkpoint_set_accepted", C.c_int, [_H, C.c_int64]),
    ("mcmc_hip_checkpoint_begin", C.c_int, [_H, C.c_int32, C.c_int64, C.c_double,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("mcmc_hip_checkpoint_solve", C.c_int, [_H, C.c_double, C.c_double]),
    ("mcmc_hip_checkpoint_fetch", C.c_int, [_H, c_double_p, c_double_p]),
    ("mcmc_hip_checkpoint_request_payload", C.c_int, [_H]),
    ("mcmc_hip_checkpoint_fetch_payload", C.c_int, [_H, c_double_p, C.c_int32]),
    ("mcmc_hip_stream_handle", C.c_uint64, [_H]),
    ("mcmc_hip_drain_samples_pinned", C.c_int, [_H, C.POINTER(c_double_p), c_int64_p]),
    ("mcmc_hip_set_drain_slots", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_set_emit_thin", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_get_thin_carry", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("mcmc_hip_set_thin_carry", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("mcmc_hip_set_target_binned_gaussian", C.c_int,
     [_H, C.c_int32, c_int32_p, C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int32,
      c_double_p, c_double_p, c_double_p, C.c_int32]),
    ("mcmc_hip_evaluate_binned", C.c_int,
     [_H, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    ("mcmc_hip_get_binned_constants", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("mcmc_hip_binned_kernel_times", C.c_int, [_H, c_double_p, c_int64_p, C.c_int32]),
    ("mcmc_hip_set_blocking", C.c_int, [_H, C.c_int32, c_int32_p, c_int32_p, c_int32_p,
                                        C.c_int32, C.c_int32]),
    ("mcmc_hip_cycle_length", C.c_int, [_H]),
    ("mcmc_hip_set_proposal_cov", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_get_proposal_cov", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_get_proposal_transform", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_evaluate", C.c_int, [_H, C.c_int32, c_double_p, c_double_p, c_double_p,
                                    c_double_p]),
    ("mcmc_hip_set_state", C.c_int, [_H, c_double_p, c_int32_p]),
    ("mcmc_hip_get_state", C.c_int, [_H, c_double_p, c_double_p, c_double_p, c_double_p,
                                     c_int32_p]),
    ("mcmc_hip_get_full_state", C.c_int, [_H, c_double_p, c_double_p, c_double_p, c_double_p,
                                          c_int32_p, c_int32_p, c_int32_p, c_int64_p,
                                          C.POINTER(C.c_uint64)]),
    ("mcmc_hip_set_full_state", C.c_int, [_H, c_double_p, c_double_p, c_double_p, c_double_p,
                                          c_int32_p, c_int32_p, c_int32_p, c_int64_p,
                                          C.c_uint64]),
    ("mcmc_hip_step", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_sync", C.c_int, [_H]),
    ("mcmc_hip_get_counters", C.c_int, [_H, c_int64_p]),
    ("mcmc_hip_drain_samples", C.c_int, [_H, c_double_p, C.c_int64, c_int64_p]),
    ("mcmc_hip_get_derived_constants", C.c_int, [_H, c_double_p, c_double_p, c_double_p,
                                                 c_double_p, c_double_p]),
    ("mcmc_hip_set_moment_shift", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_accumulate_moments", C.c_int, [_H]),
    ("mcmc_hip_read_moments", C.c_int, [_H, c_int64_p, c_double_p, c_double_p, C.c_int32]),
    ("mcmc_hip_set_moments", C.c_int, [_H, C.c_int64, c_double_p, c_double_p]),
    ("mcmc_hip_request_moments", C.c_int, [_H]),
    ("mcmc_hip_fetch_moments", C.c_int, [_H, c_int64_p, c_double_p, c_double_p, c_int64_p]),
    ("mcmc_hip_gelman_rubin", C.c_int, [C.c_int32, C.c_double, C.c_double, c_double_p,
                                        c_double_p, c_double_p, c_double_p, c_double_p]),
    ("mcmc_hip_enable_timing", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_last_step_kernel", C.c_char_p, [_H]),
    ("mcmc_hip_get_whitened", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_set_whitened", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_incremental_carries_modes", C.c_int, [_H]),
    ("mcmc_hip_incremental_carries_periodic", C.c_int, [_H]),
    ("mcmc_hip_get_mode_logdensities", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_set_mode_logdensities", C.c_int, [_H, c_double_p]),
    ("mcmc_hip_kernel_times", C.c_int, [_H, c_double_p, c_int64_p, C.c_int32]),
    ("mcmc_hip_bounds_configure", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_bounds_snapshot", C.c_int, [_H, C.c_int32]),
    ("mcmc_hip_bounds_statistics", C.c_int, [_H, C.c_int32, c_int32_p, C.c_double, c_double_p,
                                            c_double_p]),
    ("mcmc_hip_bounds_get_slot", C.c_int, [_H, C.c_int32, c_double_p]),
    ("mcmc_hip_bounds_set_slot", C.c_int, [_H, C.c_int32, c_double_p]),
    ("mcmc_hip_comm_version", C.c_char_p, []),
    ("mcmc_hip_comm_last_error", C.c_char_p, [_H]),
    ("mcmc_hip_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("mcmc_hip_comm_create", C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(_H)]),
    ("mcmc_hip_comm_destroy", None, [_H]),
    ("mcmc_hip_comm_rank", C.c_int, [_H]),
    ("mcmc_hip_comm_size", C.c_int, [_H]),
    ("mcmc_hip_comm_allreduce", C.c_int, [_H, c_double_p, C.c_int64, C.c_int32]),
    ("mcmc_hip_comm_allreduce_device", C.c_int, [_H, C.c_uint64, C.c_int64, C.c_int32, C.c_uint64]),
    ("mcmc_hip_comm_time_allreduce", C.c_int, [_H, C.c_int64, C.c_int32, c_double_p]),
    ("mcmc_hip_set_comm", C.c_int, [_H, _H]),
]

_lib = None


def load_library(path: str | None = None):
    """dlopen libmcmc_hip.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # MCMC_HIP_LIB (developer switch): an experiment build of the same ABI (tools/exp_*.sh)
    path = path or os.environ.get("MCMC_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise EngineError(ERR_DEVICE,
                          f"{path} not found: build it with `python -m cobaya_amd.build` "
                          "(hipcc, gfx950). mcmc_hip has no CPU fallback.")
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)  # AttributeError if the library does not export it
        except AttributeError:
            # developer A/B runs against an OLDER build of the library (MCMC_HIP_LIB=... with
            # MCMC_HIP_LIB_COMPAT=1, tools/gpu.sh ab): an entry point it predates answers "no" --
            # 0 for the predicates (`*_supported`, `*carries_*`), MCMC_HIP_ERR_ARG for every
            # setter and getter: a missing mcmc_hip_set_emit_thin must not look like success
            # (the sampler then skipped host thinning and rows came out un-thinned)
            if not (os.environ.get("MCMC_HIP_LIB_COMPAT") and os.environ.get("MCMC_HIP_LIB")):
                raise
            predicate = "carries_" in name or name.endswith("_supported")
            answer = 0 if predicate or restype is not C.c_int else ERR_ARG
            setattr(lib, name, C.CFUNCTYPE(restype, *argtypes)(lambda *a, _r=answer: _r))
            continue
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int32_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected array of shape {tuple(shape)}, got {a.shape}")
    return a


def incremental_supported(d, n_modes, n_periodic, n_drag, n_walkers, basis_group_size):
    """Does incremental evaluation (O(d) steps on carried whitened residuals) serve this model
    shape?  (mcmc_hip_incremental_supported: a pure function of the library, no device.)"""
    return bool(load_library().mcmc_hip_incremental_supported(
        int(d), int(n_modes), int(n_periodic), int(n_drag), int(n_walkers), int(basis_group_size)))


def gelman_rubin(n_chains, sum_N, sum_Ncov, sum_mean, sum_mm):
    """R-1 of means and the N-weighted mean of covariances from reduced sufficient
    statistics (mcmc.py:856-889).  Raises NotPositiveDefinite where the reference skips the
    update on LinAlgError (mcmc.py:870-887)."""
    lib = load_library()
    sum_mean = _f64(sum_mean)
    d = len(sum_mean)
    sum_Ncov, sum_mm = _f64(sum_Ncov, (d, d)), _f64(sum_mm, (d, d))
    R = C.c_double()
    W = np.empty((d, d))
    rc = lib.mcmc_hip_gelman_rubin(d, float(n_chains), float(sum_N), _dp(sum_Ncov),
                                   _dp(sum_mean), _dp(sum_mm), C.byref(R), _dp(W))
    if rc == ERR_NOT_PD:
        raise NotPositiveDefinite(rc, "Negative covariance eigenvectors / not enough "
                                      "information in the samples to compute R-1")
    if rc:
        raise EngineError(rc, "gelman_rubin: invalid arguments")
    return R.value, W


COMM_ID_BYTES = 128


class Communicator:
    """The walker shards' communicator inside libmcmc_hip.so (RCCL over xGMI; one handle of
    `mcmc_hip_comm_*`).  Creation is collective over all ranks."""

    _OPS = {"sum": 0, "max": 1}

    @staticmethod
    def unique_id() -> bytes:
        """Rank 0: the 128-byte id every rank's constructor needs (ncclGetUniqueId)."""
        lib = load_library()
        ident = (C.c_uint8 * COMM_ID_BYTES)()
        if lib.mcmc_hip_comm_unique_id(ident):
            raise EngineError(ERR_DEVICE, lib.mcmc_hip_comm_last_error(None).decode())
        return bytes(ident)

    def __init__(self, ident: bytes, rank: int, size: int, device: int):
        self._lib = load_library()
        if len(ident) != COMM_ID_BYTES:
            raise EngineError(ERR_ARG, f"the communicator id has {COMM_ID_BYTES} bytes")
        self._h = _H()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(ident)
        rc = self._lib.mcmc_hip_comm_create(buf, int(rank), int(size), int(device), C.byref(self._h))
        if rc:
            self._h = _H()
            raise EngineError(rc, "mcmc_hip_comm_create failed: "
                              + self._lib.mcmc_hip_comm_last_error(None).decode())
        self.rank, self.size, self.device = int(rank), int(size), int(device)
        self.version = self._lib.mcmc_hip_comm_version().decode()

    def _check(self, rc):
        if rc:
            raise EngineError(rc, self._lib.mcmc_hip_comm_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def allreduce(self, buf: np.ndarray, op="sum") -> np.ndarray:
        """In place over ranks, a float64 HOST buffer (staged through pinned memory; synchronous)."""
        flat = np.ascontiguousarray(buf, dtype=np.float64).reshape(-1)
        self._check(self._lib.mcmc_hip_comm_allreduce(self._h, _dp(flat), flat.size, self._OPS[op]))
        buf[...] = flat.reshape(buf.shape)
        return buf

    def allreduce_device(self, ptr: int, n: int, op="sum", stream: int = 0):
        """In place, n float64 at a device pointer, queued in order on `stream` (0: its own)."""
        self._check(self._lib.mcmc_hip_comm_allreduce_device(self._h, int(ptr), int(n), self._OPS[op],
                                                             int(stream)))

    def time_allreduce(self, n: int, reps: int = 20) -> float:
        """HIP-event microseconds per in-stream all-reduce of n float64 (collective)."""
        us = C.c_double()
        self._check(self._lib.mcmc_hip_comm_time_allreduce(self._h, int(n), int(reps), C.byref(us)))
        return us.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mcmc_hip_comm_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One walker ensemble on one MI355X (one handle of the C ABI)."""

    def __init__(self, d, n_walkers, group_size=64, device=0, seed=0, walker_offset=0,
                 burn_in=0, temperature=1.0, proposal_scale=2.4, max_tries=None,
                 emit_capacity=0, shared_basis=True, incremental=False, basis_group_size=None):
        self._lib = load_library()
        self.incremental = bool(incremental)
        # walkers sharing one Haar basis: group_size * 2**m (incremental mode only)
        bgs = int(basis_group_size or group_size)
        m = (bgs // int(group_size)).bit_length() - 1
        if bgs != int(group_size) << m:
            raise EngineError(ERR_ARG, "basis_group_size must be group_size times a power of two")
        self.basis_group_size = bgs
        self._h = _H()
        self.d, self.W, self.group_size = int(d), int(n_walkers), int(group_size)
        self.G = self.W // self.group_size if self.group_size else 0
        self.K = None
        self.walker_offset = int(walker_offset)
        cfg = Config(d=d, n_walkers=n_walkers, group_size=group_size, device=device,
                     seed=int(seed) & (2 ** 64 - 1), walker_offset=walker_offset,
                     burn_in=int(burn_in), temperature=float(temperature),
                     proposal_scale=float(proposal_scale),
                     max_tries=float(max_tries if max_tries is not None else 40 * d),
                     emit_capacity=int(emit_capacity), flags=(0 if shared_basis else 1) | (2 if incremental else 0) | (m << 8))
        self.cfg = cfg
        rc = self._lib.mcmc_hip_create(C.byref(cfg), C.byref(self._h))
        if rc:
            msg = self._lib.mcmc_hip_last_error(None).decode()
            self._h = _H()
            raise EngineError(rc, f"mcmc_hip_create failed: {msg}")

    # -- plumbing
    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.mcmc_hip_last_error(self._h).decode()
        cls = {ERR_NOT_PD: NotPositiveDefinite, ERR_STUCK: ChainStuck}.get(rc, EngineError)
        raise cls(rc, msg)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mcmc_hip_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- problem
    def set_prior(self, kinds, a, b, periodic=None):
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        a, b = _f64(a, (self.d,)), _f64(b, (self.d,))
        per = (np.zeros(self.d, np.int32) if periodic is None
               else np.ascontiguousarray(periodic, dtype=np.int32))
        self._check(self._lib.mcmc_hip_set_prior(self._h, _ip(kinds), _dp(a), _dp(b), _ip(per)))

    def set_target_gaussian_mixture(self, means, covs, weights=None):
        means = _f64(np.atleast_2d(means))
        K = len(means)
        covs = _f64(covs).reshape(K, self.d, self.d)
        w = None if weights is None else _f64(np.atleast_1d(weights), (K,))
        self._check(self._lib.mcmc_hip_set_target_gaussian_mixture(
            self._h, K, _dp(means), _dp(covs), _dp(w) if w is not None else None))
        self.K = K

    def set_target_gaussian(self, mean, cov, normalized=True):
        mean, cov = _f64(mean, (self.d,)), _f64(cov, (self.d, self.d))
        self._check(self._lib.mcmc_hip_set_target_gaussian(self._h, _dp(mean), _dp(cov),
                                                           int(bool(normalized))))
        self.K = 1

    def set_target_binned_gaussian(self, target, emulator, calib_index):
        """`target`: a `cobaya_amd.pliklite.BinnedGaussian` (what PlanckPlikLite.init_params
        leaves on the likelihood, planck_pliklite.py:32-141); `emulator`: a `LinearClEmulator`
        for the d - 1 other sampled parameters; `calib_index`: position of the calibration
        parameter among the sampled ones."""
        bins = np.ascontiguousarray(target.bin_table(), dtype=np.int32)
        n = len(bins)
        w = _f64(target.weights, (target.lmax + 1,))
        X, cov = _f64(target.X_data, (n,)), _f64(target.cov, (n, n))
        th = _f64(emulator.theta0, (self.d - 1,))
        D0 = _f64(emulator.D0, (3, target.lmax + 1))
        J = _f64(emulator.J, (3, target.lmax + 1, self.d - 1))
        self._check(self._lib.mcmc_hip_set_target_binned_gaussian(
            self._h, n, _ip(bins), int(target.lmax), _dp(w), _dp(X), _dp(cov), self.d - 1,
            _dp(th), _dp(D0), _dp(J), int(calib_index)))
        self.K = 0
        self.n_bins = n

    def binned_constants(self):
        n = self.n_bins
        Linv, Bc0, BJ = np.empty((n, n)), np.empty(n), np.empty((n, self.d - 1))
        self._check(self._lib.mcmc_hip_get_binned_constants(self._h, _dp(Linv), _dp(Bc0), _dp(BJ)))
        return {"Linv": Linv, "Bc0": Bc0, "BJ": BJ}

    def evaluate_binned(self, L0, cl, A):
        """chi2[n] = get_chi_squared(L0, cl[k, 0], cl[k, 1], cl[k, 2], A[k]) on the device."""
        cl = _f64(cl)
        n, three, n_ell = cl.shape
        assert three == 3
        A = _f64(np.broadcast_to(np.asarray(A, dtype=np.float64), (n,)))
        out = np.empty(n)
        self._check(self._lib.mcmc_hip_evaluate_binned(self._h, n, int(L0), n_ell, _dp(cl), _dp(A),
                                                       _dp(out)))
        return out

    def binned_kernel_times(self, reset=False):
        ms, n = np.zeros(3), np.zeros(3, np.int64)
        self._check(self._lib.mcmc_hip_binned_kernel_times(self._h, _dp(ms), n.ctypes.data_as(c_int64_p),
                                                           int(bool(reset))))
        return {"walker_ms": ms[0], "residual_ms": ms[1], "chi2_ms": ms[2],
                "launches": [int(v) for v in n]}

    def set_target_one(self):
        self._check(self._lib.mcmc_hip_set_target_one(self._h))
        self.K = 0

    def derived_constants(self):
        d, K = self.d, max(self.K or 0, 0)
        u = C.c_double()
        mls = np.empty(d)
        Linv, cn, w = np.empty((max(K, 1), d, d)), np.empty(max(K, 1)), np.empty(max(K, 1))
        self._check(self._lib.mcmc_hip_get_derived_constants(
            self._h, C.byref(u), _dp(mls), _dp(Linv) if K else None, _dp(cn) if K else None,
            _dp(w) if K else None))
        return {"uniform_logp": u.value, "mls": mls, "Linv": Linv[:K], "cnorm": cn[:K],
                "weight": w[:K]}

    def set_blocking(self, blocks, oversampling=None, drag_last_slow=-1, drag_steps=0):
        """Parameter blocks (lists of sampler indices, slow -> fast) with their oversampling
        factors (proposal.py:96-196); drag_last_slow >= 0 selects the dragging step."""
        sizes = np.array([len(b) for b in blocks], dtype=np.int32)
        over = np.array(oversampling if oversampling is not None else [1] * len(blocks),
                        dtype=np.int32)
        i_of_j = np.array([i for b in blocks for i in b], dtype=np.int32)
        if len(i_of_j) != self.d or len(over) != len(sizes):
            raise EngineError(ERR_ARG, "The blocks do not contain all the parameter indices.")
        self._check(self._lib.mcmc_hip_set_blocking(
            self._h, len(sizes), _ip(sizes), _ip(over), _ip(i_of_j), int(drag_last_slow),
            int(drag_steps)))

    def cycle_length(self):
        return int(self._lib.mcmc_hip_cycle_length(self._h))

    def set_proposal_cov(self, cov):
        cov = _f64(cov, (self.d, self.d))
        self._check(self._lib.mcmc_hip_set_proposal_cov(self._h, _dp(cov)))

    def get_proposal_cov(self):
        out = np.empty((self.d, self.d))
        self._check(self._lib.mcmc_hip_get_proposal_cov(self._h, _dp(out)))
        return out

    def get_proposal_transform(self):
        out = np.empty((self.d, self.d))
        self._check(self._lib.mcmc_hip_get_proposal_transform(self._h, _dp(out)))
        return out

    # -- evaluation / state
    def evaluate(self, x, derived=False):
        x = _f64(np.atleast_2d(x))
        n = len(x)
        lp, ll = np.empty(n), np.empty(n)
        der = np.empty((n, max(self.K or 0, 1) * self.d)) if derived else None
        self._check(self._lib.mcmc_hip_evaluate(self._h, n, _dp(x), _dp(lp), _dp(ll),
                                                _dp(der) if derived else None))
        return (lp, ll, der) if derived else (lp, ll)

    def set_state(self, x):
        x = _f64(x, (self.W, self.d))
        bad = C.c_int32()
        self._check(self._lib.mcmc_hip_set_state(self._h, _dp(x), C.byref(bad)))

    def get_state(self):
        x = np.empty((self.W, self.d))
        lpost, lpri, llik = np.empty(self.W), np.empty(self.W), np.empty(self.W)
        wt = np.empty(self.W, np.int32)
        self._check(self._lib.mcmc_hip_get_state(self._h, _dp(x), _dp(lpost), _dp(lpri),
                                                 _dp(llik), _ip(wt)))
        return {"x": x, "logpost": lpost, "logprior": lpri, "loglike": llik, "weight": wt}

    def get_full_state(self):
        """Everything needed to resume bit-identically (see mcmc_hip_get_full_state)."""
        W = self.W
        out = {"x": np.empty((W, self.d)), "logpost": np.empty(W), "logprior": np.empty(W),
               "loglike": np.empty(W), "weight": np.empty(W, np.int32),
               "prior_rej": np.empty(W, np.int32), "burn_left": np.empty(W, np.int32),
               "n_accept": np.empty(W, np.int64)}
        step = C.c_uint64()
        self._check(self._lib.mcmc_hip_get_full_state(
            self._h, _dp(out["x"]), _dp(out["logpost"]), _dp(out["logprior"]),
            _dp(out["loglike"]), _ip(out["weight"]), _ip(out["prior_rej"]),
            _ip(out["burn_left"]), out["n_accept"].ctypes.data_as(c_int64_p), C.byref(step)))
        out["step"] = np.uint64(step.value)
        if self.incremental:   # the carried whitened residual is part of the state
            out["y"] = np.empty((W, max(self.K or 1, 1) * self.d))
            self._check(self._lib.mcmc_hip_get_whitened(self._h, _dp(out["y"])))
            if self.carries_modes():
                # the carried log-density of every mode (mixtures on step_inc_mix_kernel); absent
                # until a step has formed them (they are then re-anchored on y by the next one)
                am = np.empty((W, self.K))
                if self._lib.mcmc_hip_get_mode_logdensities(self._h, _dp(am)) == 0:
                    out["amode"] = am
        if self.emit_thin > 1:   # thinned emission on the device: the per-walker remainders
            out["thin_carry"] = self.get_thin_carry()
        return out

    def set_full_state(self, st):
        W = self.W
        x = _f64(st["x"], (W, self.d))
        f = {k: _f64(st[k], (W,)) for k in ("logpost", "logprior", "loglike")}
        i = {k: np.ascontiguousarray(st[k], dtype=np.int32) for k in
             ("weight", "prior_rej", "burn_left")}
        na = np.ascontiguousarray(st["n_accept"], dtype=np.int64)
        self._check(self._lib.mcmc_hip_set_full_state(
            self._h, _dp(x), _dp(f["logpost"]), _dp(f["logprior"]), _dp(f["loglike"]),
            _ip(i["weight"]), _ip(i["prior_rej"]), _ip(i["burn_left"]),
            na.ctypes.data_as(c_int64_p), int(st["step"])))
        if self.incremental and "y" in st:
            y = _f64(st["y"], (W, max(self.K or 1, 1) * self.d))
            self._check(self._lib.mcmc_hip_set_whitened(self._h, _dp(y)))
            if "amode" in st and self.carries_modes():
                am = _f64(st["amode"], (W, self.K))
                self._check(self._lib.mcmc_hip_set_mode_logdensities(self._h, _dp(am)))
        if self.emit_thin > 1 and "thin_carry" in st:
            self.set_thin_carry(st["thin_carry"])

    def carries_modes(self):
        """Mixtures in incremental mode: does the step kernel this configuration selects carry the
        log-density of every mode (mcmc_hip_incremental_carries_modes)?  The oracle takes the rule
        from here (`oracle.cbind.Problem(carry_modes=...)`)."""
        return bool(self.incremental and self._lib.mcmc_hip_incremental_carries_modes(self._h))

    def carries_periodic(self):
        """One mode with periodic parameters in incremental mode: the rule of step_inc_kernel<.., periodic>
        (wrap only what leaves [lo, hi), carried log-likelihood) applies
        (mcmc_hip_incremental_carries_periodic); the oracle takes it from here."""
        return bool(self.incremental and self._lib.mcmc_hip_incremental_carries_periodic(self._h))

    # -- sampling
    def step(self, n_steps):
        self._check(self._lib.mcmc_hip_step(self._h, int(n_steps)))

    def sync(self):
        self._check(self._lib.mcmc_hip_sync(self._h))

    def counters(self):
        c = np.zeros(4, np.int64)
        self._check(self._lib.mcmc_hip_get_counters(self._h, c.ctypes.data_as(c_int64_p)))
        return {"steps": int(c[0]), "accepted": int(c[1]), "stuck": int(c[2]),
                "dropped_rows": int(c[3])}

    def drain_samples(self):
        n = C.c_int64()
        self._check(self._lib.mcmc_hip_drain_samples(self._h, None, 0, C.byref(n)))
        rows = np.empty((n.value, self.d + 5))
        if n.value:
            self._check(self._lib.mcmc_hip_drain_samples(self._h, _dp(rows), n.value,
                                                         C.byref(n)))
        return rows

    emit_thin = 1

    def set_emit_thin(self, thin):
        """Thinned emission on the device (mcmc_hip_set_emit_thin; collection.py:1373-1383): a
        walker's weights add up and a row is emitted when the sum reaches `thin`, with weight
        sum // thin.  Served by step_inc_kernel<.., emit>; other configurations refuse at their
        first step (EngineError): thin on the host then."""
        self._check(self._lib.mcmc_hip_set_emit_thin(self._h, int(thin)))
        self.emit_thin = int(thin)

    def get_thin_carry(self):
        out = np.empty(self.W, np.int32)
        self._check(self._lib.mcmc_hip_get_thin_carry(self._h, _ip(out)))
        return out

    def set_thin_carry(self, carry):
        c = np.ascontiguousarray(carry, dtype=np.int32)
        assert c.shape == (self.W,)
        self._check(self._lib.mcmc_hip_set_thin_carry(self._h, _ip(c)))

    def set_drain_slots(self, n_slots):
        self._check(self._lib.mcmc_hip_set_drain_slots(self._h, int(n_slots)))
        self.drain_slots = int(n_slots)

    drain_slots = 4

    def drain_samples_view(self):
        """The rows accumulated since the last drain as a READ-ONLY view of a pinned host slot
        the library owns: no host-side copy.  The view stays valid for `drain_slots - 1`
        further drains (then its slot is reused) and dies with the engine -- copy what must
        live longer."""
        p = c_double_p()
        n = C.c_int64()
        self._check(self._lib.mcmc_hip_drain_samples_pinned(self._h, C.byref(p), C.byref(n)))
        if not n.value:
            return np.empty((0, self.d + 5))
        rows = np.ctypeslib.as_array(p, shape=(n.value, self.d + 5))
        rows.flags.writeable = False
        return rows

    # -- moments
    def set_moment_shift(self, shift):
        shift = _f64(shift, (self.d,))
        self._check(self._lib.mcmc_hip_set_moment_shift(self._h, _dp(shift)))

    def accumulate_moments(self):
        self._check(self._lib.mcmc_hip_accumulate_moments(self._h))

    def read_moments(self, reset=False):
        n = C.c_int64()
        gs = np.empty((self.G, self.d))
        S = np.empty((self.d, self.d))
        self._check(self._lib.mcmc_hip_read_moments(self._h, C.byref(n), _dp(gs), _dp(S),
                                                    int(bool(reset))))
        return n.value, gs, S

    def set_moments(self, n_snapshots, group_sum, pooled_S):
        gs, S = _f64(group_sum, (self.G, self.d)), _f64(pooled_S, (self.d, self.d))
        self._check(self._lib.mcmc_hip_set_moments(self._h, int(n_snapshots), _dp(gs), _dp(S)))

    def request_moments(self):
        """Queue the read-out (and reset) of the moment accumulators and of the accept counter
        behind the work already in the stream; returns at once."""
        self._check(self._lib.mcmc_hip_request_moments(self._h))

    def fetch_moments(self):
        """(n_snapshots, group_sum[G][d], pooled_S[d][d], {"steps", "accepted"}) of the
        pending request; waits only for its copies, not for launches queued after it."""
        n = C.c_int64()
        gs = np.empty((self.G, self.d))
        S = np.empty((self.d, self.d))
        c = np.zeros(2, np.int64)
        self._check(self._lib.mcmc_hip_fetch_moments(self._h, C.byref(n), _dp(gs), _dp(S),
                                                     c.ctypes.data_as(c_int64_p)))
        return n.value, gs, S, {"steps": int(c[0]), "accepted": int(c[1])}

    # -- the checkpoint on the device
    def checkpoint_set_ring(self, intervals=(), min_capacity=16, first_index=0):
        """`intervals`: the (n_snapshots, group_sum[G][d], pooled_S[d][d]) the caller still
        holds, oldest first (none at the start of a run).  (`first_index`, the run's index of the
        first of them, is of no use to the device, which sums a window slot by slot.)"""
        n = len(intervals)
        gs = _f64(np.array([iv[1] for iv in intervals]).reshape(n, self.G * self.d)) if n else None
        S = _f64(np.array([iv[2] for iv in intervals]).reshape(n, self.d * self.d)) if n else None
        self._check(self._lib.mcmc_hip_checkpoint_set_ring(
            self._h, n, _dp(gs) if n else None, _dp(S) if n else None, int(min_capacity)))
        self.ckpt_capacity = 16
        while self.ckpt_capacity < max(n + 2, int(min_capacity)):
            self.ckpt_capacity *= 2

    def checkpoint_set_accepted(self, accepted):
        self._check(self._lib.mcmc_hip_checkpoint_set_accepted(self._h, int(accepted)))

    def checkpoint_begin(self, n_window_intervals, n_window_snapshots, steps_since):
        """-> (device pointer, length in doubles) of the buffer an all-reduce carries."""
        ptr, n = C.c_uint64(), C.c_int32()
        self._check(self._lib.mcmc_hip_checkpoint_begin(
            self._h, int(n_window_intervals), int(n_window_snapshots), float(steps_since),
            C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def checkpoint_solve(self, learn_lo, learn_hi):
        self._check(self._lib.mcmc_hip_checkpoint_solve(self._h, float(learn_lo), float(learn_hi)))

    def checkpoint_fetch(self):
        st = np.zeros(8)
        cov = np.empty((self.d, self.d))
        self._check(self._lib.mcmc_hip_checkpoint_fetch(self._h, _dp(st), _dp(cov)))
        return {"Rminus1_groups": st[0], "status": int(st[1]), "refreshed": bool(st[2]),
                "n_chains": st[3], "sum_N": st[4], "d_accepted": st[5], "d_steps": st[6],
                "accepted": st[7], "mean_of_covs": cov}

    def checkpoint_request_payload(self):
        """Instead of `checkpoint_solve`: queue the read-out of the (all-reduced) payload; the
        caller solves it on the host while the next launch runs."""
        self._check(self._lib.mcmc_hip_checkpoint_request_payload(self._h))

    def checkpoint_fetch_payload(self):
        """The payload of the pending request: [chains, sum N, accepted since, steps x walkers
        since, accepted | sum N cov | sum of chain means | sum of m m^T]."""
        n = 5 + 2 * self.d * self.d + self.d
        out = np.empty(n)
        self._check(self._lib.mcmc_hip_checkpoint_fetch_payload(self._h, _dp(out), n))
        return out

    # -- R-1 of the confidence bounds on the device
    BOUNDS_MAX_SLOTS = 64

    def bounds_configure(self, n_slots):
        """Ring of `n_slots` ensemble snapshots [slot][d][W] on the device (0 frees it)."""
        self._check(self._lib.mcmc_hip_bounds_configure(self._h, int(n_slots)))

    def bounds_snapshot(self, slot):
        """The current points -> ring slot, in stream order."""
        self._check(self._lib.mcmc_hip_bounds_snapshot(self._h, int(slot)))

    def bounds_statistics(self, slots, limfrac, want_bounds=False):
        """Per chain and parameter the lower / upper bound GetDist's `confidence(i, limfrac,
        upper)` gives for the samples in `slots`; returns the [1 + 4 d] sums over the chains of
        all ranks the statistic is formed from (and this rank's bounds [G][d][2] on request)."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        stats = np.empty(1 + 4 * self.d)
        b = np.empty((self.G, self.d, 2)) if want_bounds else None
        self._check(self._lib.mcmc_hip_bounds_statistics(
            self._h, len(sl), _ip(sl), float(limfrac), _dp(stats), _dp(b) if want_bounds else None))
        return (stats, b) if want_bounds else stats

    def bounds_get_slot(self, slot):
        x = np.empty((self.W, self.d))
        self._check(self._lib.mcmc_hip_bounds_get_slot(self._h, int(slot), _dp(x)))
        return x

    def bounds_set_slot(self, slot, x):
        x = _f64(x, (self.W, self.d))
        self._check(self._lib.mcmc_hip_bounds_set_slot(self._h, int(slot), _dp(x)))

    def stream_handle(self):
        return int(self._lib.mcmc_hip_stream_handle(self._h))

    def set_comm(self, comm):
        """Attach the shards' communicator: `checkpoint_begin` then all-reduces its payload over
        it, in place and in stream order.  The communicator must outlive the engine."""
        self._check(self._lib.mcmc_hip_set_comm(self._h, comm.handle if comm is not None else None))
        self._comm = comm
        self.comm_attached = comm is not None

    comm_attached = False

    # -- timing
    def enable_timing(self, on=True):
        self._check(self._lib.mcmc_hip_enable_timing(self._h, int(bool(on))))

    def last_step_kernel(self):
        """Name of the step kernel the last `step` launched, as its launcher reports it."""
        return self._lib.mcmc_hip_last_step_kernel(self._h).decode()

    def kernel_times(self, reset=False):
        ms = np.zeros(3)
        n = C.c_int64()
        self._check(self._lib.mcmc_hip_kernel_times(self._h, _dp(ms), C.byref(n),
                                                    int(bool(reset))))
        return {"step_ms": ms[0], "basis_ms": ms[1], "moments_ms": ms[2],
                "step_launches": n.value}
