"""ORACLE (test infrastructure): ctypes binding of oracle/mcmc_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


def build():
    """(Re)compile the oracle with gcc; cheap, incremental through make."""
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


class _Problem(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("n_modes", C.c_int32), ("group_size", C.c_int32),
        ("has_periodic", C.c_int32), ("seed", C.c_uint64), ("temperature", C.c_double),
        ("max_tries", C.c_double),
        ("kind", c_int32_p), ("lo", c_double_p), ("hi", c_double_p), ("loc", c_double_p),
        ("scale", c_double_p), ("mls", c_double_p), ("periodic", c_int32_p),
        ("uniform_logp", C.c_double),
        ("mean", c_double_p), ("Linv", c_double_p), ("cnorm", c_double_p),
        ("weight", c_double_p), ("T", c_double_p), ("blocking", C.c_void_p),
        ("incremental", C.c_int32), ("refresh_every", C.c_int32),
        ("paired_variates", C.c_int32), ("carry_modes", C.c_int32), ("carry_periodic", C.c_int32),
        ("binned", C.c_void_p),
    ]


class _Binned(C.Structure):
    _fields_ = [
        ("n_bins", C.c_int32), ("lmax", C.c_int32), ("n_lin", C.c_int32), ("calib", C.c_int32),
        ("bins", c_int32_p), ("weights", c_double_p), ("X", c_double_p), ("Linv", c_double_p),
        ("theta0", c_double_p), ("D0", c_double_p), ("J", c_double_p),
        ("Bc0", c_double_p), ("BJ", c_double_p),
    ]


class _Blocking(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_int32), ("size", c_int32_p), ("oversample", c_int32_p),
        ("i_of_j", c_int32_p), ("drag_last_slow", C.c_int32), ("drag_steps", C.c_int32),
    ]


class _State(C.Structure):
    _fields_ = [
        ("x", c_double_p), ("logprior", c_double_p), ("loglike", c_double_p),
        ("logpost", c_double_p), ("weight", c_int32_p), ("prior_rej", c_int32_p),
        ("burn_left", c_int32_p), ("n_accept", c_int64_p), ("stuck", c_int32_p),
        ("rows", c_double_p), ("n_rows", c_int32_p), ("row_cap", C.c_int32),
        ("y", c_double_p), ("amode", c_double_p),
        ("thin", C.c_int32), ("thin_acc", c_int32_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        name = "libmcmc_oracle_fma.so" if _cpu_has_fma() else "libmcmc_oracle.so"
        path = os.path.join(BUILD, name)
        src = os.path.join(HERE, "mcmc_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()   # never test against a library older than its source
        L = C.CDLL(path)
        L.orc_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.orc_dlog.restype = C.c_double
        L.orc_dlog.argtypes = [C.c_double]
        L.orc_dexp.restype = C.c_double
        L.orc_dexp.argtypes = [C.c_double]
        L.orc_dlog_tab.restype = C.c_double
        L.orc_dlog_tab.argtypes = [C.c_double]
        L.orc_dexp_tab.restype = C.c_double
        L.orc_dexp_tab.argtypes = [C.c_double]
        L.orc_neg_log_short.restype = C.c_double
        L.orc_neg_log_short.argtypes = [C.c_uint32, C.c_int32]
        L.orc_pair_variates.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, c_double_p, c_double_p]
        L.orc_find_short_tail.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                          C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.orc_find_short_tail.restype = C.c_int
        L.orc_sincos2pi.argtypes = [C.c_uint64, c_double_p, c_double_p]
        L.orc_haar_from_normals.argtypes = [C.c_int, c_double_p, c_double_p]
        L.orc_basis.argtypes = [C.POINTER(_Problem), C.c_uint32, C.c_uint32, c_double_p]
        L.orc_evaluate.argtypes = [C.POINTER(_Problem), C.c_int, c_double_p, c_double_p,
                                   c_double_p, c_double_p]
        L.orc_step_injected.restype = C.c_int
        L.orc_step_injected.argtypes = [C.POINTER(_Problem), C.POINTER(_State), c_double_p,
                                        C.c_double]
        L.orc_step_injected_delta.restype = C.c_int
        L.orc_step_injected_delta.argtypes = [C.POINTER(_Problem), C.POINTER(_State),
                                              c_double_p, C.c_double]
        L.orc_drag_injected.restype = C.c_int
        L.orc_drag_injected.argtypes = [C.POINTER(_Problem), C.POINTER(_State), c_double_p,
                                        c_double_p, c_double_p]
        L.orc_run.restype = C.c_int64
        L.orc_run.argtypes = [C.POINTER(_Problem), C.POINTER(_State), C.c_int, C.c_uint32,
                              C.c_uint64, C.c_int, C.c_int]
        L.orc_moments.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p,
                                  c_double_p, c_double_p]
        L.orc_block_slots.restype = C.c_int
        L.orc_block_slots.argtypes = [C.POINTER(_Problem), C.c_int, c_int32_p]
        L.orc_block_schedule.restype = C.c_int
        L.orc_block_schedule.argtypes = [C.POINTER(_Problem), C.c_uint32, C.c_uint32, C.c_int,
                                         c_int32_p, c_int32_p, c_int32_p]
        L.orc_basis_blocked.restype = C.c_int
        L.orc_basis_blocked.argtypes = [C.POINTER(_Problem), C.c_uint32, C.c_uint32, C.c_int,
                                        c_double_p, c_int32_p]
        L.orc_whiten.argtypes = [C.POINTER(_Problem), c_double_p, c_double_p]
        L.orc_whiten_directions.argtypes = [C.POINTER(_Problem), C.c_int, c_double_p, c_double_p]
        L.orc_direction_norms.argtypes = [C.POINTER(_Problem), C.c_int, c_double_p, c_double_p]
        L.orc_anchor_modes.argtypes = [C.POINTER(_Problem), C.POINTER(_State), C.c_int]
        L.orc_anchor_modes.restype = None
        L.orc_max_threads.restype = C.c_int
        L.orc_binned_chi2_of_delta.restype = C.c_double
        L.orc_binned_chi2_of_delta.argtypes = [C.POINTER(_Binned), c_double_p]
        L.orc_binned_delta.argtypes = [C.POINTER(_Binned), c_double_p, c_double_p]
        L.orc_binned_collapse.argtypes = [C.POINTER(_Binned), c_double_p, c_double_p]
        L.orc_binned_chi2_of_cl.argtypes = [C.POINTER(_Binned), C.c_int, C.c_int, C.c_int,
                                            c_double_p, c_double_p, c_double_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int32_p)


def philox(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(k0, k1, c0, c1, c2, c3, out)
    return [int(v) for v in out]


def dlog(x):
    return lib().orc_dlog(float(x))


def dexp(x):
    return lib().orc_dexp(float(x))


def dlog_tab(x):
    """The table-driven log of the incremental mixtures' log-sum-exp (orc_dlog_tab)."""
    return lib().orc_dlog_tab(float(x))


def dexp_tab(x):
    return lib().orc_dexp_tab(float(x))


def neg_log_short(n, b):
    """-log(n 2^-b), n odd < 2^29: the logarithm of the paired variates."""
    return lib().orc_neg_log_short(int(n), int(b))


def pair_variates(seed, gid, step):
    """(r, E_a) of walker `gid` at `step` on the paired stream (walker_variates_pair)."""
    r, e = C.c_double(), C.c_double()
    lib().orc_pair_variates(int(seed), int(gid), int(step), C.byref(r), C.byref(e))
    return r.value, e.value


def find_short_tail(seed, gid0, n_walkers, step0, n_steps, which):
    """First (walker, step) in the given ranges whose short radial (which = 0) / accept (which = 1)
    uniform fell into the lowest bin, or None."""
    g, st = C.c_uint32(), C.c_uint64()
    if lib().orc_find_short_tail(int(seed), int(gid0), int(n_walkers), int(step0), int(n_steps),
                                 int(which), C.byref(g), C.byref(st)):
        return g.value, st.value
    return None


def sincos2pi(k):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos2pi(int(k), C.byref(s), C.byref(c))
    return s.value, c.value


def haar_from_normals(d, z):
    z = np.ascontiguousarray(z, dtype=np.float64)
    H = np.empty((d, d))
    lib().orc_haar_from_normals(d, _dp(z), _dp(H))
    return H


def proposal_transform(cov, scale):
    """T = scale * diag(std) * chol(corr)  (proposal.py:256-260 for one block)."""
    cov = np.asarray(cov, dtype=np.float64)
    std = np.sqrt(np.diag(cov))
    corr = cov / std[:, None] / std[None, :]
    np.fill_diagonal(corr, 1.0)
    return scale * (np.diag(std) @ np.linalg.cholesky(corr))


def blocked_transform(cov, blocks, scale):
    """T (sorted order) of BlockedProposer.set_covariance (proposal.py:250-260): the
    covariance is reordered by i_of_j before std / corr / Cholesky."""
    i_of_j = [i for b in blocks for i in b]
    cov = np.asarray(cov, dtype=np.float64)
    return proposal_transform(cov[np.ix_(i_of_j, i_of_j)], scale)


class Binned:
    """Owns the arrays behind an `orc_binned`: the plik-lite arithmetic
    (planck_pliklite.py:143-155) in the device's operation order.  `bins` [n][3] =
    (spectrum, first l, last l); `weights` [lmax + 1] in D_l space; `cov` or its inverse
    Cholesky factor `Linv`; the linear emulator (theta0, D0, J) and the position `calib` of the
    calibration parameter among the sampled ones (both optional for `chi2_of_cl`)."""

    def __init__(self, bins, weights, X, cov=None, Linv=None, theta0=None, D0=None, J=None,
                 calib=0):
        self.bins = np.ascontiguousarray(bins, dtype=np.int32).reshape(-1, 3)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        n = len(self.X)
        if Linv is None:
            Linv = np.linalg.inv(np.linalg.cholesky(np.asarray(cov, dtype=np.float64)))
        self.Linv = np.ascontiguousarray(np.tril(Linv), dtype=np.float64)
        assert self.Linv.shape == (n, n) and len(self.bins) == n
        self.lmax = len(self.weights) - 1
        self.theta0 = np.ascontiguousarray(np.zeros(1) if theta0 is None else theta0, np.float64)
        self.n_lin = 0 if theta0 is None else len(self.theta0)
        self.D0 = np.ascontiguousarray(np.zeros((3, self.lmax + 1)) if D0 is None else D0,
                                       dtype=np.float64)
        self.J = np.ascontiguousarray(np.zeros((3, self.lmax + 1, max(self.n_lin, 1)))
                                      if J is None else J, dtype=np.float64)
        assert self.D0.shape == (3, self.lmax + 1)
        assert self.J.shape == (3, self.lmax + 1, max(self.n_lin, 1))
        b = _Binned()
        b.n_bins, b.lmax, b.n_lin, b.calib = n, self.lmax, self.n_lin, int(calib)
        b.bins, b.weights, b.X, b.Linv = _ip(self.bins), _dp(self.weights), _dp(self.X), _dp(self.Linv)
        b.theta0, b.D0, b.J = _dp(self.theta0), _dp(self.D0), _dp(self.J)
        self.calib = int(calib)
        # the binned response of the linear emulator (formed once, in the specified order)
        self.Bc0 = np.zeros(n)
        self.BJ = np.zeros((n, max(self.n_lin, 1)))
        lib().orc_binned_collapse(C.byref(b), _dp(self.Bc0), _dp(self.BJ))
        b.Bc0, b.BJ = _dp(self.Bc0), _dp(self.BJ)
        self.c = b

    def chi2_of_cl(self, L0, cl, A):
        """get_chi_squared for explicit spectra cl[n][3][m] (element l - L0 = D_l)."""
        cl = np.ascontiguousarray(cl, dtype=np.float64)
        A = np.ascontiguousarray(np.atleast_1d(A), dtype=np.float64)
        n, three, stride = cl.shape
        assert three == 3 and len(A) == n and stride + L0 > self.lmax
        out = np.empty(n)
        lib().orc_binned_chi2_of_cl(C.byref(self.c), n, int(L0), stride, _dp(cl), _dp(A), _dp(out))
        return out

    def delta(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        assert x.shape[1] == self.n_lin + 1
        out = np.empty((len(x), len(self.X)))
        for k in range(len(x)):
            lib().orc_binned_delta(C.byref(self.c), _dp(x[k]), _dp(out[k]))
        return out

    def chi2_of_delta(self, delta):
        delta = np.ascontiguousarray(np.atleast_2d(delta), dtype=np.float64)
        return np.array([lib().orc_binned_chi2_of_delta(C.byref(self.c), _dp(r)) for r in delta])


class Problem:
    """Owns the arrays behind an `orc_problem`.  All derived constants may be passed in
    (e.g. the ones the HIP engine reports) so that oracle and engine see one problem."""

    def __init__(self, d, kinds, a, b, periodic=None, means=None, covs=None, weights=None,
                 normalized=True, T=None, group_size=64, seed=1, temperature=1.0,
                 max_tries=None, derived=None, blocks=None, oversampling=None,
                 drag_last_slow=-1, drag_steps=0, incremental=False, refresh_every=None,
                 paired_variates=None, binned=None, carry_modes=False, carry_periodic=False):
        self.d = d
        # one mode with periodic parameters on step_inc_kernel<.., periodic>: wrap only what leaves
        # [lo, hi), carry the log-likelihood (Engine.carries_periodic())
        self.carry_periodic = bool(carry_periodic)
        # mixtures: the log-density of every mode is carried (step_inc_mix_kernel, the register-plane
        # kernel without periodic parameters); the engine says which: Engine.carries_modes()
        self.carry_modes = bool(carry_modes)
        self.binned = binned    # a `Binned` target instead of the mixture (means must be None)
        assert binned is None or (means is None and binned.n_lin == d - 1 and not incremental)
        # incremental evaluation (one Gaussian mode, non-periodic, one block): the whitened
        # residual is carried and refreshed every `refresh_every` (default 40 d) steps
        self.incremental = bool(incremental)
        # plain incremental steps draw the variates of two steps from one Philox block (what
        # the kernels do); False: one block per step, the stream of the from-scratch mode
        self.paired_variates = bool(incremental if paired_variates is None else paired_variates)
        self._refresh_every = refresh_every   # default: 40 cycle lengths (40 d for one block)
        # blocked proposal: `blocks` = lists of sampler indices, slow -> fast; T must then be
        # the transform of the covariance in sorted order (blocked_transform below)
        self.blocking = None
        if blocks is not None:
            self.block_size = np.array([len(b) for b in blocks], dtype=np.int32)
            self.oversample = np.array(oversampling if oversampling is not None
                                       else [1] * len(blocks), dtype=np.int32)
            self.i_of_j = np.array([i for b in blocks for i in b], dtype=np.int32)
            assert sorted(self.i_of_j.tolist()) == list(range(d))
            bl = _Blocking()
            bl.n_blocks = len(blocks)
            bl.size, bl.oversample, bl.i_of_j = (_ip(self.block_size), _ip(self.oversample),
                                                 _ip(self.i_of_j))
            bl.drag_last_slow, bl.drag_steps = drag_last_slow, drag_steps
            self.blocking = bl
        kinds = np.asarray(kinds, dtype=np.int32)
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        self.kind = np.ascontiguousarray(kinds)
        self.lo = np.where(kinds == 0, a, -np.inf).astype(np.float64)
        self.hi = np.where(kinds == 0, b, np.inf).astype(np.float64)
        self.loc = np.where(kinds == 1, a, 0.0).astype(np.float64)
        self.scale = np.where(kinds == 1, b, 1.0).astype(np.float64)
        self.periodic = (np.zeros(d, np.int32) if periodic is None
                         else np.ascontiguousarray(periodic, dtype=np.int32))
        if means is None:
            K = 0
            self.mean = np.zeros(1)
            Linv = np.zeros(1)
            cnorm = np.zeros(1)
            w = np.zeros(1)
        else:
            self.mean = np.ascontiguousarray(np.atleast_2d(means), dtype=np.float64)
            covs = np.asarray(covs, dtype=np.float64)
            covs = covs if covs.ndim == 3 else covs[None]
            K = len(self.mean)
            Ls = [np.linalg.cholesky(c) for c in covs]
            Linv = np.array([np.linalg.inv(L) for L in Ls])
            cnorm = np.array([d * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(L)))
                              if normalized else 0.0 for L in Ls])
            w = (np.full(K, 1.0 / K) if weights is None
                 else np.asarray(weights, dtype=np.float64))
            if not np.isclose(w.sum(), 1):
                w = w / w.sum()
        self.K = K
        uni = kinds == 0
        self.uniform_logp = float(-np.sum(np.log(self.hi[uni] - self.lo[uni])))
        self.mls = -np.log(self.scale) - np.log(2 * np.pi) / 2
        self.Linv, self.cnorm, self.weight = Linv, cnorm, w
        if derived is not None:  # constants reported by the engine
            self.uniform_logp = float(derived["uniform_logp"])
            self.mls = np.array(derived["mls"], dtype=np.float64)
            if K:
                self.Linv = np.array(derived["Linv"], dtype=np.float64)
                self.cnorm = np.array(derived["cnorm"], dtype=np.float64)
                self.weight = np.array(derived["weight"], dtype=np.float64)
        self.Linv = np.ascontiguousarray(self.Linv, dtype=np.float64)
        self.cnorm = np.ascontiguousarray(self.cnorm, dtype=np.float64)
        self.weight = np.ascontiguousarray(self.weight, dtype=np.float64)
        self.mls = np.ascontiguousarray(self.mls, dtype=np.float64)
        self.T = (np.zeros((d, d)) if T is None
                  else np.ascontiguousarray(np.tril(T), dtype=np.float64))
        self.group_size, self.seed, self.temperature = group_size, seed, temperature
        self.max_tries = float(max_tries if max_tries is not None else 40 * d)
        self._refresh()

    def _refresh(self):
        p = _Problem()
        p.d, p.n_modes, p.group_size = self.d, self.K, self.group_size
        p.has_periodic = int(self.periodic.any())
        p.seed, p.temperature, p.max_tries = self.seed, self.temperature, self.max_tries
        p.kind, p.lo, p.hi, p.loc = _ip(self.kind), _dp(self.lo), _dp(self.hi), _dp(self.loc)
        p.scale, p.mls, p.periodic = _dp(self.scale), _dp(self.mls), _ip(self.periodic)
        p.uniform_logp = self.uniform_logp
        p.mean, p.Linv, p.cnorm = _dp(self.mean), _dp(self.Linv), _dp(self.cnorm)
        p.weight, p.T = _dp(self.weight), _dp(self.T)
        p.blocking = (C.cast(C.pointer(self.blocking), C.c_void_p) if self.blocking is not None
                      else None)
        p.binned = (C.cast(C.pointer(self.binned.c), C.c_void_p) if self.binned is not None
                    else None)
        self.c = p    # (orc_block_slots below reads the blocking through it)
        self.refresh_every = int(self._refresh_every or
                                 40 * (lib().orc_block_slots(
                                     C.byref(p), 1 if self.blocking.drag_last_slow >= 0 else 0, None)
                                       if self.blocking is not None else self.d))
        p.incremental, p.refresh_every = int(self.incremental), self.refresh_every
        p.paired_variates = int(self.paired_variates)
        p.carry_modes = int(self.carry_modes and self.incremental and self.K > 1
                            and not self.periodic.any())
        p.carry_periodic = int(self.carry_periodic and self.incremental and self.K == 1
                               and bool(self.periodic.any()))
        self.c = p

    def set_T(self, T):
        self.T = np.ascontiguousarray(np.tril(T), dtype=np.float64)
        self._refresh()

    def evaluate(self, x, derived=False):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n = len(x)
        lp, ll = np.empty(n), np.empty(n)
        der = np.empty((n, max(self.K, 1) * self.d)) if derived else None
        lib().orc_evaluate(C.byref(self.c), n, _dp(x), _dp(lp), _dp(ll),
                           _dp(der) if derived else None)
        return (lp, ll, der) if derived else (lp, ll)

    def whiten(self, x):
        """y[n][K*d] = L_k^-1 (x - mu_k) per point and mode, in the oracle's chain order."""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        y = np.empty((len(x), max(self.K, 1) * self.d))
        for i in range(len(x)):
            lib().orc_whiten(C.byref(self.c), _dp(x[i]), _dp(y[i]))
        return y

    def whiten_directions(self, V):
        """U[K][ncol][d] (squeezed for one mode)."""
        V = np.ascontiguousarray(V, dtype=np.float64)
        U = np.empty((max(self.K, 1),) + V.shape)
        lib().orc_whiten_directions(C.byref(self.c), len(V), _dp(V), _dp(U))
        return U[0] if self.K <= 1 else U

    def direction_norms(self, U):
        """|u_c|^2 of whitened directions U[ncol][d] (one mode) in the four-chain pattern of every
        chi2 (orc_direction_norms): what step_inc_kernel's carried log-likelihood moves by."""
        U = np.ascontiguousarray(U, dtype=np.float64)
        UU = np.empty(len(U))
        lib().orc_direction_norms(C.byref(self.c), len(U), _dp(U), _dp(UU))
        return UU

    def basis(self, group, cycle):
        V = np.empty((self.d, self.d))
        lib().orc_basis(C.byref(self.c), group, cycle, _dp(V))
        return V  # V[c] = direction of column c

    def cycle_length(self, which=0):
        return lib().orc_block_slots(C.byref(self.c), which, None)

    def schedule(self, group, cycle, which=0):
        L = self.cycle_length(which)
        blk, bas, col = (np.zeros(max(L, 1), np.int32) for _ in range(3))
        lib().orc_block_schedule(C.byref(self.c), group, cycle, which, _ip(blk), _ip(bas),
                                 _ip(col))
        return blk[:L], bas[:L], col[:L]

    def basis_blocked(self, group, cycle, which=0):
        L = self.cycle_length(which)
        V = np.zeros((max(L, 1), self.d))
        flag = np.zeros(max(L, 1), np.int32)
        lib().orc_basis_blocked(C.byref(self.c), group, cycle, which, _dp(V), _ip(flag))
        return V[:L], flag[:L]


class State:
    """Walker state arrays of the oracle (walker-major x)."""

    def __init__(self, problem, x0, burn_in=0, row_cap=0, thin=1):
        self.p = problem
        x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        self.W, d = x0.shape
        self.x = x0.copy()
        lp, ll = problem.evaluate(x0)
        self.logprior, self.loglike = lp, ll
        self.logpost = lp + ll
        self.weight = np.ones(self.W, np.int32)
        self.prior_rej = np.zeros(self.W, np.int32)
        self.burn_left = np.full(self.W, burn_in + 1, np.int32)
        self.n_accept = np.zeros(self.W, np.int64)
        self.stuck = np.zeros(1, np.int32)
        self.row_cap = row_cap
        self.rows = np.zeros((self.W, max(row_cap, 1), d + 4)) if row_cap else None
        self.n_rows = np.zeros(self.W, np.int32)
        self.step = 0
        self.y = problem.whiten(self.x) if problem.incremental else np.zeros((1, 1))
        # carried mode log-densities (anchored at the first step, like the carried log-likelihood)
        self.amode = np.zeros((self.W, max(problem.K, 1)))
        s = _State()
        s.x, s.logprior, s.loglike = _dp(self.x), _dp(self.logprior), _dp(self.loglike)
        s.logpost, s.weight, s.prior_rej = _dp(self.logpost), _ip(self.weight), _ip(self.prior_rej)
        s.burn_left = _ip(self.burn_left)
        s.n_accept = self.n_accept.ctypes.data_as(c_int64_p)
        s.stuck = _ip(self.stuck)
        s.rows = _dp(self.rows) if row_cap else None
        s.n_rows = _ip(self.n_rows)
        s.row_cap = row_cap
        s.y = _dp(self.y)
        s.amode = _dp(self.amode)
        # thinned emission (OneSamplePoint.add_to_collection with output_thin, collection.py:1373-1383)
        self.thin = int(thin)
        self.thin_acc = np.zeros(self.W, np.int32)
        s.thin = self.thin
        s.thin_acc = _ip(self.thin_acc)
        self.c = s

    def run(self, n_steps, walker0=0, n_threads=1):
        acc = lib().orc_run(C.byref(self.p.c), C.byref(self.c), self.W, walker0, self.step,
                            n_steps, n_threads)
        self.step += n_steps
        return acc

    def anchor_modes(self):
        """carry_modes: a_k, loglike and logpost of every walker from its carried y (what a launch
        that finds no carried values does: orc_anchor_modes)."""
        for w in range(self.W):
            lib().orc_anchor_modes(C.byref(self.p.c), C.byref(self.c), w)

    def step_injected(self, vec, exp_draw):
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        return lib().orc_step_injected(C.byref(self.p.c), C.byref(self.c), _dp(vec),
                                       float(exp_draw))

    def step_injected_delta(self, delta, exp_draw):
        delta = np.ascontiguousarray(delta, dtype=np.float64)
        return lib().orc_step_injected_delta(C.byref(self.p.c), C.byref(self.c), _dp(delta),
                                             float(exp_draw))

    def drag_injected(self, slow, fast, e):
        slow = np.ascontiguousarray(slow, dtype=np.float64)
        fast = np.ascontiguousarray(fast, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        return lib().orc_drag_injected(C.byref(self.p.c), C.byref(self.c), _dp(slow),
                                       _dp(fast), _dp(e))

    def drain(self):
        """rows as (walker, weight, logpost, logprior, loglike, x...) like the engine."""
        out = []
        for w in range(self.W):
            n = min(int(self.n_rows[w]), self.row_cap)
            for r in range(n):
                out.append(np.concatenate(([w], self.rows[w, r])))
        self.n_rows[:] = 0
        d = self.x.shape[1]
        return np.array(out) if out else np.zeros((0, d + 5))


def moments(x, group_size, shift=None, group_sum=None, pooled=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    W, d = x.shape
    G = W // group_size
    shift = np.zeros(d) if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
    group_sum = np.zeros((G, d)) if group_sum is None else group_sum
    pooled = np.zeros((d, d)) if pooled is None else pooled
    lib().orc_moments(d, W, group_size, _dp(x), _dp(shift), _dp(group_sum), _dp(pooled))
    return group_sum, pooled


def max_threads():
    return lib().orc_max_threads()
