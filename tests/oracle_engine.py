"""TEST DOUBLE for `cobaya_amd.engine.Engine`: the same Python interface served by the CPU
oracle (oracle/cbind.py) instead of libmcmc_hip.so.

It exists so that the HOST side of the product -- `EnsembleMCMC` under the standalone class
and under a real Cobaya (`cobaya.run.run` with `sampler: mcmc_hip`) -- can be exercised end to
end in the CPU-only build container: initialize -> run -> checkpoints -> products -> resume.
It is test infrastructure: nothing under cobaya_amd/ or mcmc_hip/ imports it, and the product
has no CPU fallback (the real `Engine` raises without a gfx950 device).  Because the oracle is
the bit-exact specification of the kernels, a run on this double is the run the GPU would do.
"""
from __future__ import annotations

import numpy as np

from cobaya_amd.engine import (ERR_ARG, ERR_NOT_PD, ERR_STUCK, ChainStuck, EngineError,
                               NotPositiveDefinite)
from oracle import cbind as O


class OracleEngine:
    n_threads = 4

    def __init__(self, d, n_walkers, group_size=64, device=0, seed=0, walker_offset=0,
                 burn_in=0, temperature=1.0, proposal_scale=2.4, max_tries=None,
                 emit_capacity=0, shared_basis=True, incremental=False, basis_group_size=None):
        if n_walkers % group_size:
            raise EngineError(ERR_ARG, "n_walkers must be a multiple of group_size")
        # shared_basis False: every walker is its own basis "group" (the R-1 groups stay)
        self.own_basis = (not shared_basis) and d > 1
        if self.own_basis and incremental:
            raise EngineError(ERR_ARG, "incremental evaluation needs the shared basis")
        self.d, self.W, self.group_size = int(d), int(n_walkers), int(group_size)
        self.G = self.W // self.group_size
        self.K = None
        self.walker_offset = int(walker_offset)
        self.seed, self.burn_in = int(seed), int(burn_in)
        self.temperature, self.scale = float(temperature), float(proposal_scale)
        self.max_tries = float(max_tries if max_tries is not None else 40 * d)
        self.cap = int(emit_capacity)
        self.incremental = bool(incremental)
        self.basis_group_size = int(basis_group_size or group_size)
        if self.basis_group_size != group_size and (
                not incremental or n_walkers % self.basis_group_size
                or walker_offset % self.basis_group_size):
            raise EngineError(ERR_ARG, "a basis group wider than group_size needs incremental "
                                       "evaluation and must divide n_walkers and walker_offset")
        if self.incremental and (d < 2 or group_size % 64):
            raise EngineError(ERR_ARG, "incremental evaluation needs d >= 2 and group_size % 64 == 0")
        self._prior = self._target = self._blocking = None
        self._cov = None
        self._problem = self._state = None
        self._shift = np.zeros(d)
        self._n_snap, self._gsum, self._S = 0, np.zeros((self.G, d)), np.zeros((d, d))
        self._steps = 0
        self.closed = False

    # -- problem ------------------------------------------------------------------------
    def set_prior(self, kinds, a, b, periodic=None):
        self._prior = (np.array(kinds, np.int32), np.array(a, float), np.array(b, float),
                       None if periodic is None else np.array(periodic, np.int32))
        self._problem = None

    def set_target_gaussian_mixture(self, means, covs, weights=None):
        means = np.atleast_2d(np.array(means, float))
        self.K = len(means)
        self._target = dict(means=means, covs=np.array(covs, float).reshape(self.K, self.d, self.d),
                            weights=None if weights is None else np.array(weights, float),
                            normalized=True)
        self._problem = None

    def set_target_gaussian(self, mean, cov, normalized=True):
        self.K = 1
        self._target = dict(means=np.array(mean, float)[None], covs=np.array(cov, float)[None],
                            weights=None, normalized=bool(normalized))
        self._problem = None

    def set_target_one(self):
        self.K = 0
        self._target = dict(means=None, covs=None, weights=None, normalized=True)
        self._problem = None

    def set_target_binned_gaussian(self, target, emulator, calib_index):
        if self.incremental or self.own_basis or self.cap:
            raise EngineError(ERR_ARG, "the binned Gaussian target is evaluated from scratch with "
                                       "the shared basis and emit_capacity 0")
        if emulator.n != self.d - 1 or not 0 <= calib_index < self.d:
            raise EngineError(ERR_ARG, "the binned Gaussian target takes d - 1 emulator "
                                       "parameters and one calibration parameter")
        try:
            B = O.Binned(target.bin_table(), target.weights, target.X_data, cov=target.cov,
                         theta0=emulator.theta0, D0=emulator.D0, J=emulator.J, calib=calib_index)
        except np.linalg.LinAlgError:
            raise NotPositiveDefinite(ERR_NOT_PD, "the covariance of the binned data is not a "
                                                  "symmetric positive-definite matrix")
        self.K = 0
        self.n_bins = len(B.X)
        self._target = dict(means=None, covs=None, weights=None, normalized=True, binned=B)
        self._problem = None

    def set_blocking(self, blocks, oversampling=None, drag_last_slow=-1, drag_steps=0):
        blocks = [list(map(int, b)) for b in blocks]
        if sorted(i for b in blocks for i in b) != list(range(self.d)):
            raise EngineError(ERR_ARG, "The blocks do not contain all the parameter indices.")
        self._blocking = dict(blocks=blocks,
                              oversampling=list(oversampling or [1] * len(blocks)),
                              drag_last_slow=int(drag_last_slow), drag_steps=int(drag_steps))
        self._problem = None

    def _prob(self):
        if self._problem is None:
            kinds, a, b, per = self._prior
            bl = self._blocking or {}
            self._problem = O.Problem(
                self.d, kinds, a, b, per, **self._target,
                group_size=1 if self.own_basis else self.basis_group_size,
                seed=self.seed, temperature=self.temperature, max_tries=self.max_tries,
                incremental=self.incremental, carry_modes=self.carries_modes(),
                carry_periodic=self.carries_periodic(), **bl)
            if self._cov is not None:
                self._problem.set_T(self._transform(self._cov))
            if self._state is not None:      # re-point the state at the new problem
                self._state.p = self._problem
        return self._problem

    def cycle_length(self):
        p = self._prob()
        if self._blocking and self._blocking["drag_last_slow"] >= 0:
            return p.cycle_length(1)
        return p.cycle_length(0) if self._blocking else self.d

    def _transform(self, cov):
        if self._blocking:
            return O.blocked_transform(cov, self._blocking["blocks"], self.scale)
        return O.proposal_transform(cov, self.scale)

    def set_proposal_cov(self, cov):
        cov = np.array(cov, float).reshape(self.d, self.d)
        # proposal.py:243-246: symmetric and positive definite, else LinAlgError
        if not np.allclose(cov.T, cov) or not np.all(np.linalg.eigvalsh(cov) > 0):
            raise NotPositiveDefinite(ERR_NOT_PD, "The given covmat is not a positive-definite, "
                                                  "symmetric square matrix.")
        self._cov = cov.copy()
        if self._problem is not None:
            self._problem.set_T(self._transform(cov))

    def get_proposal_cov(self):
        return self._cov.copy()

    # -- evaluation / state -------------------------------------------------------------
    def evaluate(self, x, derived=False):
        return self._prob().evaluate(x, derived=derived)

    def set_state(self, x):
        x = np.array(x, float).reshape(self.W, self.d)
        self._state = O.State(self._prob(), x, burn_in=self.burn_in, row_cap=self.cap,
                              thin=self.emit_thin)
        self._state.step = self._steps

    def get_state(self):
        s = self._state
        return {"x": s.x.copy(), "logpost": s.logpost.copy(), "logprior": s.logprior.copy(),
                "loglike": s.loglike.copy(), "weight": s.weight.copy()}

    def get_full_state(self):
        s = self._state
        out = self.get_state()
        out.update(prior_rej=s.prior_rej.copy(), burn_left=s.burn_left.copy(),
                   n_accept=s.n_accept.copy(), step=np.uint64(s.step))
        if self.incremental:
            out["y"] = s.y.copy()
            if self.carries_modes() and s.step > 0:
                out["amode"] = s.amode.copy()
        if self.emit_thin > 1:
            out["thin_carry"] = s.thin_acc.copy()
        return out

    def carries_periodic(self):
        """The engine's rule (mcmc_hip_incremental_carries_periodic): one mode with 1..16 periodic
        parameters on Metropolis steps without emitted rows runs on step_inc_kernel<.., periodic>."""
        drag = bool(self._blocking) and self._blocking["drag_last_slow"] >= 0
        n_per = 0 if self._prior is None or self._prior[3] is None else int(self._prior[3].sum())
        return bool(self.incremental and self.K == 1 and 1 <= n_per <= 16 and not drag and self.cap == 0)

    def carries_modes(self):
        """The engine's rule (mcmc_hip_incremental_carries_modes): step_inc_mix_kernel serves 2..4
        modes at d <= 64 (5 at d <= 32, 6 at d <= 28) without periodic parameters, dragging or emitted rows."""
        drag = bool(self._blocking) and self._blocking["drag_last_slow"] >= 0
        periodic = self._prior is not None and self._prior[3] is not None and self._prior[3].any()
        mix = self.K is not None and ((2 <= self.K <= 4 and self.d <= 64) or (5 <= self.K <= 6 and self.d <= 32 and ((self.d + 3) // 4) * (self.K + 1) <= 50))
        return bool(self.incremental and mix and not drag and not periodic and self.cap == 0)

    def set_full_state(self, st):
        self.set_state(st["x"])
        s = self._state
        for k in ("logpost", "logprior", "loglike", "weight", "prior_rej", "burn_left",
                  "n_accept"):
            getattr(s, k)[...] = st[k]
        s.step = self._steps = int(st["step"])
        if self.incremental and "y" in st:
            s.y[...] = st["y"]
            if self.carries_modes():
                if "amode" in st:
                    s.amode[...] = st["amode"]
                else:   # (the engine re-anchors them on y at the next launch)
                    s.anchor_modes()
        if self.emit_thin > 1 and "thin_carry" in st:
            s.thin_acc[...] = st["thin_carry"]

    # -- sampling -----------------------------------------------------------------------
    def step(self, n_steps):
        if self.own_basis and self._blocking:
            raise EngineError(ERR_ARG, "shared_basis: False serves a single parameter block "
                                       "without dragging")
        drag = bool(self._blocking) and self._blocking["drag_last_slow"] >= 0
        periodic = self._prior[3] is not None and self._prior[3].any()
        if self.incremental and (not 1 <= self.K <= 4 or (self.K > 1 and self.d > 64)
                                 or (self.K != 1 and drag)
                                 or (periodic and (self.K != 1 or drag))):
            raise EngineError(ERR_ARG, "incremental evaluation serves one Gaussian mode (or a "
                                       "mixture without dragging and with non-periodic priors; "
                                       "periodic parameters without dragging)")
        self._prob()
        self._state.run(int(n_steps), walker0=self.walker_offset, n_threads=self.n_threads)
        self._steps = self._state.step

    def _raise_if_stuck(self):
        # like the real engine: only mcmc_hip_sync and mcmc_hip_fetch_moments report a walker
        # that tripped max_tries (step is asynchronous there)
        if int(self._state.stuck[0]):
            raise ChainStuck(ERR_STUCK, "The chain has been stuck for %g attempts, stopping "
                                        "sampling." % self.max_tries)

    def sync(self):
        self._raise_if_stuck()

    def counters(self):
        s = self._state
        return {"steps": int(s.step), "accepted": int(s.n_accept.sum()),
                "stuck": int(s.stuck[0]), "dropped_rows": 0}

    def drain_samples(self):
        rows = self._state.drain()
        if len(rows):
            rows[:, 0] += self.walker_offset
        return rows

    drain_slots = 4
    emit_thin = 1

    def set_emit_thin(self, thin):
        """The engine's rule (mcmc_hip_set_emit_thin): thinned on the device by every incremental
        Metropolis kernel (round 6); the from-scratch and dragging kernels thin on the host."""
        thin = int(thin)
        if thin < 1 or (thin > 1 and not self.cap):
            raise EngineError(ERR_ARG, "emit_thin needs emitted rows and thin >= 1")
        drag = bool(self._blocking) and self._blocking["drag_last_slow"] >= 0
        if thin > 1 and (not self.incremental or not self.K or self.K < 1 or drag):
            raise EngineError(ERR_ARG, "emit_thin: thin on the host")
        self.emit_thin = thin
        if self._state is not None:
            self._state.thin = thin
            self._state.c.thin = thin

    def get_thin_carry(self):
        return self._state.thin_acc.copy()

    def set_thin_carry(self, carry):
        self._state.thin_acc[...] = carry

    def set_drain_slots(self, n):
        self.drain_slots = int(n)

    def drain_samples_view(self):
        """Like the engine's: a read-only view of a slot that is REUSED `drain_slots` drains
        later -- here the expired slot is poisoned with NaN, so a caller that keeps a view
        too long is caught."""
        ring = self.__dict__.setdefault("_slot_ring", [])
        rows = self.drain_samples()
        if len(ring) >= self.drain_slots:
            old = ring.pop(0)
            old.flags.writeable = True
            old[...] = np.nan
        ring.append(rows)
        out = rows.view()
        rows.flags.writeable = False
        out.flags.writeable = False
        return out

    # -- R-1 of the confidence bounds (the engine's mcmc_hip_bounds_*) ---------------------------
    BOUNDS_MAX_SLOTS = 64

    def bounds_configure(self, n_slots):
        self._bring = [None] * int(n_slots)

    def bounds_snapshot(self, slot):
        self._bring[int(slot)] = self._state.x.copy()

    def bounds_get_slot(self, slot):
        return self._bring[int(slot)].copy()

    def bounds_set_slot(self, slot, x):
        self._bring[int(slot)] = np.array(x, float)

    def bounds_statistics(self, slots, limfrac, want_bounds=False):
        """oracle/ref_numpy.py: `confidence` per chain (= group: its walkers in every listed
        snapshot, unit weights) and parameter, then the sums the all-reduce carries."""
        from oracle import ref_numpy as R
        gs = self.group_size
        chains = [np.vstack([self._bring[int(s)][g * gs:(g + 1) * gs] for s in slots])
                  for g in range(self.G)]
        b = R.bounds_of_chains(chains, [np.ones(len(c)) for c in chains], 2.0 * limfrac)
        stats = R.bounds_payload(b, self._shift)
        return (stats, b) if want_bounds else stats

    # -- moments ------------------------------------------------------------------------
    def set_moment_shift(self, shift):
        self._shift = np.array(shift, float)

    def accumulate_moments(self):
        O.moments(self._state.x, self.group_size, self._shift, self._gsum, self._S)
        self._n_snap += 1

    def read_moments(self, reset=False):
        out = (self._n_snap, self._gsum.copy(), self._S.copy())
        if reset:
            self._n_snap = 0
            self._gsum[...] = 0
            self._S[...] = 0
        return out

    def set_moments(self, n_snapshots, group_sum, pooled_S):
        self._n_snap = int(n_snapshots)
        self._gsum[...] = group_sum
        self._S[...] = pooled_S

    def request_moments(self):
        n, gs, S = self.read_moments(reset=True)
        c = self.counters()
        self._requested = (n, gs, S, {"steps": c["steps"], "accepted": c["accepted"]})
        self._requested_stuck = int(self._state.stuck[0])

    def fetch_moments(self):
        out, self._requested = self._requested, None
        if self._requested_stuck:
            self._raise_if_stuck()
        return out

    # -- the device side of `device_checkpoint: reduce` (mcmc_hip_checkpoint_set_ring / _begin /
    # _request_payload / _fetch_payload): the ring of intervals, the window sums and the payload an
    # all-reduce carries, with the arithmetic of the sampler's host path
    def checkpoint_set_ring(self, intervals=(), min_capacity=16, first_index=0):
        from cobaya_amd.sampler import WindowSums
        self._ck_ring = [(int(n), np.array(gs, float), np.array(S, float)) for n, gs, S in intervals]
        self._ck_base = int(first_index)     # the run's index of _ck_ring[0]
        self._ck_sums = WindowSums()
        self.ckpt_capacity = 16
        while self.ckpt_capacity < max(len(self._ck_ring) + 2, int(min_capacity)):
            self.ckpt_capacity *= 2
        self._ck_payload = self._ck_out = None
        if not hasattr(self, "_ck_acc_prev"):
            self._ck_acc_prev = 0

    def checkpoint_set_accepted(self, accepted):
        self._ck_acc_prev = int(accepted)

    def checkpoint_begin(self, n_window_intervals, n_window_snapshots, steps_since):
        if self._requested is None:
            raise RuntimeError("request_moments must precede checkpoint_begin")
        if self._ck_payload is not None or self._ck_out is not None:
            raise RuntimeError("a device checkpoint is already in flight")
        n, gs, S, c = self._requested
        self._ck_ring.append((n, gs.copy(), S.copy()))
        if not 1 <= n_window_intervals <= self.ckpt_capacity:
            raise RuntimeError("the window outgrew the ring")
        ivs = self._ck_ring[-int(n_window_intervals):]
        if sum(iv[0] for iv in ivs) != n_window_snapshots:
            raise RuntimeError("the window's snapshot count disagrees with the ring's intervals")
        hi = self._ck_base + len(self._ck_ring)
        ring, base = self._ck_ring, self._ck_base
        # (the host path's arithmetic: WindowSums over the run's interval indices)
        gsum, Ssum = self._ck_sums.total(hi - len(ivs), hi, lambda i: ring[i - base][1:])
        drop = max(0, len(self._ck_ring) - self.ckpt_capacity)
        self._ck_ring = self._ck_ring[drop:]
        self._ck_base += drop
        self._ck_sums.forget_below(self._ck_base)
        G, W = self.G, self.W
        N_c = float(n_window_snapshots * self.group_size)
        means = gsum / N_c
        mm = means.T @ means
        acc = int(c["accepted"])
        self._ck_payload = np.concatenate((
            [float(G), N_c * G, float(acc - self._ck_acc_prev), float(steps_since * W), float(acc)],
            (Ssum - N_c * mm).ravel(), means.sum(0), mm.ravel()))
        self._ck_acc_prev = acc
        return 0, len(self._ck_payload)

    def checkpoint_request_payload(self):
        if self._ck_payload is None:
            raise RuntimeError("checkpoint_begin must precede checkpoint_request_payload")
        self._ck_out, self._ck_payload = self._ck_payload, None

    def checkpoint_fetch_payload(self):
        if self._ck_out is None:
            raise RuntimeError("no payload read-out is pending")
        out, self._ck_out = self._ck_out, None
        return out

    def close(self):
        self.closed = True
