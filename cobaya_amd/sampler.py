"""`mcmc_hip`: the walker-ensemble Metropolis sampler behind Cobaya's sampler plugin surface.

Host-side mirror of `cobaya.samplers.mcmc.MCMC` (reference: cobaya/samplers/mcmc/mcmc.py)
for the path the HIP engine accelerates.  Same constructor signature as
`cobaya.sampler.Sampler.__init__` (sampler.py:257-264), same life cycle
(`initialize()` -> `run()` -> `products()` / `samples()`), same option names and defaults as
cobaya/samplers/mcmc/mcmc.yaml, same `progress` table and collection columns.  What differs,
by design:

  * a "chain" is a GROUP of `group_size` walkers that share one Haar proposal basis; all
    walkers advance in lockstep on the GPU (one wavefront lane per walker);
  * covariance learning and R-1 use streaming sufficient statistics accumulated on the
    device (snapshots of the ensemble) instead of stored rows, and ranks exchange ONE
    all-reduce per checkpoint (RCCL over xGMI) instead of gather/bcast of means and covs
    (mcmc.py:791-793, 914, 1005, 1021);
  * `max_samples` counts accepted steps over ALL walkers; the R-1 of confidence-interval
    bounds (mcmc.py:918-1002) is formed on the device from a ring of ensemble snapshots
    (`bounds_snapshots`), with GetDist's `confidence` restated as exact order statistics
    (GetDist is absent here; parity unpinned for that number).

There is no CPU fallback: constructing the sampler without a usable gfx950 device raises.
"""
from __future__ import annotations

import copy
import datetime
import logging
import math
import os
import re

import numpy as np
import pandas as pd

from . import dist
from .collection import SampleCollection
from .engine import (ChainStuck, Engine, EngineError, NotPositiveDefinite, gelman_rubin,
                     incremental_supported)
from .model import ProblemSpec, UnsupportedModel

log = logging.getLogger("mcmc_hip")


class LoggedError(Exception):
    """Stand-in for cobaya.log.LoggedError when Cobaya is not importable: logs, then raises
    (log.py:22-46)."""

    def __init__(self, logger, msg, *args):
        text = msg % args if args else msg
        logger.error(text)
        super().__init__(text)


# cobaya/samplers/mcmc/mcmc.yaml:1-80, verbatim defaults
MCMC_DEFAULTS = {
    "burn_in": 0, "max_tries": "40d", "covmat": None, "covmat_params": None,
    "proposal_scale": 2.4, "output_every": "60s", "learn_every": "40d", "temperature": 1,
    "learn_proposal": True, "learn_proposal_Rminus1_max": 2.0,
    "learn_proposal_Rminus1_max_early": 30.0, "learn_proposal_Rminus1_min": 0.0,
    "max_samples": math.inf, "Rminus1_stop": 0.01, "Rminus1_cl_stop": 0.2,
    "Rminus1_cl_level": 0.95, "Rminus1_single_split": 4, "measure_speeds": True,
    "oversample_power": 0.4, "oversample_thin": True, "drag": False, "blocking": None,
    "callback_function": None, "callback_every": None, "seed": None,
    "check_every": None, "oversample": None, "drag_limits": None,
}
# options of the ensemble engine (new; typed class attributes in the Cobaya subclass)
HIP_DEFAULTS = {
    "n_walkers": 65536,       # walkers PER PROCESS (= per GPU)
    "group_size": None,       # walkers sharing one Haar basis = one R-1 "chain";
                              # default: 256 from 16384 walkers per process up, else 64
    "device": None,           # HIP ordinal; default LOCAL_RANK
    "steps_per_launch": None,  # Metropolis steps fused per call of the engine (between two moment
                               # snapshots); default "40d"
    "moments_every": 1,       # launches between moment snapshots
    "emit": "snapshots",      # "snapshots": ensemble state every snapshot_every steps,
                              # "chains": every accepted row with its integer weight
    "snapshot_every": None,   # steps; default = one checkpoint interval
    "max_rows": 1 << 21,      # cap on stored rows per process
    "device_checkpoint": None,  # Where the learn / convergence checkpoint runs:
                              # False: on the host, from the pinned read-back of the moments while
                              # the next launch runs (the host all-reduces: fastest on one GPU).
                              # "reduce": window sums, the payload and the all-reduce ON THE DEVICE
                              # in stream order (RCCL in place, queued by the library on the
                              # engine's stream: no host bounce, no RCCL kernel beside a step
                              # kernel); only the reduced 15 KB come back and the host solves them
                              # (R-1, Cholesky, upload) while the next launch runs.
                              # True (= "solve"): the solve on the device as well (one workgroup,
                              # checkpoint_kernels.hip): the refreshed proposal is in force for
                              # the very next launch and no host is in the loop, at ~0.15 ms of
                              # single-workgroup linear algebra IN the stream per checkpoint.
                              # None (default): "reduce" for several processes joined by the
                              # library's RCCL communicator, else False
    "bounds_snapshots": 16,   # R-1 of the confidence bounds (mcmc.py:918-1002) on the device: ring of
                              # this many ensemble snapshots (d * n_walkers doubles each, a thinned
                              # record of the later half of the run) the per-chain bounds are
                              # selected from -- in every emit mode; at most 16384 / group_size.
                              # 0: no ring, convergence is judged on the means only
    "drain_copy": False,      # emit: chains -- True: every drained block is copied out of the
                              # engine's pinned slot at once (the store owns its rows from the
                              # start); False: blocks are read in place and copied only when their
                              # slot is about to be reused
    "drain_ring_bytes": 1 << 33,  # emit: chains, zero-copy drain -- pinned host memory the engine's
                              # ring of drain slots may take (the ring is sized to outlive the
                              # `max_rows` retention window, so that stored rows are read in place
                              # and never copied again; 4 .. 64 slots)
    "row_buffer_bytes": 1 << 32,  # emit: chains -- device buffer of accepted rows between two
                              # drains (bounds steps_per_launch: every step may accept)
    "basis_group_size": None,  # walkers sharing one Haar basis per cycle (group_size times a
                              # power of two; incremental evaluation only).  Default: 1024
                              # from 16384 walkers per process up, 4096 from 65536 walkers
                              # up, else group_size.  The R-1
                              # "chains" stay the groups of group_size walkers.
    "evaluation": "auto",     # "full": every trial is evaluated from scratch (O(d^2));
                              # "incremental": the whitened residual L^-1 (x - mu) is carried
                              # and moved along the whitened shared direction (O(d), same
                              # posterior; one Gaussian mode or a mixture of <= 4 at d <= 64,
                              # non-periodic priors, parameter blocks / oversampling /
                              # dragging, snapshots); "auto": incremental where it applies
    "emit_thin": None,        # emit: chains -- thin the emitted rows by this factor (the rule of
                              # OneSamplePoint.add_to_collection, collection.py:1373-1383: weights
                              # add up, a row of weight sum // thin is written when the sum reaches
                              # thin).  None: the reference's rule -- 1, or with oversampling and
                              # oversample_thin the oversampling ratio (mcmc.py:377-389).  Thinned
                              # ON THE DEVICE where the incremental single-mode kernel emits
                              # (PCIe carries thin times fewer rows), on the host otherwise
    "checkpoint_lag": None,   # launches between the request of a learn / convergence checkpoint
                              # and its processing on the host (the refreshed proposal takes
                              # effect with the launch queued after that).  Default: 2 -- the
                              # host's pass over a checkpoint (~ 1 ms of Python, R-1, Cholesky,
                              # upload) then has two launches of cover; with one (round 3's
                              # single-process default) a launch of 0.9 ms no longer covers it
                              # on a slow host, and with several processes the collective's
                              # kernel needs the room as well -- see `advance`
    "shared_basis": True,     # True: the walkers of a group share one Haar basis per cycle;
                              # False: every walker draws its own (proposal.py:59-69 to the
                              # letter: the reference-faithful control, much slower)
}


def _number_with_units(value, unit, scale):
    """tools.py:454-511 NumberWithUnits: '40d' -> 40*scale, plain numbers unchanged."""
    if isinstance(value, str):
        m = re.fullmatch(r"\s*([0-9.eE+-]+|\.?inf)\s*(%s)?\s*" % unit, value)
        if not m:
            raise ValueError(f"cannot parse {value!r} (expected a number, optionally "
                             f"followed by '{unit}')")
        num = float(m.group(1).lstrip(".") if "inf" in m.group(1) else m.group(1))
        return num * scale if m.group(2) else num
    return value


class WindowSums:
    """Sums of the statistics of CONSECUTIVE checkpoint intervals [lo, hi) -- absolute indices:
    interval i is the i-th one the run has closed -- as a fixed function of those intervals: the
    canonical decomposition of [lo, hi) into aligned dyadic blocks, added in ascending order, a
    block's sum being left half + right half.  Block sums of four intervals and more are cached,
    so that a checkpoint costs O(log n) array additions instead of the n of a running `sum()` over
    the window (the window is the later half of the run: it grows with it, and at a thousand
    checkpoints the host spent milliseconds per checkpoint adding 60 KB arrays).  Nothing of the
    cache is state: a resumed run rebuilds it and forms the same sums, bit for bit."""

    def __init__(self):
        self._cache = {}

    def _block(self, level, k, get):
        if level == 0:
            return get(k)
        key = (level, k)
        v = self._cache.get(key)
        if v is None:
            a, b = self._block(level - 1, 2 * k, get), self._block(level - 1, 2 * k + 1, get)
            v = tuple(x + y for x, y in zip(a, b))
            if level >= 2:
                self._cache[key] = v
        return v

    def total(self, lo, hi, get):
        """Sum over the intervals lo <= i < hi (hi > lo); `get(i)` -> tuple of arrays."""
        i, acc = lo, None
        while i < hi:
            level = 0
            while i % (2 << level) == 0 and i + (2 << level) <= hi:
                level += 1
            part = self._block(level, i >> level, get)
            acc = part if acc is None else tuple(x + y for x, y in zip(acc, part))
            i += 1 << level
        return acc

    def forget_below(self, lo):
        for key in [k for k in self._cache if ((k[1] + 1) << k[0]) <= lo]:
            del self._cache[key]


class EnsembleMCMC:
    """The engine-backed sampler logic, host-agnostic.  Two hosts give it a constructor:

      * `MCMCHip` below (standalone: no Cobaya needed), and
      * `mcmc_hip.MCMCHip(EnsembleMCMC, cobaya.samplers.mcmc.MCMC)` (Cobaya-hosted:
        `cobaya.sampler.Sampler.__init__`, sampler.py:257-322, sets the options, the model and
        the `Output` object and then calls `initialize()`).

    What a host must provide: `self.log`, every option of MCMC_DEFAULTS + HIP_DEFAULTS as an
    attribute, the read-only attributes `model` / `output`, `get_name()`, and -- optionally --
    `self.spec` (else it is read from the live model in `initialize`)."""

    file_base_name = "mcmc_hip"
    sampler_type = "mcmc"
    supports_periodic_params = True
    fallback_covmat_scale = 4.0  # sampler.py:474
    _LoggedError = LoggedError   # the hosted class raises cobaya.log.LoggedError instead
    _engine_factory = staticmethod(Engine)  # the seam to libmcmc_hip.so (tests swap it)
    MAX_DIM = 128    # capi.hip: kMaxDimBig (mixtures: at most 64 modes, model.py)

    # ------------------------------------------------------------------ host seams
    def _fail(self, msg, *args, cause=None):
        """Log, then raise the host's LoggedError (log.py:22-46)."""
        err = self._LoggedError(self.log, msg, *args)
        if cause is not None:
            raise err from cause
        raise err

    def _out_parts(self):
        """(folder, file prefix) of the output, or None: from a `cobaya.output.Output`
        (output.py:199-200 `split_prefix`; OutputDummy is falsy) or from a plain string."""
        out = self.output
        if not out:
            return None
        if hasattr(out, "folder") and hasattr(out, "prefix"):
            return out.folder, out.prefix
        text = str(out)
        folder, prefix = os.path.split(text)   # 'chains/' -> ('chains', '')
        return folder or ".", prefix

    def _out_file(self, ext):
        """`[folder]/[prefix]<ext>`, e.g. ext = '.checkpoint' (sampler.py:324-338)."""
        folder, prefix = self._out_parts()
        return os.path.join(folder, prefix + ext)

    def _chain_file(self, suffix="txt"):
        """`[folder]/[prefix].<rank+1>.<suffix>` (output.py:298-322 prepare_collection)."""
        folder, prefix = self._out_parts()
        return os.path.join(folder, (prefix + "." if prefix else "")
                            + f"{1 + self.rank}.{suffix}")

    def _is_resuming(self):
        out = self.output
        if hasattr(out, "is_resuming"):
            return bool(out.is_resuming())
        return bool(getattr(self, "_resume", False))

    def _export_collection(self, coll):
        """What `products()['sample']` hands out; the hosted class converts to Cobaya's own
        `SampleCollection`."""
        return coll

    # ------------------------------------------------------------------ MCMC.initialize
    def initialize(self):
        """mcmc.py:111-271."""
        if getattr(self, "spec", None) is None:
            try:  # the attributes of the live model SURVEY.md 8b lists
                self.spec = ProblemSpec.from_cobaya_model(self.model)
            except UnsupportedModel as e:
                self._fail("mcmc_hip cannot sample this model: %s", str(e), cause=e)
        self.engine = None
        self.converged = bool(getattr(self, "converged", False))
        if getattr(self, "Rminus1_last", None) is None:
            self.Rminus1_last = np.inf
        spec = self.spec
        d = spec.d
        if d > self.MAX_DIM:
            # (fails here, with the reason, instead of at mcmc_hip_create's "no kernels compiled":
            # the reference has no cap, proposal.py:96-201)
            self._fail("mcmc_hip samples at most %d parameters (this model has %d): a walker group's "
                       "Haar basis of d x d doubles is built in the 160 KiB of LDS of one compute "
                       "unit.  Fix or marginalise parameters, or use the reference sampler `mcmc` "
                       "for this model.", self.MAX_DIM, d)
        if self.temperature is None:
            self.temperature = 1
        if self.temperature < 1:
            self.log.warning("Sampling temperatures <1 can lead to innacurate inference.")
        self.temperature = float(self.temperature)
        self.set_proposer_blocking()
        # 'd' units: one cycle of the proposer, thinned (mcmc.py:400-410)
        unit = max(1, self.cycle_length // self.output_thin)
        self.max_tries = _number_with_units(self.max_tries, "d", unit)
        self.learn_every = int(_number_with_units(self.learn_every, "d", unit))
        self.burn_in = int(_number_with_units(self.burn_in, "d", unit))
        self.steps_per_launch = max(1, int(_number_with_units(
            "40d" if self.steps_per_launch is None else self.steps_per_launch, "d", unit)))
        if self.drag:  # a dragging step costs 1 + 2 * drag_interp_steps evaluations
            self.steps_per_launch = max(1, self.steps_per_launch // (1 + self.drag_interp_steps))
        if self.callback_every is None:
            self.callback_every = self.learn_every
        if self.emit not in ("snapshots", "chains"):
            self._fail("emit must be 'snapshots' or 'chains', got %r", self.emit)
        dist.init_from_env()
        self.rank, self.size = dist.rank(), dist.size()
        if self.checkpoint_lag is None:
            self.checkpoint_lag = 2
        if int(self.checkpoint_lag) != self.checkpoint_lag or int(self.checkpoint_lag) < 1:
            self._fail("checkpoint_lag must be an integer >= 1, got %r", self.checkpoint_lag)
        self._ckpt_lag = int(self.checkpoint_lag)
        # seed: one key for the whole job; walkers are keyed by their global id
        if self.seed is None:
            seed = np.array([float(int.from_bytes(os.urandom(4), "little"))])
            if self.rank != 0:
                seed[:] = 0
            self.seed = int(dist.all_reduce_sum(seed)[0])
        else:
            self.log.warning("This run has been SEEDED with seed %s", self.seed)
        ss = np.random.SeedSequence(self.seed).spawn(self.size)[self.rank]  # sampler.py:378-384
        self._rng = np.random.default_rng(ss)
        W = int(self.n_walkers)
        if self.group_size is None:
            # large ensembles: wide groups (fewer Haar bases to generate, still >> d chains)
            self.group_size = 64
            for gs in (256, 128):
                if W >= 16384 and W % gs == 0:
                    self.group_size = gs
                    break
        self._split_of = None
        if W == int(self.group_size) and self.size == 1:
            # ONE group = the reference's single chain (mcmc.py:796-813), which it splits in time
            # into Rminus1_single_split parts for the R-1 test.  Here the group is split into that
            # many sub-groups of WALKERS (each a multiple of the 64-lane wavefront), which are
            # the chains of the test from then on -- or into the largest smaller number (>= 2)
            # that divides the walkers so.  Only the R-1 bookkeeping changes: the walkers keep
            # sharing ONE Haar basis where the engine can do that (`basis_group_size`, below).
            want = int(self.Rminus1_single_split)
            split = next((n for n in range(want, 1, -1) if W % (64 * n) == 0), None)
            if split:
                self._split_of = W
                self.group_size = W // split
                self.log.info("A single group of %d walkers: split into %d groups of %d for the "
                              "R-1 test (Rminus1_single_split: %d).", W, split, self.group_size, want)
        if W % int(self.group_size) or (W // int(self.group_size)) * self.size < 2:
            # R-1 needs at least two chains (= walker groups) over all processes
            # (mcmc.py:856-889; the reference splits a single chain instead, 796-813)
            self._fail("n_walkers (%d) must be a multiple of group_size (%d) and give at least "
                       "two groups over all processes: a group is one chain of the R-1 test",
                       W, int(self.group_size))
        device = self.device if self.device is not None else dist.default_device()
        cap = 0
        if self.emit == "chains":
            # every accepted row is kept on the device between drains: bound the buffer
            row_bytes = 8 * (d + 4) * W
            self.steps_per_launch = int(max(1, min(self.steps_per_launch,
                                                   int(self.row_buffer_bytes) // row_bytes)))
            cap = self.steps_per_launch
        if self.evaluation not in ("auto", "full", "incremental"):
            self._fail("evaluation must be 'auto', 'full' or 'incremental', got %r",
                       self.evaluation)
        # (the engine's own answer: tuned kernels for one mode, up to four at d <= 64 (six at d <= 28), up to
        # 16 periodic parameters, dragging of one non-periodic mode; the general kernel for
        # any other mixture / periodic set whose residuals fit the LDS)
        can_inc = (d >= 2 and int(self.group_size) % 64 == 0 and W % int(self.group_size) == 0
                   and incremental_supported(d, spec.n_modes, int(np.sum(spec.periodic)),
                                             self.drag_interp_steps if self.drag else 0,
                                             W, int(self.group_size))
                   and (not self.drag or (1 + self.drag_interp_steps) * ((d + 3) // 4) <= 128)
                   # accepted rows (emit: chains): Metropolis steps
                   and (self.emit == "snapshots" or not self.drag)
                   and d >= 2 and int(self.group_size) % 64 == 0
                   and bool(self.shared_basis))
        if spec.like_kind == "planck_pliklite":
            # not Gaussian in the calibration parameter: every trial is evaluated from scratch,
            # on the matrix cores (pliklite_kernels.hip); one launch = a few steps of 3 kernels
            can_inc = False
            if (len(self.blocks) > 1 or self.oversampling_factors[0] != 1 or self.drag
                    or self.emit != "snapshots" or not self.shared_basis or np.any(spec.periodic)):
                self._fail("the planck_pliklite likelihood is sampled with one parameter block, "
                           "the shared basis, non-periodic priors and emit: snapshots")
        if not self.shared_basis and (len(self.blocks) > 1 or self.oversampling_factors[0] != 1):
            self._fail("shared_basis: False serves a single parameter block without "
                       "oversampling or dragging")
        if self.evaluation == "incremental" and not can_inc:
            self._fail("evaluation: incremental serves Gaussian mixtures whose whitened residuals "
                       "(n_modes * d doubles per walker) fit the LDS, with Metropolis steps; "
                       "dragging for one Gaussian mode with non-periodic priors; d >= 2 and a "
                       "group_size that is a multiple of 64; emit: chains with Metropolis steps; use "
                       "'full' (or 'auto')")
        self.incremental = can_inc and self.evaluation != "full"
        # (Longer calls were tried as the default for large ensembles in round 5 -- "160d": the
        # engine forms the directions of a whole call at once and its launches follow each other
        # directly, whole job / step kernel 1.077 -> 1.036 at config 2 -- and NOT kept: the moment
        # snapshot is taken once per call, R-1 is estimated from the snapshots, and with a quarter
        # of them the default stopping rule needed four times the steps
        # (test_config2_full_size_run_converges).  `steps_per_launch: 160d` remains a choice for
        # runs that are not waiting for R-1.)
        if self.basis_group_size is None:
            self.basis_group_size = int(self.group_size)
            n_split = (self._split_of or 0) // int(self.group_size)
            if self.incremental and n_split >= 2 and n_split & (n_split - 1) == 0:
                # a single group that was split for the R-1 test: one basis for all, as asked
                self.basis_group_size = int(self._split_of)
            elif self.incremental and W >= 16384 and W % 1024 == 0 and 1024 % int(self.group_size) == 0:
                self.basis_group_size = 1024
                # at the benchmark size the Haar bases and the whitened columns of a launch
                # are 6 % of the step kernel with a basis per 1024 walkers (a tenth above
                # d = 64): a basis per 4096 walkers there
                if W >= 65536 and W % 4096 == 0:
                    self.basis_group_size = 4096
                    # ... and per 16 384 above d = 64, where the bases (sequential reflections,
                    # 80 KB of LDS each) are the longest chain between two step kernels:
                    # d = 100 whole job 3.03 -> 3.11e10 evals/s (8 192: 3.08)
                    if d > 64 and W % 16384 == 0:
                        self.basis_group_size = 16384
        if int(self.basis_group_size) != int(self.group_size) and not self.incremental:
            self._fail("basis_group_size (%s) differs from group_size (%s): this needs "
                       "incremental evaluation", self.basis_group_size, self.group_size)
        try:
            self.engine = self._engine_factory(d, W, group_size=int(self.group_size), device=int(device),
                                 seed=self.seed, walker_offset=self.rank * W,
                                 burn_in=self.burn_in * self.output_thin,  # mcmc.py:265
                                 temperature=self.temperature,
                                 proposal_scale=float(self.proposal_scale),
                                 max_tries=float(self.max_tries), emit_capacity=cap,
                                 shared_basis=bool(self.shared_basis),
                                 incremental=self.incremental,
                                 basis_group_size=int(self.basis_group_size))
            spec.configure(self.engine)
            self._size_drain_ring(d, W)
            if len(self.blocks) > 1 or self.oversampling_factors[0] != 1:
                self.engine.set_blocking(
                    [[spec.sampled.index(p) for p in b] for b in self.blocks],
                    self.oversampling_factors,
                    self.i_last_slow_block if self.drag else -1,
                    self.drag_interp_steps if self.drag else 0)
                assert self.engine.cycle_length() == self.cycle_length
        except EngineError as e:
            self._fail("%s", str(e), cause=e)
        # thinned output: on the device where the engine's emitting kernel does it (PCIe then
        # carries output_thin times fewer rows), else on the host (`_thin_rows`)
        self._device_thin = False
        if self.emit == "chains" and self.row_thin > 1 and hasattr(self.engine, "set_emit_thin"):
            try:
                self.engine.set_emit_thin(self.row_thin)
                self._device_thin = True
                self.log.info("Emitted rows are thinned by %d on the device.", self.row_thin)
            except EngineError as e:
                self.log.info("Emitted rows are thinned by %d on the host (%s).", self.row_thin, e)
        # initial proposal covariance (sampler.py:485-685), tempered (mcmc.py:438-440)
        self._initial_covmat, where_nan = self.initial_proposal_covmat()
        if np.any(where_nan) and self.learn_proposal:
            self.log.info("Covariance matrix %s. We will start learning the covariance of the "
                     "proposal earlier: R-1 = %g (would be %g if all params loaded).",
                     "not present" if np.all(where_nan) else "not complete",
                     self.learn_proposal_Rminus1_max_early, self.learn_proposal_Rminus1_max)
            self.learn_proposal_Rminus1_max = self.learn_proposal_Rminus1_max_early
        try:
            self.engine.set_proposal_cov(self._initial_covmat * self.temperature)
        except NotPositiveDefinite as e:
            self._fail("%s", str(e), cause=e)
        self.output_every = _number_with_units(self.output_every, "s", 1)
        self._last_state_dump = 0.0
        if self._is_resuming() and self.output and os.path.exists(self._state_file()):
            self._init_bookkeeping()
            self._init_bounds_ring()
            self._load_checkpoint()
            self._init_device_checkpoint()
            return
        # initial points (model.py:707-754 get_valid_point, one per walker)
        self.log.info("Getting initial points... (%d walkers)", W)
        # One generator per walker GROUP, keyed by the group's GLOBAL index: the initial points
        # -- like the Philox streams of the steps -- do not depend on how the walkers are
        # sharded, so a rank's shard equals the same slice of a single-process ensemble.
        gsz = int(self.group_size)
        g0 = self.rank * (W // gsz)
        rngs = [np.random.default_rng(np.random.SeedSequence(self.seed, spawn_key=(g0 + g,)))
                for g in range(W // gsz)]
        x0 = np.vstack([spec.sample_reference(gsz, r) for r in rngs])
        for _ in range(int(min(self.max_tries, 1000))):
            lp, ll = self.engine.evaluate(x0)
            bad = ~np.isfinite(lp + ll)
            if not bad.any():
                break
            for g in np.unique(np.flatnonzero(bad) // gsz):
                idx = g * gsz + np.flatnonzero(bad[g * gsz:(g + 1) * gsz])
                x0[idx] = spec.sample_reference(len(idx), rngs[g])
        else:
            self._fail("Could not find random point giving finite posterior after "
                                   "%g tries", self.max_tries)
        self.engine.set_state(x0)
        shift = np.concatenate((x0.sum(0), [W]))
        dist.all_reduce_sum(shift)
        self._shift = shift[:d] / shift[d]
        self.engine.set_moment_shift(self._shift)
        self._init_bookkeeping()
        self._init_bounds_ring()
        self._init_device_checkpoint()

    def _init_bounds_ring(self):
        """`bounds_snapshots` ensemble snapshots on the device (mcmc_hip_bounds_configure)."""
        n = int(self.bounds_snapshots or 0)
        if n < 0:
            self._fail("bounds_snapshots must be >= 0, got %r", self.bounds_snapshots)
        if n and hasattr(self.engine, "bounds_configure"):
            n = min(n, getattr(self.engine, "BOUNDS_MAX_SLOTS", 64), 16384 // int(self.group_size))
            self.engine.bounds_configure(n)
            self._bslots = [-1] * n

    def _bounds_take(self):
        """Called with every moment snapshot (index i): the ring keeps every `stride`-th one of
        the later half of the run -- the window of mcmc.py:787-790, `use_first = n / 2`.  A slot
        is free once its snapshot has left that window; when none is, the record is thinned
        (stride doubled, every other kept snapshot dropped)."""
        i = self._bsnap_idx
        self._bsnap_idx += 1
        if not self._bslots or i % self._bstride:
            return
        start = (i + 1) / 2.0
        free = [k for k, j in enumerate(self._bslots) if j < start]
        if not free:
            self._bstride *= 2
            self._bslots = [j if j % self._bstride == 0 else -1 for j in self._bslots]
            if i % self._bstride:
                return
            free = [k for k, j in enumerate(self._bslots) if j < 0]
        k = min(free, key=lambda k_: self._bslots[k_])
        self._bslots[k] = i
        self.engine.bounds_snapshot(k)

    def _init_device_checkpoint(self):
        """`device_checkpoint`: which part of the checkpoint runs on the device -- nothing (False),
        window sums + payload + all-reduce ("reduce"), or the solve as well (True / "solve")."""
        can = hasattr(self.engine, "checkpoint_begin") and dist.device_collective()
        mode = self.device_checkpoint
        if mode not in (None, False, True, "reduce", "solve"):
            self._fail("device_checkpoint must be one of False, 'reduce', True (= 'solve') or None, "
                       "got %r", mode)
        if mode and not can:
            self._fail("device_checkpoint: %r needs the HIP engine and, with several processes, "
                       "the library's RCCL communicator (not the gloo stand-in)", mode)
        # the communicator goes to the engine: its checkpoint reduces in place, in stream order
        attached = can and dist.attach(self.engine)
        if mode is None:
            mode = "reduce" if (attached and self.size > 1) else False
        if mode in (True, "solve") and not hasattr(self.engine, "checkpoint_solve"):
            self._fail("device_checkpoint: %r: this engine has no device-side solve", mode)
        self.device_checkpoint = mode
        self._device_ckpt = bool(mode)
        self._ckpt_solve_on_device = mode in (True, "solve")
        if self._device_ckpt:
            self.engine.checkpoint_set_ring(self._intervals, first_index=self._iv0)
            self.engine.checkpoint_set_accepted(self._acc_last)

    def set_proposer_blocking(self):
        """mcmc.py:320-410: parameter blocks and oversampling factors (manual `blocking` or
        from the likelihoods' speeds), the dragging decision, output thinning and the cycle
        length that the 'd' units refer to."""
        spec = self.spec
        if self.blocking:
            try:  # model.py:1469-1508 check_blocking
                factors, blocks = zip(*list(self.blocking))
                blocks = [list(b) for b in blocks]
                factors = [int(f) for f in factors]
            except (TypeError, ValueError) as e:
                self._fail("Manual blocking not understood. Check "
                                       "documentation.", cause=e)
            flat = [p for b in blocks for p in b]
            dup = sorted({p for p in flat if flat.count(p) > 1})
            if dup:
                self._fail("Manual blocking: repeated parameters: %r", dup)
            missing = [p for p in spec.sampled if p not in flat]
            if missing:
                self._fail("Manual blocking: missing parameters: %r", missing)
            unknown = [p for p in flat if p not in spec.sampled]
            if unknown:
                self._fail("Manual blocking: unknown parameters: %r", unknown)
            if list(factors) != sorted(factors):
                self.log.warning("Manual blocking: speed-blocking *apparently* non-optimal: "
                            "oversampling factors must go from small (slow) to large (fast).")
        else:
            try:
                blocks, factors = spec.param_blocking(
                    oversample_power=float(self.oversample_power or 0),
                    split_fast_slow=bool(self.drag))
            except UnsupportedModel as e:
                self._fail("%s", str(e), cause=e)
        self.blocks, self.oversampling_factors = blocks, list(factors)
        self.drag = bool(self.drag)
        if self.drag:  # mcmc.py:333-360
            if len(blocks) == 1:
                self.drag = False
                self.log.warning("Dragging disabled: not possible if there is only one block.")
            elif max(factors) / min(factors) < 2:
                self.drag = False
                self.log.warning("Dragging disabled: speed ratios < 2.")
        self.drag_interp_steps = 0
        if self.drag:
            n_slow = sum(len(b) for b in blocks[:1 + self.i_last_slow_block])
            n_fast = spec.d - n_slow
            self.drag_interp_steps = int(np.round(
                factors[self.i_last_slow_block + 1] * n_fast / n_slow))
            if self.drag_interp_steps < 2:
                self.drag = False
                self.log.warning("Dragging disabled: speed ratio and fast-to-slow ratio not large "
                            "enough.")
        self.output_thin = 1
        if self.drag:
            self.log.info("Dragging with number of interpolating steps: %d", self.drag_interp_steps)
            self.cycle_length = sum(len(b) for b in blocks[:1 + self.i_last_slow_block])
            # (emit: chains: rows leave the from-scratch dragging kernel, d <= 32 -- `initialize`)
        else:
            if any(f > 1 for f in factors):
                self.log.info("Oversampling with factors: %r", list(zip(factors, blocks)))
                if self.oversample_thin:  # mcmc.py:377-389
                    self.output_thin = int(np.round(
                        sum(len(b) * o for b, o in zip(blocks, factors)) / spec.d))
            self.cycle_length = sum(len(b) * o for b, o in zip(blocks, factors))
        # the factor emitted rows are thinned by: the reference's output_thin, or mcmc_hip's own
        # `emit_thin` (which leaves the 'd' units and the burn-in of output_thin alone)
        self.row_thin = self.output_thin
        if self.emit_thin:
            if int(self.emit_thin) < 1:
                self._fail("emit_thin must be a positive integer, got %r", self.emit_thin)
            self.row_thin = int(self.emit_thin)

    @property
    def i_last_slow_block(self):
        """mcmc.py:273-279: index of the last slow block of the fast/slow split."""
        if self.drag:
            return next(i for i, o in enumerate(self.oversampling_factors) if o != 1) - 1
        return 0

    def _init_bookkeeping(self):
        spec = self.spec
        self.collection = self._export_collection(
            SampleCollection(spec.sampled, spec.derived, self._like_names(),
                             self.temperature, name=str(1 + self.rank)))
        # chains mode with output thinned on the host: added weight per walker (dense, saved in
        # the state file under the key the device-thinned path uses: `thin_carry`)
        self._thin_carry = np.zeros(int(self.n_walkers), dtype=np.int64)
        self._snap_stride, self._snap_count, self._rows_capped = 1, 0, False
        self._rows = []          # (walker, weight, logpost, logprior, loglike, x...) blocks
        self._n_rows = 0
        self._pending = []       # stored blocks not yet appended to the chain file
        self._txt_rows = 0       # data rows of this process' chain file
        self._carried = None     # table rows of earlier legs, read back at resume
        self._intervals = []     # per checkpoint: (n_snapshots, group_sum[G,d], pooled_S[d,d])
        self._iv0 = 0            # absolute index of _intervals[0] (intervals dropped so far)
        self._wsums = WindowSums()
        self._dropped_snapshots = 0
        self._progress_rows = {}  # i_learn -> row dict (DataFrame built on demand: `progress`)
        self.i_learn = 1
        self.n_steps_raw = 0     # Metropolis steps per walker (mcmc.py:472)
        self._accepted_total = 0
        self._acc_last = 0
        self._steps_last = 0
        self._acc_rate = 0.25
        self._launches = 0
        self._since_snapshot = 0
        self._next_ckpt = None   # steps per walker at which the next learn checkpoint is due
        self._ckpt_pending = False  # moments requested, checkpoint not processed yet
        self._ckpt_age = 0       # launches queued since the request
        self._ckpt_post = False  # a checkpoint was processed in this pass: its snapshot / callback / files follow
        self._ckpt_on_device = False   # the pending checkpoint was solved on the device
        self._snaps_in_interval = 0    # moment snapshots since the last checkpoint request
        self._ckpt_steps_last = 0      # steps per walker at the last request
        # the device ring behind R-1 of the bounds: slot -> index of the moment snapshot it holds
        # (-1: free), the stride of the thinned record, moment snapshots taken so far
        self._bslots, self._bstride, self._bsnap_idx = [], 1, 0

    # ------------------------------------------------------------------ a17
    def initial_proposal_covmat(self):
        """sampler.py:485-685: `covmat` (matrix or file) > `proposal`^2 > ref variance / 4 >
        prior variance / 4; off-diagonals only from the given matrix."""
        spec = self.spec
        names = spec.sampled
        d = spec.d
        cov = np.diag([np.nan] * d)
        covmat, covmat_params = self.covmat, self.covmat_params
        if isinstance(covmat, str):
            if covmat.lower() == "auto":
                self._fail("covmat: auto (cosmology database) is not available in "
                                       "mcmc_hip")
            try:
                with open(covmat, encoding="utf-8-sig") as f:
                    header = f.readline()
                loaded = np.atleast_2d(np.loadtxt(covmat))
            except OSError as e:
                self._fail("Can't open covmat file '%s'.", covmat, cause=e)
            if header[0] != "#":
                self._fail("The first line of the covmat file '%s' must be one "
                                       "list of parameter names separated by spaces and "
                                       "staring with '#'", covmat)
            covmat_params = header.strip("#").strip().split()
            covmat = loaded
        if covmat is not None:
            if not covmat_params:
                self._fail("If a covariance matrix is passed as a numpy array, you "
                                       "also need to pass the parameters it corresponds to via "
                                       "'covmat_params: [name1, name2, ...]'.")
            covmat = np.atleast_2d(np.array(covmat, dtype=float))
            covmat_params = list(covmat_params)
            if len(covmat_params) != len(set(covmat_params)):
                self._fail("Parameter(s) appear more than once in `covmat_params`")
            if covmat.shape != (len(covmat_params),) * 2:
                self._fail("The number of parameters in `covmat_params` and the "
                                       "dimensions of the matrix do not agree: %d vs %r",
                                  len(covmat_params), covmat.shape)
            if not np.allclose(covmat.T, covmat):
                self._fail("The covariance matrix passed is not a symmetric square "
                                       "matrix.")
            idx_l = [j for j, p in enumerate(covmat_params) if p in names]
            idx_s = [names.index(covmat_params[j]) for j in idx_l]
            if not idx_s:
                self._fail("A proposal covariance matrix has been loaded, but none "
                                       "of its parameters are actually sampled here.")
            cov[np.ix_(idx_s, idx_s)] = covmat[np.ix_(idx_l, idx_l)]
        # variances the matrix did not provide, in order of preference: `proposal` width
        # squared, then the ref pdf's variance (prior's where there is none) over
        # `fallback_covmat_scale`; the first mask is what "incomplete covmat" means
        from_props = np.array([np.nan if not p else float(p) ** 2 for p in spec.proposal])
        from_ref = spec.reference_variances() / self.fallback_covmat_scale
        missing = np.isnan(cov.diagonal())
        where_nan = missing.copy()
        for source in (from_props, from_ref):
            idx = np.flatnonzero(missing)
            cov[idx, idx] = source[idx]
            missing = np.isnan(cov.diagonal())
        return cov, where_nan

    PROGRESS_COLUMNS = ["N", "timestamp", "acceptance_rate", "Rminus1", "Rminus1_cl"]

    @property
    def progress(self):
        """mcmc.py:165-181: table (N, timestamp, acceptance_rate, Rminus1, Rminus1_cl), one row
        per learn/convergence checkpoint, indexed from 1."""
        df = pd.DataFrame.from_dict(self._progress_rows, orient="index",
                                    columns=self.PROGRESS_COLUMNS)
        return df.astype({c: float for c in self.PROGRESS_COLUMNS if c != "timestamp"})

    # ------------------------------------------------------------------ MCMC.run
    def n(self):
        """Accepted steps so far, over all walkers of all processes."""
        return self._accepted_total

    def _checkpoint_steps(self):
        """Steps per walker between checkpoints: `learn_every` ACCEPTED rows per chain
        (mcmc.py:757-760) at the measured acceptance rate, rounded up to whole launches."""
        steps = self.learn_every / max(self._acc_rate, 0.02)
        return max(1, math.ceil(steps / self.steps_per_launch)) * self.steps_per_launch

    def run(self):
        """mcmc.py:451-528."""
        if self.converged:
            # a resumed run whose checkpoint says "converged" under unchanged stop criteria
            # (sampler.py:339-351, mcmc.py:1080-1088): nothing to do, nothing is rewritten
            self.log.info("The run was already converged: nothing to do (change Rminus1_stop, "
                          "Rminus1_cl_stop, Rminus1_cl_level or max_samples to continue).")
            return
        if self._accepted_total >= self.max_samples:
            self.log.info("The maximum number of accepted steps (%s) was already reached: "
                          "nothing to do.", self.max_samples)
            return
        self.log.info("Sampling!%s", (" (NB: no accepted step will be saved until %d burn-in "
                                      "samples have been obtained)" % self.burn_in)
                      if self.burn_in else "")
        if self._next_ckpt is None:
            self._next_ckpt = self._checkpoint_steps()
        try:
            self._request_checkpoint_if_due()   # (a resumed run: the request the dump preceded)
            while self._accepted_total < self.max_samples and not self.converged:
                self.advance()
            if self._ckpt_pending:
                self._finish_checkpoint()
            if self._ckpt_post:
                self._after_checkpoint()
            self.engine.sync()
            self._update_counters()
        except ChainStuck as e:
            self._fail("%s Make sure the reference point is sensible and initial "
                       "covmat. (see `max_tries`)", str(e), cause=e)
        if self._accepted_total >= self.max_samples:
            self.log.info("Reached maximum number of accepted steps allowed (%s). Stopping.",
                          self.max_samples)
        self.log.info("Sampling complete after %d accepted steps.", self._accepted_total)
        if self.output:
            self.write_checkpoint(force_state=True)

    def advance(self):
        """One pass of the hot loop (the body of mcmc.py:451-528 for every walker): a fused
        launch of `steps_per_launch` Metropolis steps, the moment snapshot, sample emission,
        and the learn/convergence checkpoint -- which is taken OFF the critical path: when a
        checkpoint falls due, the read-out of the sufficient statistics is only QUEUED behind
        the launch (`request_moments`); the next launch is queued right after it, and while
        that one runs the host fetches the statistics, all-reduces them, forms R-1 and uploads
        the refreshed proposal in stream order.  The new proposal therefore takes effect
        `checkpoint_lag` launches after the checkpoint (deterministically, also across a
        resume).

        `checkpoint_lag: 2` (the default): the host's pass over a checkpoint -- about a
        millisecond of Python, R-1, a Cholesky factor, the upload -- has two launches of cover.
        With one, a launch that has become shorter than that pass (0.9 ms at BASELINE config 2
        since round 4) leaves the device waiting on a slow host; and with several processes
        on the host path the step kernel leaves no room on the device beside it (DESIGN.md 4:
        its workgroups are exactly what the chip holds), so the RCCL kernel of the all-reduce
        only runs when the first workgroups of the launch in flight retire: the host gets the
        reduced statistics at the END of that launch.  A launch is therefore always queued
        behind the one the checkpoint's processing overlaps."""
        eng, spl = self.engine, self.steps_per_launch
        eng.step(spl)
        self.n_steps_raw += spl
        self._launches += 1
        self._since_snapshot += spl
        # A pending checkpoint is processed FIRST, right behind the launch just queued: its
        # statistics came back long ago, and a refreshed proposal is then uploaded ahead of the
        # moment snapshot -- the directions of the next launch, formed on the second stream
        # behind that upload, start at once when this launch ends (tools/gpu.sh timeline, round 4:
        # 127 -> 85 us between two step kernels after a learn checkpoint)
        if self._ckpt_pending:
            self._ckpt_age += 1
            if self._ckpt_age >= self._ckpt_lag:
                self._finish_checkpoint()
        if self._launches % max(1, int(self.moments_every)) == 0:
            eng.accumulate_moments()
            self._snaps_in_interval += 1
            self._bounds_take()
        snap_every = int(self.snapshot_every) if self.snapshot_every else None
        if self.emit == "chains":
            if hasattr(eng, "drain_samples_view"):
                # rows land in a pinned host slot of the engine at PCIe speed and are read there
                self._expire_row_views()
                rows = eng.drain_samples_view()
                if self.drain_copy:
                    self._store_rows(np.array(rows))
                else:
                    self._store_rows(rows, view=True)
            else:
                self._store_rows(eng.drain_samples())
        elif snap_every and self._since_snapshot >= snap_every:
            self._snapshot()
        if self._ckpt_post:
            self._after_checkpoint()
        # (a run that has just converged or reached max_samples leaves the loop: a request
        # queued now would be processed after convergence -- an extra progress row, possibly
        # `converged` flipped back under a "Sampling complete" log, mcmc.py:470)
        if not self.converged and self._accepted_total < self.max_samples:
            self._request_checkpoint_if_due()

    def _request_checkpoint_if_due(self):
        if self._ckpt_pending or self._next_ckpt is None or self.n_steps_raw < self._next_ckpt:
            return
        if hasattr(self.engine, "request_moments"):
            self.engine.request_moments()
        self._ckpt_on_device = False
        if self._device_ckpt and self._snaps_in_interval > 0:
            self._begin_device_checkpoint()
        self._snaps_in_interval = 0
        self._ckpt_steps_last = self.n_steps_raw
        self._ckpt_pending = True
        self._ckpt_age = 0
        self._next_ckpt = self.n_steps_raw + self._checkpoint_steps()

    @staticmethod
    def _window_len(counts, dropped):
        """Intervals of the window (`_window`): the shortest suffix of `counts` (snapshots per
        checkpoint interval) holding at least half of all snapshots taken so far."""
        total = dropped + sum(counts)
        k = 0
        while k + 1 < len(counts) and sum(counts[k + 1:]) >= total / 2:
            k += 1
        return len(counts) - k

    def _begin_device_checkpoint(self):
        """Queue the checkpoint ON THE DEVICE behind the launch (mcmc_hip_checkpoint_*): window
        sums over the ring of intervals, the all-reduce in place (RCCL, on the engine's stream),
        R-1 and -- inside the learning window -- the refreshed transform, written where the next
        launch reads it.  The host only reads the outcome later (`_finish_checkpoint`)."""
        eng = self.engine
        counts = [iv[0] for iv in self._intervals] + [self._snaps_in_interval]
        k = self._window_len(counts, self._dropped_snapshots)
        if k + 1 > eng.ckpt_capacity:     # the window outgrew the ring: reload it, larger
            eng.checkpoint_set_ring(self._intervals, min_capacity=2 * (k + 1), first_index=self._iv0)
        ptr, n = eng.checkpoint_begin(k, sum(counts[-k:]), self.n_steps_raw - self._ckpt_steps_last)
        if self.size > 1 and not getattr(eng, "comm_attached", False):
            # (an engine the communicator is attached to has queued the all-reduce itself)
            dist.all_reduce_sum_device(ptr, n, eng.stream_handle())
        if self._ckpt_solve_on_device:
            learn = bool(self.learn_proposal)
            eng.checkpoint_solve(self.learn_proposal_Rminus1_min if learn else np.inf,
                                 self.learn_proposal_Rminus1_max if learn else -np.inf)
        else:   # only the reduced payload comes back; the host solves it beside the next launch
            eng.checkpoint_request_payload()
        self._ckpt_on_device = True

    def _finish_checkpoint(self):
        self._ckpt_pending = False
        moments = (self.engine.fetch_moments() if hasattr(self.engine, "fetch_moments")
                   else None)
        dev = payload = None
        if self._ckpt_on_device and self._ckpt_solve_on_device:
            dev = self.engine.checkpoint_fetch()
        elif self._ckpt_on_device:
            payload = self.engine.checkpoint_fetch_payload()
        self._ckpt_on_device = False
        self.check_convergence_and_learn_proposal(moments, dev, payload)
        self.i_learn += 1
        self._ckpt_post = True

    def _after_checkpoint(self):
        """What follows a processed checkpoint once the launch's own bookkeeping (moment
        snapshot, emission) is done: the checkpoint's sample snapshot, the callback, the files."""
        self._ckpt_post = False
        if self.emit == "snapshots" and not self.snapshot_every:
            self._snapshot()
        if self.callback_function:
            self._callback()(self)
        if self.output:
            self.write_checkpoint()

    def _callback(self):
        """mcmc.py:160-163: a callable, or a string resolved like Cobaya's external functions
        (a lambda / `import_module('m').f` expression)."""
        fn = self.callback_function
        if callable(fn):
            return fn
        import importlib
        return eval(fn, {"np": np, "import_module": importlib.import_module})  # noqa: S307

    # ------------------------------------------------------------------ checkpoint / resume
    def _state_file(self):
        return self._chain_file("state.npz")

    BOUNDS_RING_SAVE_BYTES = 1 << 26   # snapshots of the bounds ring kept in the state file

    # the options whose change re-opens a converged run (mcmc.py:1080-1088)
    CONVERGE_OPTIONS = ("Rminus1_stop", "Rminus1_cl_stop", "Rminus1_cl_level", "max_samples")

    def write_checkpoint(self, force_state=False):
        """mcmc.py:1045-1078: `prefix.checkpoint` (yaml), `prefix.covmat`, `prefix.progress` on
        the root; plus, per process and at most every `output_every` seconds (mcmc.py:473-481,
        697-699), the rows stored since the last dump appended to `prefix.<rank+1>.txt` and
        then the complete ensemble state `prefix.<rank+1>.state.npz` (walkers, counters,
        Philox step counter, moment window, number of rows in the text file) from which
        `resume` continues bit-identically -- the reference can only restart from its last
        stored row and does not save its RNG state (sampler.py:373)."""
        import time

        import yaml
        if not self.output:
            return
        os.makedirs(self._out_parts()[0], exist_ok=True)
        if self.rank == 0:
            np.savetxt(self._out_file(".covmat"),
                       self.engine.get_proposal_cov() / self.temperature,  # mcmc.py:1049-1051
                       header=" ".join(self.spec.sampled))
            ck = {"sampler": {self.get_name(): {
                "converged": bool(self.converged), "Rminus1_last": float(self.Rminus1_last),
                "burn_in": 0, "mpi_size": int(self.size)}}}
            with open(self._out_file(".checkpoint"), "w", encoding="utf-8") as f:
                yaml.safe_dump(ck, f)
            with open(self._out_file(".progress"), "w", encoding="utf-8") as f:
                f.write("# " + " ".join(f"{c:>15}" for c in self.progress.columns) + "\n")
                if len(self.progress):
                    f.write(self.progress.to_string(header=False, index=False) + "\n")
        now = time.time()
        if force_state or now - self._last_state_dump >= float(self.output_every):
            self._last_state_dump = now
            self._flush_rows()
            st = self.engine.get_full_state()
            if self.emit == "chains" and self.row_thin > 1 and not self._device_thin:
                st["thin_carry"] = self._thin_carry.astype(np.int32)   # (host-thinned rows)
            st["geometry"] = np.array([int(self.group_size), int(self.basis_group_size)], dtype=np.int64)
            acc_n, acc_gs, acc_S = self.engine.read_moments(reset=False)
            st.update(acc_n=np.int64(acc_n), acc_gs=acc_gs, acc_S=acc_S)
            ivs = self._intervals
            # the bounds ring: its books always, its snapshots while they are small (a resumed run
            # then forms the same Rminus1_cl; a large ring restarts empty)
            held = [k for k, j in enumerate(self._bslots) if j >= 0]
            ring_bytes = 8 * len(held) * int(self.n_walkers) * self.spec.d
            if held and ring_bytes <= self.BOUNDS_RING_SAVE_BYTES:
                st["bring"] = np.array([self.engine.bounds_get_slot(k) for k in held])
            elif held:
                self._save_bounds_sidecar()
            st["bslots"] = np.array(self._bslots, dtype=np.int64)
            st["bbook"] = np.array([self._bstride, self._bsnap_idx], dtype=np.int64)
            tmp = self._state_file() + ".tmp.npz"
            np.savez(tmp, **st,
                     proposal_cov=self.engine.get_proposal_cov(), shift=self._shift,
                     iv_n=np.array([iv[0] for iv in ivs], dtype=np.int64),
                     iv_gs=np.array([iv[1] for iv in ivs]), iv_S=np.array([iv[2] for iv in ivs]),
                     iv0=np.int64(self._iv0),
                     book=np.array([self.n_steps_raw, self.i_learn, self._acc_last,
                                    self._steps_last, self._launches, self._dropped_snapshots,
                                    self._accepted_total, self.seed, self.size,
                                    int(self.n_walkers), int(self._next_ckpt or 0),
                                    self._snap_stride, self._snap_count, self._txt_rows],
                                   dtype=np.int64),
                     fbook=np.array([self._acc_rate, self.Rminus1_last, float(self.converged),
                                     self.learn_proposal_Rminus1_max]
                                    + [float(getattr(self, k)) for k in self.CONVERGE_OPTIONS]),
                     progress=self.progress.to_numpy(dtype=object).astype(str))
            os.replace(tmp, self._state_file())   # never leave a half-written state behind

    def _bounds_sidecar(self):
        return self._chain_file("bounds.npy"), self._chain_file("bounds_tags.npy")

    def _save_bounds_sidecar(self):
        """A bounds ring too large for the state file (251 MB at BASELINE config 2: 16 slots of
        65 536 x 30 doubles) lives in a sidecar `prefix.<n>.bounds.npy` of fixed shape
        [slots][W][d], written IN PLACE and only where a slot's snapshot changed since the last
        dump; `prefix.<n>.bounds_tags.npy` (replaced atomically, after the data) says which
        snapshot each slot of the file holds.  A resumed run restores exactly the slots whose tag
        equals the state file's book -- after a crash between the two writes a slot is dropped,
        never mixed up -- so Rminus1_cl after a resume is the uninterrupted run's."""
        data_f, tags_f = self._bounds_sidecar()
        n, shape = len(self._bslots), (len(self._bslots), int(self.n_walkers), self.spec.d)
        tags = np.full(n, -1, dtype=np.int64)
        mm = None
        if os.path.exists(data_f) and os.path.exists(tags_f):
            try:
                mm = np.load(data_f, mmap_mode="r+")
                old = np.load(tags_f)
                if mm.shape == shape and old.shape == tags.shape:
                    tags = old
                else:
                    mm = None
            except Exception:
                mm = None
        if mm is None:
            mm = np.lib.format.open_memmap(data_f, mode="w+", dtype=np.float64, shape=shape)
            tags[:] = -1
        dirty = [k for k, j in enumerate(self._bslots) if j >= 0 and tags[k] != j]
        if dirty:   # (the tags of the slots being rewritten are invalid until the data is down)
            tags[dirty] = -1
            np.save(tags_f + ".tmp.npy", tags)
            os.replace(tags_f + ".tmp.npy", tags_f)
            for k in dirty:
                mm[k] = self.engine.bounds_get_slot(k)
            mm.flush()
        for k, j in enumerate(self._bslots):
            tags[k] = j if j >= 0 else -1
        del mm
        np.save(tags_f + ".tmp.npy", tags)
        os.replace(tags_f + ".tmp.npy", tags_f)

    def _load_bounds_sidecar(self, saved):
        """-> the slot books after restoring what the sidecar vouches for (see above)."""
        data_f, tags_f = self._bounds_sidecar()
        out = [-1] * len(saved)
        if not (os.path.exists(data_f) and os.path.exists(tags_f)):
            return out, 0
        try:
            mm, tags = np.load(data_f, mmap_mode="r"), np.load(tags_f)
        except Exception:
            return out, 0
        if mm.shape != (len(saved), int(self.n_walkers), self.spec.d) or len(tags) != len(saved):
            return out, 0
        n = 0
        for k, j in enumerate(saved):
            if j >= 0 and int(tags[k]) == j:
                self.engine.bounds_set_slot(k, np.array(mm[k]))
                out[k] = j
                n += 1
        return out, n

    def _load_checkpoint(self):
        """Resume (sampler.py:291-310, mcmc.py:131-139, 189-214): same number of processes and
        walkers required; seed, proposal covariance, walkers, counters and the R-1 window come
        back from the state file, the rows of the earlier legs from the chain file (cut back
        to the number of rows the state file vouches for)."""
        z = np.load(self._state_file(), allow_pickle=False)
        book, fbook = z["book"], z["fbook"]
        if int(book[8]) != self.size or int(book[9]) != int(self.n_walkers):
            self._fail("Cannot resume a run with a different number of chains: was "
                       "%d processes x %d walkers and now is %d x %d.",
                       int(book[8]), int(book[9]), self.size, int(self.n_walkers))
        if "geometry" in z and tuple(int(v) for v in z["geometry"]) != (
                int(self.group_size), int(self.basis_group_size)):
            self._fail("Cannot resume: the run was written with group_size %d / basis_group_size %d "
                       "and now has %d / %d (the walkers' variate streams and the R-1 groups "
                       "depend on them).", int(z["geometry"][0]), int(z["geometry"][1]),
                       int(self.group_size), int(self.basis_group_size))
        if "thin_carry" in z and not self._device_thin:
            self._thin_carry = z["thin_carry"].astype(np.int64)
        self.engine.set_proposal_cov(z["proposal_cov"])
        self.engine.set_full_state({k: z[k] for k in ("x", "logpost", "logprior", "loglike",
                                                      "weight", "prior_rej", "burn_left",
                                                      "n_accept", "step", "y", "amode",
                                                      "thin_carry") if k in z})
        self._shift = z["shift"]
        self.engine.set_moment_shift(self._shift)
        if "acc_n" in z:   # snapshots accumulated on the device since the last read-out
            self.engine.set_moments(int(z["acc_n"]), z["acc_gs"], z["acc_S"])
            self._snaps_in_interval = int(z["acc_n"])   # (snapshots since the last request)
        if "bbook" in z and self._bslots:
            self._bstride, self._bsnap_idx = (int(v) for v in z["bbook"])
            saved = [int(j) for j in z["bslots"]]
            held = [k for k, j in enumerate(saved) if j >= 0]
            if "bring" in z and len(saved) == len(self._bslots):
                self._bslots = saved
                for k, x in zip(held, z["bring"]):
                    self.engine.bounds_set_slot(k, x)
            elif held and len(saved) == len(self._bslots):
                self._bslots, n_back = self._load_bounds_sidecar(saved)
                if n_back < len(held):
                    self.log.info("%d of the %d snapshots behind R-1 of the bounds were not found "
                                  "beside the state file: those slots restart empty.",
                                  len(held) - n_back, len(held))
            elif held:
                self.log.info("bounds_snapshots changed since the checkpoint: the ring behind R-1 "
                              "of the bounds restarts empty.")
        self._intervals = [(int(n), gs, S) for n, gs, S in zip(z["iv_n"], z["iv_gs"], z["iv_S"])]
        self._iv0 = int(z["iv0"]) if "iv0" in z else 0
        self._wsums = WindowSums()
        (self.n_steps_raw, self.i_learn, self._acc_last, self._steps_last, self._launches,
         self._dropped_snapshots, self._accepted_total) = (int(v) for v in book[:7])
        self._next_ckpt = int(book[10]) or None
        self._ckpt_steps_last = self._steps_last     # (steps per walker at the last request)
        if len(book) > 12:
            self._snap_stride, self._snap_count = int(book[11]), int(book[12])
        txt_rows = int(book[13]) if len(book) > 13 else 0
        if int(book[7]) != int(self.seed):
            self.log.warning("Resuming with the seed of the checkpoint (%d), not %d",
                             int(book[7]), int(self.seed))
            self.seed = int(book[7])
        self._acc_rate, self.Rminus1_last = float(fbook[0]), float(fbook[1])
        was_converged = bool(fbook[2])
        self.learn_proposal_Rminus1_max = float(fbook[3])
        old_stop = [float(v) for v in fbook[4:4 + len(self.CONVERGE_OPTIONS)]]
        new_stop = [float(getattr(self, k)) for k in self.CONVERGE_OPTIONS]
        self.converged = was_converged and old_stop == new_stop
        if was_converged and not self.converged:
            self.log.info("The convergence criteria changed since the checkpoint: the run is "
                          "continued.")
        for i, prow in enumerate(z["progress"], start=1):
            self._progress_rows[i] = {c: (v if c == "timestamp" else float(v))
                                      for c, v in zip(self.PROGRESS_COLUMNS, prow)}
        self._load_chain_file(txt_rows)
        self.log.info("Resumed from %s at %d steps per walker (%d stored rows).",
                      self._state_file(), self.n_steps_raw, self._txt_rows)

    def _load_chain_file(self, n_rows):
        """The first `n_rows` rows of this process' chain file become the head of the
        collection; anything after them (written after the last state dump) is cut off."""
        path = self._chain_file()
        self._txt_rows, self._carried = 0, None
        if not n_rows or not os.path.exists(path):
            if os.path.exists(path):
                os.remove(path)
            return
        with open(path, encoding="utf-8") as f:
            lines = f.readlines()
        header, data = lines[:1], lines[1:]
        if len(data) < n_rows:
            self._fail("The chain file %s holds %d rows but the checkpoint expects %d: "
                       "cannot resume.", path, len(data), n_rows)
        if len(data) > n_rows:
            with open(path, "w", encoding="utf-8") as f:
                f.writelines(header + data[:n_rows])
        self._carried = np.loadtxt(path, ndmin=2)
        self._txt_rows = n_rows

    # ------------------------------------------------------------------ storage
    def _like_names(self):
        return [c["name"] for c in self.spec.components] or [self.spec.like_name]

    def _thin_rows(self, rows):
        """OneSamplePoint.add_to_collection (collection.py:1362-1383) for emitted chain rows:
        a walker's weights accumulate; a row is written when the sum reaches `output_thin`,
        with weight sum // output_thin, the remainder carried to its next rows."""
        thin = getattr(self, "row_thin", None) or self.output_thin
        order = np.argsort(rows[:, 0], kind="stable")
        rows = rows[order]
        ids = rows[:, 0].astype(np.int64)
        w = rows[:, 1].astype(np.int64)
        cum = np.cumsum(w)
        first = np.r_[0, np.flatnonzero(np.diff(ids)) + 1]
        base = np.repeat(cum[first] - w[first], np.diff(np.r_[first, len(ids)]))
        off = self.rank * int(self.n_walkers)     # rows carry GLOBAL walker ids
        carry = self._thin_carry[ids[first] - off]
        total = cum - base + np.repeat(carry, np.diff(np.r_[first, len(ids)]))
        q, q_prev = total // thin, (total - w) // thin
        last = np.r_[first[1:] - 1, len(ids) - 1]
        self._thin_carry[ids[last] - off] = total[last] % thin
        keep = q > q_prev
        out = rows[keep].copy()
        out[:, 1] = (q - q_prev)[keep]
        return out

    def _size_drain_ring(self, d, W):
        """emit: chains, zero-copy drain: the engine's ring of pinned host slots IS the sample
        store as long as it outlives the retention window -- a stored block is dropped (oldest
        half when `max_rows` is reached, `_store_rows`) before its slot comes round again, so no
        row is ever copied a second time on the host.  Slots for the window's launches + 2 at a
        LOW acceptance rate (8 %), within `drain_ring_bytes` of pinned memory at a high one (35 %);
        a run outside that range simply has its oldest views copied out in time
        (`_expire_row_views`)."""
        eng = self.engine
        if (self.emit != "chains" or self.drain_copy or self.max_rows <= 0
                or not hasattr(eng, "set_drain_slots")):
            return
        # (the window holds max_rows / rows-per-launch blocks: MANY when few steps are accepted,
        # while every slot grows to one launch's rows: LARGE when many are)
        steps = W * float(self.steps_per_launch)
        want = int(np.ceil(self.max_rows / max(1.0, 0.08 * steps))) + 2
        fit = int(float(self.drain_ring_bytes) // (1.25 * max(1.0, 0.35 * steps) * 8 * (d + 5)))
        n = int(min(64, max(4, min(want, fit))))
        if n != getattr(eng, "drain_slots", 4):
            eng.set_drain_slots(n)

    def _expire_row_views(self):
        """Before a drain: blocks of `_rows` that are views of the engine's pinned slots stay
        readable for `drain_slots - 1` further drains only; the ones that would not survive the
        drain that comes are copied out of their slot here (they stay in the store, which only
        `max_rows` bounds -- `_store_rows`)."""
        views = [i for i, r in enumerate(self._rows) if self._is_slot_view(r)]
        keep = max(0, getattr(self.engine, "drain_slots", 4) - 2)
        for i in views[:max(0, len(views) - keep)]:
            self._rows[i] = np.array(self._rows[i])

    @staticmethod
    def _is_slot_view(r):
        return getattr(r, "base", None) is not None and not r.flags.owndata and not r.flags.writeable

    def _materialise_row_views(self):
        """Every stored block becomes the sampler's own memory (the engine's pinned slots die
        with it: `close`)."""
        self._rows = [np.array(r) if self._is_slot_view(r) else r for r in self._rows]

    def _store_rows(self, rows, view=False):
        """Keeps at most `max_rows` rows per process WITHOUT freezing: the bounds criterion
        (mcmc.py:918-1002) looks at the later half of the stored samples, so the store must
        keep following the run.  Snapshots: when full, every other stored snapshot is dropped
        and from then on only every second (fourth, ...) snapshot is kept -- a uniformly
        thinned record of the whole run.  Chains: the oldest half of the rows is dropped.
        `view`: the block is a read-only view of an engine-owned pinned slot (zero-copy drain);
        it is kept as such unless something has to outlive the slot (the chain file's pending
        rows, thinned rows)."""
        row_thin = getattr(self, "row_thin", None) or self.output_thin
        if len(rows) and self.emit == "chains" and row_thin > 1 and not getattr(self, "_device_thin", False):
            rows = self._thin_rows(rows)
            view = False
        if not len(rows) or self.max_rows <= 0:
            return
        if view and self.output:
            rows, view = np.array(rows), False    # the chain file receives them later
        if self.emit == "chains" and not view:
            # within a launch: chain after chain; launches follow each other in time -- the
            # order of the collection and of the chain file alike
            if np.any(rows[1:, 0] < rows[:-1, 0]):   # (the engine already drains in this order)
                rows = rows[np.argsort(rows[:, 0], kind="stable")]
        if self._n_rows + len(rows) > self.max_rows and len(self._rows) > 1:
            if self.emit == "snapshots":
                self._rows = self._rows[1::2]
                self._snap_stride *= 2
            else:
                self._rows = self._rows[len(self._rows) // 2:]
            self._n_rows = sum(len(r) for r in self._rows)
            if not self._rows_capped:
                self._rows_capped = True
                self.log.info("max_rows (%d) reached: older stored samples are thinned out as the "
                         "run goes on.", self.max_rows)
        if self._n_rows + len(rows) <= self.max_rows or not self._rows:
            self._rows.append(rows)
            self._n_rows += len(rows)
            if self.output:   # the chain file receives every stored row, also those that the
                self._pending.append(rows)  # in-memory store thins out later

    def _snapshot(self):
        """Thinned sample emission: the current point of every walker with weight 1 (the
        ensemble analogue of `output_thin`, collection.py:1362-1372)."""
        self._since_snapshot = 0
        self._snap_count += 1
        if self.max_rows <= 0 or self._snap_count % self._snap_stride:
            return
        s = self.engine.get_state()
        W = len(s["x"])
        ids = self.rank * W + np.arange(W, dtype=np.float64)
        rows = np.column_stack((ids, np.ones(W), s["logpost"], s["logprior"], s["loglike"],
                                s["x"]))
        self._store_rows(rows)

    def _update_counters(self):
        c = self.engine.counters()
        buf = np.array([float(c["accepted"])])
        dist.all_reduce_sum(buf)
        self._accepted_total = int(buf[0])
        return c

    # ------------------------------------------------------------------ a15 + a16
    def _window(self, sums=True):
        """Statistics over the later half of the run ([n/2:], mcmc.py:787-790) at interval
        granularity: the shortest suffix of checkpoint intervals holding >= half of all
        snapshots taken so far.  The window start only moves forward, so earlier intervals
        are dropped (their snapshot count is remembered).  `sums=False`: the books only (the
        device summed the same window)."""
        if getattr(self, "_wsums", None) is None:   # (bookkeeping set up by hand: tests)
            self._iv0, self._wsums = 0, WindowSums()
        ivs = self._intervals
        counts = [iv[0] for iv in ivs]
        total = self._dropped_snapshots + sum(counts)
        k, rest = 0, sum(counts)
        while k + 1 < len(ivs) and rest - counts[k] >= total / 2:
            rest -= counts[k]
            k += 1
        self._dropped_snapshots += sum(counts[:k])
        self._intervals = ivs = ivs[k:]
        self._iv0 += k
        if k:
            self._wsums.forget_below(self._iv0)
        if not sums:
            return rest, None, None
        i0 = self._iv0
        gsum, Ssum = self._wsums.total(i0, i0 + len(ivs), lambda i: ivs[i - i0][1:])
        return rest, gsum, Ssum

    def check_convergence_and_learn_proposal(self, moments=None, dev=None, payload=None):
        """mcmc.py:773-1032 on pooled sufficient statistics; one all-reduce.
        `moments`: what `engine.fetch_moments()` returned for the checkpoint (None: read them
        out now, synchronously).  `dev`: the outcome of the same checkpoint solved ON THE DEVICE
        (`engine.checkpoint_fetch()`): R-1, the mean of covariances and whether the proposal was
        refreshed there -- the host then only keeps the books (window, progress table, stop
        criteria) and logs.  `payload`: the statistics of the same checkpoint formed AND
        all-reduced on the device (`engine.checkpoint_fetch_payload()`): the host solves them."""
        d, eng = self.spec.d, self.engine
        if moments is None:
            n_snap, gs, S = eng.read_moments(reset=True)  # synchronises the stream
            eng.sync()
            c = eng.counters()
        else:
            n_snap, gs, S, c = moments
        if n_snap:
            self._intervals.append((n_snap, gs, S))
        if not self._intervals:
            return
        gsz = eng.group_size
        from_device = payload is not None
        if dev is None:
            if payload is None:
                n, gsum, Ssum = self._window()
                N_c = float(n * gsz)                       # samples per chain (= group)
                means = gsum / N_c                         # [G, d], relative to the shift
                sum_mm = means.T @ means
                payload = np.concatenate((
                    [float(eng.G), N_c * eng.G, float(c["accepted"] - self._acc_last),
                     float((c["steps"] - self._steps_last) * eng.W), float(c["accepted"])],
                    (Ssum - N_c * sum_mm).ravel(), means.sum(0), sum_mm.ravel()))
                dist.all_reduce_sum(payload)               # RCCL over xGMI when size > 1
            else:
                self._window(sums=False)  # (the books only; the device summed the same window, in stream order)
            n_chains, sum_N, d_acc, d_steps, n_acc_all = payload[:5]
            sum_Ncov = payload[5:5 + d * d].reshape(d, d)
            sum_mean = payload[5 + d * d:5 + d * d + d]
            sum_mm = payload[5 + d * d + d:].reshape(d, d)
        else:
            self._window(sums=False)      # (the books only: which intervals the window holds from now on)
            d_acc, d_steps, n_acc_all = dev["d_accepted"], dev["d_steps"], dev["accepted"]
        self._acc_last, self._steps_last = c["accepted"], c["steps"]
        if dev is None and not from_device and getattr(self, "_device_ckpt", False):
            # a checkpoint without new snapshots took the host path: the device's own copy of
            # "accepted at the last checkpoint" must follow, or the next device checkpoint
            # would report the accepted steps of two intervals over the steps of one
            eng.checkpoint_set_accepted(self._acc_last)
        acceptance_rate = d_acc / max(d_steps, 1.0)
        self._acc_rate = acceptance_rate
        self._accepted_total = int(n_acc_all)
        row = {"N": float(self._accepted_total),
               "timestamp": datetime.datetime.now().isoformat(),
               "acceptance_rate": float(acceptance_rate), "Rminus1": np.nan,
               "Rminus1_cl": np.nan}
        self._progress_rows[self.i_learn] = row
        self.log.info("Learn + convergence test @ %d samples accepted.", self._accepted_total)
        self.log.info(" - Acceptance rate: %.3f", acceptance_rate)
        try:
            if dev is None:
                Rminus1, mean_of_covs = gelman_rubin(n_chains, sum_N, sum_Ncov, sum_mean, sum_mm)
            elif dev["status"] != 0:
                raise NotPositiveDefinite(dev["status"], "device checkpoint: matrix not positive definite")
            else:
                Rminus1, mean_of_covs = dev["Rminus1_groups"], dev["mean_of_covs"]
        except NotPositiveDefinite:
            self.log.warning("Negative covariance eigenvectors. This may mean that the covariance of "
                        "the samples does not contain enough information at this point. "
                        "Skipping learning a new covmat for now.")
            return
        # A chain of the statistic is a GROUP of `gsz` walkers: the variance of its mean is
        # 1/gsz of a single walker's.  Expressed per walker (x gsz), R-1 keeps the reference's
        # meaning -- roughly one over the number of independent samples EACH chain has drawn
        # -- so `Rminus1_stop`, `learn_proposal_Rminus1_max` ... mean what they mean in
        # mcmc.yaml, and a transient shared by all groups (they start from one ref pdf)
        # cannot pass for convergence just because group means average it out.
        Rminus1 = float(Rminus1) * gsz
        row["Rminus1"] = float(Rminus1)
        self.log.info(" - Convergence of means: R-1 = %f after %d accepted steps", Rminus1,
                      self._accepted_total)
        # means criterion twice in a row (mcmc.py:908), then the bounds criterion (918-1002)
        if max(Rminus1, self.Rminus1_last) < self.Rminus1_stop:
            Rcl = self._rminus1_of_bounds(mean_of_covs) if self._bslots else None
            if Rcl is None and not self._bslots:
                self.log.info("bounds_snapshots: 0 -- no bounds criterion: convergence judged on "
                              "the means only.")
                self.converged = True
            elif Rcl is None:     # mcmc.py:996-1001
                self.log.info("Computation of the bounds was not possible. Waiting until the "
                              "next converge check.")
            else:
                row["Rminus1_cl"] = float(Rcl)
                self.log.info(" - Convergence of bounds: R-1 = %f after %d accepted steps", Rcl,
                         self._accepted_total)
                self.converged = Rcl < self.Rminus1_cl_stop
            if self.converged:
                self.log.info("The run has converged!")
        self.Rminus1_last = Rminus1
        if self.learn_proposal and not self.converged:
            if Rminus1 > self.learn_proposal_Rminus1_max:
                self.log.info("Convergence less than requested for updates: waiting until the next "
                         "convergence check.")
            elif Rminus1 < self.learn_proposal_Rminus1_min:
                self.log.info("Convergence better than `learn_proposal_Rminus1_min`: covmat will "
                         "not be updated.")
            elif dev is not None:
                # (the device refreshed its transform in stream order when it solved the
                # checkpoint: the launches queued since already propose with it)
                if dev["refreshed"]:
                    self.log.info(" - Updated covariance matrix of proposal pdf.")
                else:
                    self.log.debug("Updating covariance matrix failed unexpectedly. waiting until "
                                   "next covmat learning attempt.")
            else:
                try:
                    eng.set_proposal_cov(mean_of_covs)  # is already tempered (mcmc.py:1023)
                    self.log.info(" - Updated covariance matrix of proposal pdf.")
                except NotPositiveDefinite:
                    self.log.debug("Updating covariance matrix failed unexpectedly. waiting until "
                              "next covmat learning attempt.")

    def _rminus1_of_bounds(self, mean_of_covs):
        """R-1 of the confidence-interval bounds (mcmc.py:918-1002) ON THE DEVICE, in every emit
        mode: per chain (= walker group) the lower / upper bound of every parameter as GetDist's
        `confidence(i, limfrac=Rminus1_cl_level / 2, upper=which)` gives them (mcmc.py:927-929) for
        the chain's samples in the window's ring snapshots -- exact order statistics, selected
        in LDS (`ckpt_bounds_kernel`) --, summed over the chains of all ranks (one all-reduce,
        in stream order with the library's communicator; mcmc.py:957 gathers the bounds
        instead), then statistic = max_i std_chains(bound_i) / sigma_i (mcmc.py:977-979).
        GetDist is absent from the build container: PARITY UNPINNED for this number (the
        oracle restates its published `confidence`, oracle/ref_numpy.py).  Returns None when the
        ring holds no snapshot of the window."""
        d, eng = self.spec.d, self.engine
        start = self._bsnap_idx / 2.0
        window = sorted((j, k) for k, j in enumerate(self._bslots) if j >= start)
        if not window:
            return None
        stats = eng.bounds_statistics([k for _, k in window], self.Rminus1_cl_level / 2.0)
        if self.size > 1 and not getattr(eng, "comm_attached", False):
            dist.all_reduce_sum(stats)     # (the gloo stand-in; RCCL: reduced in stream order)
        m = stats[0]
        if m < 2:
            return None
        mean_lo, mean_hi = stats[1:1 + d] / m, stats[1 + d:1 + 2 * d] / m
        var_lo = np.maximum(stats[1 + 2 * d:1 + 3 * d] / m - mean_lo ** 2, 0.0)  # np.std: ddof 0
        var_hi = np.maximum(stats[1 + 3 * d:] / m - mean_hi ** 2, 0.0)
        sig = np.sqrt(np.diag(mean_of_covs))
        # NOT rescaled per walker (unlike R-1 of the means, which certifies the mixing): this
        # one is a precision test of the sample's quantiles, and the unit whose bounds must
        # agree is the group
        return float(max(np.max(np.sqrt(var_lo) / sig), np.max(np.sqrt(var_hi) / sig)))

    # ------------------------------------------------------------------ products
    def _table_collection(self, rows):
        """(walker, weight, logpost, logprior, loglike, x...) rows -> SampleCollection with
        the derived parameters (device) and the per-likelihood chi2 columns (host) filled."""
        spec = self.spec
        coll = SampleCollection(spec.sampled, spec.derived, self._like_names(), self.temperature,
                                name=str(1 + self.rank))
        if len(rows):
            derived = None
            if spec.derived:
                derived = np.vstack([self.engine.evaluate(rows[i:i + 65536, 5:], derived=True)[2]
                                     for i in range(0, len(rows), 65536)])
            parts = (spec.component_loglikes(rows[:, 5:]) if len(spec.components) > 1 else None)
            coll.add_rows(rows[:, 1], rows[:, 2], rows[:, 5:], rows[:, 3], rows[:, 4], derived,
                          parts)
        coll.chain_ids = rows[:, 0].astype(np.int64) if len(rows) else np.zeros(0, np.int64)
        return coll

    def _build_collection(self):
        """All rows this process holds: those of earlier legs (read back from the chain file
        at resume) followed by the ones stored in memory."""
        d = self.spec.d
        coll = self._table_collection(np.vstack(self._rows) if self._rows
                                      else np.zeros((0, d + 5)))
        self._chain_ids = coll.chain_ids
        if self._carried is not None and len(self._carried):
            coll._blocks.insert(0, self._carried)
            coll._data = None
            self._chain_ids = np.concatenate((np.full(len(self._carried), -1, np.int64),
                                              coll.chain_ids))
        return coll

    def samples(self, combined=False, skip_samples=0, to_getdist=False):
        """mcmc.py:1092-1148.  `to_getdist` needs the Cobaya-hosted class (it goes through
        `cobaya.collection.SampleCollection.to_getdist`)."""
        if self.temperature != 1 and not to_getdist:
            self.log.warning("The MCMC chain(s) are stored with temperature != 1. Keep that in "
                             "mind when operating on them, or detemper (in-place) with "
                             "products()['sample'].reset_temperature()'.")
        coll = self._build_collection()
        if skip_samples:
            n0 = int(skip_samples * len(coll)) if skip_samples < 1 else int(skip_samples)
            arr = coll.data.to_numpy()[n0:]
            coll._blocks, coll._data = [arr], None
        if (combined or to_getdist) and self.size > 1:
            blocks = dist.gather_rows(coll.data.to_numpy())
            if self.rank == 0:
                coll._blocks, coll._data = [np.vstack(blocks)], None
        out = self._export_collection(coll)
        if to_getdist:
            if not hasattr(out, "to_getdist"):
                self._fail("GetDist export needs Cobaya and GetDist (use `sampler: mcmc_hip` "
                           "inside cobaya.run); or write the chain with `output` and load it "
                           "with GetDist")
            return out.to_getdist()
        return out

    def products(self, combined=False, skip_samples=0, to_getdist=False):
        """mcmc.py:1150-1184: {"sample": SampleCollection, "progress": DataFrame}."""
        self.collection = self.samples(combined, skip_samples, to_getdist)
        return {"sample": self.collection, "progress": self.progress}

    # ------------------------------------------------------------------ reference-style views
    class _ProposerView:
        def __init__(self, engine, scale):
            self._e, self._s = engine, scale

        def get_covariance(self):
            return self._e.get_proposal_cov()

        def get_scale(self):
            return self._s

    @property
    def proposer(self):
        return self._ProposerView(self.engine, float(self.proposal_scale))

    @property
    def current_point(self):
        return self.engine.get_state()

    def info(self):
        """sampler.py:324-330: the options the sampler was set up with, plus what is only
        known after initialisation (mcmc.py:391: the blocking actually used, so that a
        resumed run repeats it)."""
        if hasattr(self, "_updated_info"):      # Cobaya-hosted: Sampler.info()
            out = copy.deepcopy({k: v for k, v in self._updated_info.items()
                                 if not callable(v)})
            out.update({k: v for k, v in self._updated_info.items() if callable(v)})
        else:
            out = {k: getattr(self, k) for k in {**MCMC_DEFAULTS, **HIP_DEFAULTS}}
        if hasattr(self, "blocks"):
            out["blocking"] = [[int(o), list(b)]
                               for o, b in zip(self.oversampling_factors, self.blocks)]
        return out

    # ------------------------------------------------------------------ output
    def _flush_rows(self):
        """collection.py:1268-1315 `out_update`: append the rows stored since the last flush
        to `prefix.<rank+1>.txt` (header first when the file is new).  An existing file is
        only ever appended to; a fresh run starts by removing a stale one."""
        if not self.output:
            return
        path = self._chain_file()
        if self._txt_rows == 0 and os.path.exists(path):
            os.remove(path)
        blocks, self._pending = self._pending, []
        if not blocks:
            if self._txt_rows == 0:
                self._table_collection(np.zeros((0, self.spec.d + 5))).to_txt(path)
            return
        coll = self._table_collection(np.vstack(blocks))
        if self._txt_rows == 0:
            coll.to_txt(path)
        else:
            with open(path, "a", encoding="utf-8") as out:
                np.savetxt(out, coll.data.to_numpy(dtype=np.float64),
                           fmt=[f"%{max(15, len(c))}.8g" for c in coll.columns])
        self._txt_rows += len(coll)

    def close(self):
        if self.engine is not None:
            self._materialise_row_views()   # products()/samples() stay valid after close
            self.engine.close()
            self.engine = None


class MCMCHip(EnsembleMCMC):
    """Standalone host (no Cobaya needed): the constructor of `Sampler.__init__`
    (sampler.py:257-264) plus `resume` / `force`, which Cobaya keeps in its `Output` object
    (output.py:447-495, sampler.py:417-458 `check_force_resume`)."""

    log = log
    _output = _model = spec = None
    _resume = False
    _name = "mcmc_hip"

    def __init__(self, info_sampler=None, model=None, output=None, packages_path=None,
                 name=None, resume=False, force=False):
        info_sampler = dict(info_sampler or {})
        known = {**MCMC_DEFAULTS, **HIP_DEFAULTS}
        unknown = set(info_sampler) - set(known)
        if unknown:  # input.py:403-435: unknown options are rejected
            self._fail("mcmc_hip does not recognise the option(s) %s. Valid options: %s",
                       sorted(unknown), sorted(known))
        for k, v in known.items():
            setattr(self, k, copy.deepcopy(info_sampler.get(k, v)))
        self._name = name or "mcmc_hip"
        self._output = output or None
        self._resume = bool(resume)
        self.packages_path = packages_path
        if resume and force and output:
            self._fail("Make 'resume: True' or 'force: True', not both at the same time: "
                       "can't simultaneously overwrite a chain and resume from it.")
        if isinstance(model, ProblemSpec):
            self.spec, self._model = model, None
        elif model is not None:  # a live cobaya.model.Model: read in initialize()
            self.spec, self._model = None, model
        else:
            self._fail("mcmc_hip needs a model")
        self.converged = False
        self.Rminus1_last = np.inf
        self._check_force_resume(bool(force))
        self.initialize()

    @property
    def model(self):
        return self._model

    @property
    def output(self):
        return self._output

    def get_name(self):
        return self._name

    def _old_files(self):
        """Files of an earlier run with this prefix (mcmc.py:1186-1198 output_files_regexps,
        plus the per-process ensemble state)."""
        folder, prefix = self._out_parts()
        if not os.path.isdir(folder):
            return [], []
        head = re.escape(prefix) + (r"[\._]" if prefix else "")
        chain = re.compile(head + r"\d+\.txt$")
        rest = re.compile(head + r"(checkpoint|progress|covmat|\d+\.(state\.npz|bounds\.npy|bounds_tags\.npy))$")
        names = sorted(os.listdir(folder))
        return ([os.path.join(folder, n) for n in names if chain.match(n)],
                [os.path.join(folder, n) for n in names if rest.match(n)])

    def _check_force_resume(self, force):
        """sampler.py:417-458: old chains are an error unless `force` (delete them) or
        `resume` (continue them); `resume` without old chains starts anew."""
        if not self._output:
            return
        if int(os.environ.get("RANK", "0")) != 0:
            return
        chains, rest = self._old_files()
        if force or (self._resume and not chains):
            for f in chains + rest:
                os.remove(f)
        elif chains and not self._resume:
            self._fail("Delete the previous output manually, automatically ('-f', '--force', "
                       "'force: True') or request resuming ('-r', '--resume', 'resume: True')")
