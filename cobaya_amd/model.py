"""Host-side description of the posterior the engine samples: the subset of Cobaya's input
format the analytic hot path covers (SURVEY.md 8b "Python counterpart").

`ProblemSpec.from_info(info)` reads a Cobaya-style info dict (the form of
docs/src_examples/quickstart/gaussian.yaml and tests/common_sampler.py:24-50 in the
reference); `ProblemSpec.from_cobaya_model(model)` introspects a real `cobaya.model.Model`
when Cobaya itself is importable.  Anything outside the supported subset raises
`UnsupportedModel` -- never a silent fallback.

Reference behaviour mirrored here (paths relative to the reference checkout):
  parameter classes        cobaya/parameterization.py (sampled = has `prior`; derived = no value)
  likelihood param routing cobaya/model.py:1115-1335 (prefix rule 1169-1172)
  1-d priors               cobaya/prior.py:464-533 ; cobaya/tools.py:611-717 (min/max | dist,loc,scale)
  reference pdf            cobaya/prior.py:866-961 (initial points), 963-985 (variances)
  gaussian_mixture         cobaya/likelihoods/gaussian_mixture/gaussian_mixture.py:45-163
  gaussian / one           cobaya/likelihoods/gaussian/gaussian.py:30-112 ; likelihoods/one/one.py
"""
from __future__ import annotations

import itertools
import numbers
from dataclasses import dataclass, field

import numpy as np


OVERHEAD_TIME = 0.0003  # conventions.py:141


def sort_parameter_blocks(blocks, speeds, footprints, oversample_power=0.0):
    """Block ordering of tools.py:955-1006 (`sort_parameter_blocks`), from its cost model.

    After the Cholesky mixing, moving a parameter of the block at position i also moves every
    later (faster) block, so it costs one evaluation of each likelihood that any block at
    position >= i feeds: cost_i = sum_l [1/speed_l] * [some block at position >= i has l in
    its footprint].  A block is oversampled by (cost_0 / cost_i) ** oversample_power and an
    ordering is charged sum_i n_i * factor_i * cost_i.  All m! orderings are scored in one
    array pass (lexicographic order, first minimum wins -- the reference's tie-break, pinned
    by golden G11); for oversample_power >= 1 the ordering is the one that is optimal just
    below 1 (tools.py:996-998).  Returns (ordering, cost per position, integer factors)."""
    sizes = np.array([len(b) for b in blocks], dtype=float)
    like_cost = 1.0 / np.asarray(speeds, dtype=float)
    foot = np.asarray(footprints, dtype=float)
    perms = np.array(list(itertools.permutations(range(len(blocks)))), dtype=int)
    # likelihoods touched from position i onwards: reversed running sum of the footprints
    touched = np.cumsum(foot[perms][:, ::-1, :], axis=1)[:, ::-1, :]
    cost = np.minimum(touched, 1.0) @ like_cost                       # [m!, m]
    rank_power = oversample_power if oversample_power < 1 else 1 - 1e-3
    charge = np.sum(sizes[perms] * (cost[:, :1] / cost) ** rank_power * cost, axis=1)
    best = int(np.argmin(charge))
    factors = np.floor((cost[best, 0] / cost[best]) ** oversample_power).astype(int)
    return tuple(int(i) for i in perms[best]), cost[best], factors


class UnsupportedModel(ValueError):
    """The model is outside what the mcmc_hip kernels cover (analytic Gaussian targets,
    uniform/normal separable priors, one parameter block)."""


@dataclass
class RefPdf:
    """Reference pdf of one parameter (prior.py:866-961): a fixed number, a normal, a
    uniform, or None (= sample the prior)."""
    kind: str | None = None  # None | "point" | "norm" | "uniform"
    a: float = np.nan  # point value | loc | min
    b: float = np.nan  # scale | max

    def variance(self):
        if self.kind == "norm":
            return self.b ** 2
        if self.kind == "uniform":
            return (self.b - self.a) ** 2 / 12.0
        return np.nan  # point-like or absent: prior variance is used (prior.py:963-985)


def _parse_1d(info, what, allow_number=False):
    """min/max or dist/loc/scale -> (kind, a, b) with kind in {"uniform","norm"}
    (tools.py:611-717: `min`/`max` are translated to loc/scale of scipy's uniform)."""
    if info is None:
        return None
    if allow_number and isinstance(info, numbers.Real):
        return ("point", float(info), np.nan)
    if isinstance(info, (list, tuple)) and len(info) == 2:
        return ("uniform", float(info[0]), float(info[1]))
    if not isinstance(info, dict):
        raise UnsupportedModel(f"{what}: cannot interpret {info!r}")
    info = dict(info)
    dist = str(info.pop("dist", "uniform")).lower()
    if dist == "uniform":
        if "min" in info or "max" in info:
            lo, hi = float(info.pop("min", 0.0)), float(info.pop("max", 1.0))
        else:
            lo = float(info.pop("loc", 0.0))
            hi = lo + float(info.pop("scale", 1.0))
        if info:
            raise UnsupportedModel(f"{what}: unknown keys {sorted(info)}")
        if not hi > lo:
            raise UnsupportedModel(f"{what}: needs min < max")
        return ("uniform", lo, hi)
    if dist == "norm":
        if "min" in info or "max" in info:
            raise UnsupportedModel(f"{what}: 'min'|'max' used for an unbounded distribution")
        loc, scale = float(info.pop("loc", 0.0)), float(info.pop("scale", 1.0))
        if info:
            raise UnsupportedModel(f"{what}: unknown keys {sorted(info)}")
        return ("norm", loc, scale)
    raise UnsupportedModel(f"{what}: distribution '{dist}' is not supported by mcmc_hip "
                           "(uniform and norm are)")


@dataclass
class ProblemSpec:
    sampled: list  # parameter names in info order (= sampler order, mcmc.py:392-393)
    derived: list
    kinds: np.ndarray  # 0 uniform, 1 norm
    a: np.ndarray
    b: np.ndarray
    periodic: np.ndarray
    refs: list  # RefPdf per sampled parameter
    proposal: list  # `proposal` width or None
    like_name: str = "one"
    like_kind: str = "one"  # one | gaussian_mixture | gaussian | planck_pliklite
    means: np.ndarray | None = None
    covs: np.ndarray | None = None
    weights: np.ndarray | None = None
    normalized: bool = True
    has_derived: bool = False
    labels: dict = field(default_factory=dict)
    components: list = field(default_factory=list)  # one dict per likelihood (_parse_likelihood)
    # planck_pliklite (cobaya_amd.pliklite): the binned data, the linear Cl(theta) stand-in and
    # the position of the calibration parameter among the sampled ones
    binned: object = None
    emulator: object = None
    calib_index: int = -1

    @property
    def d(self):
        return len(self.sampled)

    @property
    def n_modes(self):
        return 0 if self.like_kind in ("one", "planck_pliklite") else len(self.means)

    # ----------------------------------------------------------------- prior facts
    def prior_variances(self):
        return np.where(self.kinds == 0, (self.b - self.a) ** 2 / 12.0, self.b ** 2)

    def reference_variances(self):
        """prior.py:963-985: variance of the ref pdf, prior variance where absent."""
        v = np.array([r.variance() for r in self.refs])
        return np.where(np.isnan(v), self.prior_variances(), v)

    def bounds(self):
        inf = np.inf
        lo = np.where(self.kinds == 0, self.a, -inf)
        hi = np.where(self.kinds == 0, self.b, inf)
        return lo, hi

    def sample_reference(self, n, rng, max_tries=1000):
        """n initial points from the reference pdf, falling back to the prior where no ref
        is given, redrawn until inside the prior support (prior.py:866-961, vectorised over
        walkers; model.get_valid_point's finite-posterior retry is done by the caller)."""
        d = self.d
        lo, hi = self.bounds()
        out = np.empty((n, d))
        todo = np.arange(n)
        for _ in range(max_tries):
            m = len(todo)
            x = np.empty((m, d))
            for i, r in enumerate(self.refs):
                if r.kind == "point":
                    x[:, i] = r.a
                elif r.kind == "norm":
                    x[:, i] = r.a + r.b * rng.standard_normal(m)
                elif r.kind == "uniform":
                    x[:, i] = rng.uniform(r.a, r.b, m)
                elif self.kinds[i] == 0:
                    x[:, i] = rng.uniform(self.a[i], self.b[i], m)
                else:
                    x[:, i] = self.a[i] + self.b[i] * rng.standard_normal(m)
            ok = np.all((x <= hi) & (x >= lo), axis=1)
            out[todo[ok]] = x[ok]
            todo = todo[~ok]
            if not len(todo):
                return out
        if all(r.kind == "point" for r in self.refs):
            raise UnsupportedModel("The reference point provided has null prior. "
                                   "Set 'ref' to a different point or a pdf.")
        raise UnsupportedModel("Could not sample from the reference pdf a point with "
                               f"non-null prior density after {max_tries} tries.")

    # ----------------------------------------------------------------- construction
    @classmethod
    def _from_params(cls, params):
        """The `params` block -> a spec without likelihoods (parameterization.py: sampled =
        has `prior`; derived = neither prior nor value)."""
        sampled, derived, fixed = [], [], {}
        kinds, a, b, per, refs, props, labels = [], [], [], [], [], [], {}
        for name, p in (params or {}).items():
            if isinstance(p, numbers.Real):
                fixed[name] = float(p)
                continue
            if isinstance(p, str) or callable(p):
                raise UnsupportedModel(f"parameter '{name}': function-valued parameters are "
                                       "not supported")
            p = dict(p or {})
            if "value" in p:
                if isinstance(p["value"], numbers.Real):
                    fixed[name] = float(p["value"])
                    continue
                raise UnsupportedModel(f"parameter '{name}': function-valued parameters are "
                                       "not supported")
            if p.get("latex"):
                labels[name] = p["latex"]
            if p.get("prior") is None:
                if p.get("derived", True) is not True:
                    raise UnsupportedModel(f"parameter '{name}': derived functions are not "
                                           "supported")
                derived.append(name)
                continue
            if p.get("drop") or p.get("renames"):
                raise UnsupportedModel(f"parameter '{name}': drop/renames are not supported")
            kind, pa, pb = _parse_1d(p["prior"], f"prior of '{name}'")
            sampled.append(name)
            kinds.append(0 if kind == "uniform" else 1)
            a.append(pa)
            b.append(pb)
            periodic = bool(p.get("periodic", False))
            if periodic and kind != "uniform":
                raise UnsupportedModel(f"Parameter '{name}' cannot be periodic if it is not "
                                       "bounded.")
            per.append(int(periodic))
            ref = _parse_1d(p.get("ref"), f"ref of '{name}'", allow_number=True)
            refs.append(RefPdf(*ref) if ref else RefPdf())
            props.append(p.get("proposal"))
        if fixed:
            raise UnsupportedModel(f"fixed parameters {sorted(fixed)} are not supported by "
                                   "the analytic likelihoods handled here")
        if not sampled:
            raise UnsupportedModel("No parameters being varied for sampler")
        return cls(sampled, derived, np.array(kinds), np.array(a, float), np.array(b, float),
                   np.array(per), refs, props, labels=labels)

    @classmethod
    def from_info(cls, info):
        if info.get("prior"):
            raise UnsupportedModel("external priors (`prior:` block) are arbitrary Python "
                                   "and cannot run on the device")
        if info.get("theory"):
            raise UnsupportedModel("theory codes are out of scope for mcmc_hip")
        spec = cls._from_params(info.get("params"))
        sampled, derived = spec.sampled, spec.derived
        likes = info.get("likelihood") or {}
        if not likes:
            raise UnsupportedModel("no likelihood given (use `one` for prior-only sampling)")
        comps = [cls._parse_likelihood(lname, linfo, sampled, derived, single=len(likes) == 1)
                 for lname, linfo in likes.items()]
        return spec._with_components(comps)

    def _with_components(self, comps):
        """Attach the parsed likelihoods: one is taken as it is, several are merged."""
        spec, cls = self, type(self)
        spec.components = comps
        if len(comps) == 1:
            c = comps[0]
            spec.like_name, spec.like_kind = c["name"], c["kind"]
            spec.means, spec.covs, spec.weights = c["means"], c["covs"], c["weights"]
            if c["means"] is not None and len(c["means"]) > 64:
                # (said here, not by the engine at its first launch: kernels.h kMaxModes; the
                # reference has no cap, gaussian_mixture.py:45-136)
                raise UnsupportedModel(f"likelihood '{c['name']}' has {len(c['means'])} modes: mcmc_hip "
                                       "evaluates mixtures of at most 64 (one register / LDS plane of "
                                       "whitened residuals per mode)")
            spec.normalized, spec.has_derived = c["normalized"], c["has_derived"]
            if c["kind"] == "planck_pliklite":
                spec.binned, spec.emulator = c["binned"], c["emulator"]
                spec.calib_index = c["calib_index"]
            return spec
        # several likelihoods over disjoint parameter sets: the posterior is the product, i.e.
        # ONE mixture whose modes are all combinations of the components' modes, with
        # block-diagonal covariances (what the device evaluates); the per-likelihood chi2
        # columns of the output are recomputed on the host (component_loglikes).
        claimed = [i for c in comps for i in c["idx"]]
        if sorted(claimed) != list(range(spec.d)):
            raise UnsupportedModel(
                "with several likelihoods every sampled parameter must be the input of "
                "exactly one of them (distinct `input_params_prefix`es)")
        for c in comps:
            if c["kind"] == "planck_pliklite":
                raise UnsupportedModel(
                    f"likelihood '{c['name']}' (planck_pliklite) cannot be combined with others")
            if c["kind"] == "one" or not c["normalized"] or c["has_derived"]:
                raise UnsupportedModel(
                    f"likelihood '{c['name']}': only normalized gaussian / gaussian_mixture "
                    "likelihoods without derived parameters can be combined")
        n_modes = int(np.prod([len(c["means"]) for c in comps]))
        if n_modes > 16:
            raise UnsupportedModel(f"the product of the mixtures has {n_modes} modes (max 16)")
        means, covs, weights = [], [], []
        for combo in itertools.product(*[range(len(c["means"])) for c in comps]):
            m, S, w = np.zeros(spec.d), np.zeros((spec.d, spec.d)), 1.0
            for c, k in zip(comps, combo):
                m[c["idx"]] = c["means"][k]
                S[np.ix_(c["idx"], c["idx"])] = c["covs"][k]
                wk = c["weights"]
                wk = np.full(len(c["means"]), 1.0 / len(c["means"])) if wk is None else wk / wk.sum()
                w *= wk[k]
            means.append(m), covs.append(S), weights.append(w)
        spec.like_name, spec.like_kind = comps[0]["name"], "gaussian_mixture"
        spec.means, spec.covs = np.array(means), np.array(covs)
        spec.weights = np.array(weights) if n_modes > 1 else None
        return spec

    # keys Cobaya itself adds to a likelihood block when it updates the input
    # (input.py update_info, model.py:1320-1328, component.py get_versions): accepted so that
    # an `*.updated.yaml` -- or `model.info()` -- can be fed back in
    _FRAMEWORK_KEYS = ("delay", "type", "version", "stop_at_error", "python_path", "params")

    @classmethod
    def _parse_likelihood(cls, lname, linfo, sampled, derived, single):
        """One entry of the `likelihood` block -> component dict (see `_component`).
        Parameter routing: an explicit `input_params` / `output_params` list
        (model.py:1155-1167), else `input_params_prefix` / `output_params_prefix`
        (model.py:1169-1172)."""
        linfo = dict(linfo or {})
        lclass = linfo.pop("class", lname)
        lclass = str(lclass).split(".")[-1].lower().replace("_", "")
        in_prefix = linfo.pop("input_params_prefix", "") or ""
        out_prefix = linfo.pop("output_params_prefix", "") or ""
        inputs, outputs = linfo.pop("input_params", None), linfo.pop("output_params", None)
        speed = linfo.pop("speed", -1)
        for k in cls._FRAMEWORK_KEYS:
            linfo.pop(k, None)
        if lclass == "one":
            linfo.pop("noise", None)
            return cls._component(lname, "one", [], [], sampled, derived, single, speed=speed)
        if lclass in ("planckpliklite", "ttteeelitenative", "ttlitenative"):
            return cls._pliklite_component(lname, linfo, sampled, derived, speed)
        if lclass not in ("gaussianmixture", "gaussian"):
            raise UnsupportedModel(f"likelihood '{lname}' is not one of gaussian_mixture, "
                                   "gaussian, one, planck_pliklite")
        if not isinstance(inputs, (list, tuple)):
            inputs = [p for p in sampled if p.startswith(in_prefix)]  # model.py:1169-1172
        if not isinstance(outputs, (list, tuple)):
            outputs = [p for p in derived if p.startswith(out_prefix)]
        if lclass == "gaussianmixture":
            kw = dict(means=linfo.pop("means", None), covs=linfo.pop("covs", None),
                      weights=linfo.pop("weights", None),
                      has_derived=bool(linfo.pop("derived", False)))
            kind = "gaussian_mixture"
        else:
            kw = dict(means=linfo.pop("mean", None), covs=linfo.pop("cov", None),
                      normalized=bool(linfo.pop("normalized", True)))
            kind = "gaussian"
        if linfo:
            raise UnsupportedModel(f"unknown options for likelihood '{lname}': "
                                   f"{sorted(linfo)}")
        return cls._component(lname, kind, list(inputs), list(outputs), sampled, derived,
                              single, speed=speed, **kw)

    @staticmethod
    def _pliklite_component(lname, linfo, sampled, derived, speed):
        """`planck_pliklite` (base_classes/planck_pliklite.py).  The data: `dataset_file`
        (+ `path`) = the .dataset file of the installed likelihood, read with the files it names
        as the reference does (planck_pliklite.py:44-73, `pliklite.PlikLiteDataset.from_files`:
        text tables, the Fortran-binary covariance), with `dataset_params` overriding its keys;
        or `dataset` = the same contents handed over directly (a `pliklite.PlikLiteDataset`, a
        dict of its fields, a path to an .npz of them, or `{synthetic: seed}` for the
        plik-lite-shaped stand-in -- the Planck zip cannot be downloaded in the build
        container).  Then the options of planck_pliklite.py:33-43
        (`use_cl`, `use_bins`, `bins_for_L_range`, `calibration_param`), and `cl_emulator` = the
        linear stand-in for the theory code (`provider.get_Cl`, planck_pliklite.py:170-178): a
        `pliklite.LinearClEmulator`, a dict / .npz of its fields, or `{synthetic: n}`; its
        parameters are the sampled ones other than the calibration parameter, in order."""
        from . import pliklite as P
        linfo = dict(linfo)

        def load(obj, what):
            if isinstance(obj, str):
                with np.load(obj, allow_pickle=False) as z:
                    return {k: z[k] for k in z.files}
            if not isinstance(obj, dict):
                raise UnsupportedModel(f"likelihood '{lname}': cannot interpret `{what}`")
            return dict(obj)

        # `dataset_params` override the keys of the .dataset file (DataSetLikelihood.py:64,
        # e.g. TT_lite_native.yaml's `use_cl: tt`); the same names given directly as options of
        # the likelihood win over both
        overrides = dict(linfo.pop("dataset_params", None) or {})
        for k in ("use_cl", "use_bins", "bins_for_L_range", "calibration_param"):
            if linfo.get(k) is not None:
                overrides[k] = linfo[k]
            linfo.pop(k, None)
        ds = linfo.pop("dataset", None)
        dataset_file, path = linfo.pop("dataset_file", None), linfo.pop("path", None)
        if ds is None and dataset_file:
            # the real data: plik_lite_v22.dataset and the files it names (planck_pliklite.py:44-73;
            # DataSetLikelihood.py:28-57: `dataset_file` absolute, or relative to `path`)
            import os
            full = dataset_file if os.path.isabs(dataset_file) else os.path.join(path or ".", dataset_file)
            try:
                ds = P.PlikLiteDataset.from_files(full, overrides)
            except (OSError, ValueError, KeyError) as e:
                raise UnsupportedModel(
                    f"likelihood '{lname}': the data set '{full}' could not be read ({e}); "
                    "install planck_2018_pliklite_native and set `path`") from e
        if ds is None:
            raise UnsupportedModel(
                f"likelihood '{lname}': give `dataset_file` (+ `path`: the folder of "
                "plik_lite_v22.dataset), or `dataset`: the arrays / {synthetic: seed}")
        if not isinstance(ds, P.PlikLiteDataset):
            ds = load(ds, "dataset")
            ds = (P.synthetic_dataset(int(ds["synthetic"])) if "synthetic" in ds
                  else P.PlikLiteDataset(**{k: (int(v) if np.ndim(v) == 0 else np.asarray(v))
                                            for k, v in ds.items()}))
        opts = {**{"use_cl": "tt te ee", "use_bins": (), "bins_for_L_range": (),
                   "calibration_param": "A_planck"},
                **{k: v for k, v in ds.options.items() if v not in (None, "", [], ())}, **overrides}

        def as_ints(v):
            return [int(x) for x in (v.split() if isinstance(v, str) else (v or ()))]

        use_cl = opts["use_cl"]
        use_cl = use_cl.lower().split() if isinstance(use_cl, str) else [str(c).lower() for c in use_cl]
        calib = str(opts["calibration_param"])
        try:
            target = P.BinnedGaussian.from_dataset(
                ds, use_cl=use_cl, use_bins=as_ints(opts["use_bins"]),
                bins_for_L_range=as_ints(opts["bins_for_L_range"]), calibration_param=calib)
        except ValueError as e:
            raise UnsupportedModel(f"likelihood '{lname}': {e}") from e
        if calib not in sampled:
            raise UnsupportedModel(f"likelihood '{lname}': the calibration parameter '{calib}' "
                                   "must be a sampled parameter")
        emu = linfo.pop("cl_emulator", None)
        if emu is None:
            raise UnsupportedModel(
                f"likelihood '{lname}': `cl_emulator` is required (theory codes are out of scope: "
                "Cl(theta) is a linear emulator)")
        if not isinstance(emu, P.LinearClEmulator):
            emu = load(emu, "cl_emulator")
            emu = (P.synthetic_emulator(int(emu["synthetic"]), target.lmax) if "synthetic" in emu
                   else P.LinearClEmulator(np.asarray(emu["theta0"], float), np.asarray(emu["D0"], float),
                                           np.asarray(emu["J"], float), list(emu.get("names", []))))
        linfo.pop("aliases", None)
        if linfo:
            raise UnsupportedModel(f"unknown options for likelihood '{lname}': {sorted(linfo)}")
        if derived:
            raise UnsupportedModel("planck_pliklite has no derived parameters")
        d = len(sampled)
        if emu.n != d - 1 or emu.lmax != target.lmax or not 2 <= d <= 32:
            raise UnsupportedModel(
                f"likelihood '{lname}': the emulator has {emu.n} parameters up to l = {emu.lmax}; "
                f"it must serve the {d - 1} sampled parameters other than '{calib}' "
                f"(2 <= d <= 32) up to lmax = {target.lmax}")
        if target.n_bins > 640:
            raise UnsupportedModel(f"likelihood '{lname}': {target.n_bins} bins (max 640)")
        return {"name": lname, "kind": "planck_pliklite", "idx": list(range(d)), "means": None,
                "covs": None, "weights": None, "normalized": True, "has_derived": False,
                "speed": float(speed if speed is not None else -1), "binned": target,
                "emulator": emu, "calib_index": sampled.index(calib)}

    @staticmethod
    def _component(lname, kind, inputs, outputs, sampled, derived, single, means=None,
                   covs=None, weights=None, normalized=True, has_derived=False, speed=-1):
        """Validated description of one likelihood: dict(name, kind, idx, means[K][n],
        covs[K][n][n], weights[K] | None, normalized, has_derived, speed), with the checks of
        gaussian_mixture.py:45-136 / gaussian.py:30-94."""
        comp = {"name": lname, "kind": kind, "idx": [], "means": None, "covs": None,
                "weights": None, "normalized": bool(normalized), "has_derived": False,
                "speed": float(speed if speed is not None else -1)}
        if kind == "one":
            if derived:
                raise UnsupportedModel("derived parameters need a gaussian_mixture with "
                                       "`derived: True`")
            return comp
        foreign = [p for p in inputs if p not in sampled]
        if foreign:
            raise UnsupportedModel(f"likelihood '{lname}': input parameters {foreign} are not "
                                   "sampled parameters")
        if single and list(inputs) != list(sampled):
            raise UnsupportedModel(
                f"likelihood '{lname}' takes {list(inputs)} but all sampled "
                f"parameters {list(sampled)} must feed the likelihood")
        comp["idx"] = [sampled.index(p) for p in inputs]
        d = len(inputs)
        if means is None or covs is None:
            raise UnsupportedModel(
                "You must specify both a mean (or a list of them) and a covariance matrix, "
                "or a list of them." if kind == "gaussian_mixture" else
                "You must specify both a mean and a covariance matrix.")
        means = np.atleast_1d(np.array(means, dtype=float))
        while means.ndim < 2:
            means = means[None]
        covs = np.atleast_1d(np.array(covs, dtype=float))
        while covs.ndim < 3:
            covs = covs[None]
        K = len(means)
        if covs.shape != (K, means.shape[1], means.shape[1]):
            raise UnsupportedModel("The dimensionalities guessed from mean(s) and "
                                   "cov(s) do not match!")
        if means.shape[1] != d:
            raise UnsupportedModel(
                f"The dimensionality is {means.shape[1]} (guessed from given means and "
                f"covmats) but was passed {d} parameters instead.")
        if weights is not None and not np.isscalar(weights):
            weights = np.array(weights, dtype=float)
            if len(weights) != K:
                raise UnsupportedModel("There must be as many weights as components.")
        else:
            weights = None  # equal weights (gaussian_mixture.py:131-132)
        comp["means"], comp["covs"], comp["weights"] = means, covs, weights
        comp["has_derived"] = bool(has_derived)
        if comp["has_derived"]:
            if len(outputs) != d * K or list(outputs) != list(derived):
                raise UnsupportedModel(
                    "The number of derived parameters must be equal to the "
                    f"dimensionality times the number of modes, i.e. {d} x {K} = "
                    f"{d * K}, but was given {len(derived)} derived parameters.")
        elif derived and single:
            raise UnsupportedModel("Derived parameters were requested, but 'derived' "
                                   "option is False.")
        return comp

    # ----------------------------------------------------------------- several likelihoods
    def component_loglikes(self, x):
        """log-likelihood of every likelihood component at the points x[n][d] (host side, for
        the chi2__<name> output columns; gaussian_mixture.py:138-163)."""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        out = np.zeros((len(x), len(self.components)))
        for ic, c in enumerate(self.components):
            if c["kind"] == "one":
                continue
            xs = x[:, c["idx"]]
            K, n = c["means"].shape
            w = c["weights"]
            w = np.full(K, 1.0 / K) if w is None else w / w.sum()
            a = np.empty((len(x), K))
            for k in range(K):
                L = np.linalg.cholesky(c["covs"][k])
                y = np.linalg.solve(L, (xs - c["means"][k]).T)
                norm = n * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(L)))
                a[:, k] = -0.5 * ((norm if c["normalized"] else 0.0) + np.sum(y * y, axis=0))
            amax = a.max(axis=1, keepdims=True)
            out[:, ic] = np.log(np.sum(w * np.exp(a - amax), axis=1)) + amax[:, 0]
        return out

    def param_blocking(self, oversample_power=0.0, split_fast_slow=False):
        """Model.get_param_blocking_for_sampler (model.py:1340-1467) with
        tools.sort_parameter_blocks (tools.py:955-1006): parameters grouped by the likelihood
        they feed, blocks ordered by the cost-optimal permutation, oversampling factors
        floor((cost_0 / cost_b) ** oversample_power).  Returns (blocks of names, factors)."""
        comps = self.components
        speeds = np.array([c["speed"] for c in comps], dtype=float)
        pos = speeds[speeds > 0]
        min_speed = pos.min() if len(pos) else 1.0
        speeds = 1.0 / (1.0 / np.maximum(speeds, min_speed) + OVERHEAD_TIME)
        if len(comps) == 1 or all(not c["idx"] for c in comps[1:]):
            blocks = [list(self.sampled)]
            footprints = [tuple([1] + [0] * (len(comps) - 1))]
        else:
            blocks = [[self.sampled[i] for i in c["idx"]] for c in comps if c["idx"]]
            footprints = [tuple(int(j == ic) for j in range(len(comps)))
                          for ic, c in enumerate(comps) if c["idx"]]
        if not split_fast_slow:
            order, costs, factors = sort_parameter_blocks(blocks, speeds, footprints,
                                                          oversample_power)
            return [blocks[i] for i in order], [int(f) for f in factors]
        if len(blocks) == 1:
            raise UnsupportedModel("Requested fast/slow separation, but all parameters have "
                                   "the same speed.")
        order, costs, _ = sort_parameter_blocks(blocks, speeds, footprints, 0.0)
        blocks_sorted = [blocks[i] for i in order]
        fp_sorted = np.array(footprints)[list(order)]
        per_block = costs - np.concatenate([costs[1:], [0]])
        i_last_slow = int(np.argmax(np.log(per_block[:-1]) - np.log(per_block[1:])))
        split = [sum(blocks_sorted[:i_last_slow + 1], []), sum(blocks_sorted[i_last_slow + 1:], [])]
        fp_split = np.clip(np.array([fp_sorted[:i_last_slow + 1].sum(axis=0),
                                     fp_sorted[i_last_slow + 1:].sum(axis=0)]), 0, 1)
        _, _, factors = sort_parameter_blocks(split, speeds, fp_split, oversample_power)
        # (the reference only WARNS when the fast factor is 1, model.py:1437-1444)
        factors = ([int(factors[0])] * (1 + i_last_slow)
                   + [int(factors[1])] * (len(blocks) - (1 + i_last_slow)))
        return blocks_sorted, factors

    @classmethod
    def from_cobaya_model(cls, model):
        """Introspect a live `cobaya.model.Model` (the attributes SURVEY.md 8b lists): the
        parameter block of `model.info()` (prior / ref / proposal / periodic of every sampled
        parameter, cross-checked against `model.parameterization` and `model.prior`), and the
        INITIALISED likelihood objects `model.likelihood[name]` -- their routed
        `input_params` / `output_params` (model.py:1115-1335) and the arrays they were built
        with (`means`, `covs`, `weights`: gaussian_mixture.py:53-136; `mean`, `cov`,
        `normalized`: gaussian.py:30-94)."""
        info = model.info()
        if info.get("prior") or len(list(model.prior)) > 1:
            raise UnsupportedModel("external priors (`prior:` block) are arbitrary Python "
                                   "and cannot run on the device")
        if len(getattr(model, "theory", None) or {}):
            raise UnsupportedModel("theory codes are out of scope for mcmc_hip")
        spec = cls._from_params(info["params"])
        live = list(model.parameterization.sampled_params())
        if live != spec.sampled:
            raise UnsupportedModel(f"sampled parameters of the model {live} differ from the "
                                   f"ones read from its info {spec.sampled}")
        lo, hi = np.asarray(model.prior.bounds(confidence_for_unbounded=1.0)).T
        mine = spec.bounds()
        if not (np.array_equal(lo, mine[0]) and np.array_equal(hi, mine[1])):
            raise UnsupportedModel("prior bounds of the model and of its info disagree")
        comps = []
        likes = dict(model.likelihood.items())
        for lname, like in likes.items():
            mro = [c.__name__ for c in type(like).__mro__]
            speed = getattr(like, "speed", -1)
            ins, outs = list(like.input_params), list(like.output_params)
            if "GaussianMixture" in mro:
                w = getattr(like, "weights", None)
                comps.append(cls._component(
                    lname, "gaussian_mixture", ins, outs, spec.sampled, spec.derived,
                    len(likes) == 1, means=like.means, covs=like.covs,
                    weights=None if np.isscalar(w) else w, has_derived=bool(like.derived),
                    speed=speed))
            elif "Gaussian" in mro:
                comps.append(cls._component(
                    lname, "gaussian", ins, outs, spec.sampled, spec.derived, len(likes) == 1,
                    means=like.mean, covs=like.cov,
                    normalized=bool(getattr(like, "normalized", True)), speed=speed))
            elif "one" in mro or "One" in mro:
                comps.append(cls._component(lname, "one", [], [], spec.sampled, spec.derived,
                                            len(likes) == 1, speed=speed))
            else:
                raise UnsupportedModel(
                    f"likelihood '{lname}' ({type(like).__name__}) is not one of "
                    "gaussian_mixture, gaussian, one")
        if not comps:
            raise UnsupportedModel("no likelihood given (use `one` for prior-only sampling)")
        return spec._with_components(comps)

    # ----------------------------------------------------------------- engine hookup
    def configure(self, engine):
        engine.set_prior(self.kinds, self.a, self.b, self.periodic)
        if self.like_kind == "one":
            engine.set_target_one()
        elif self.like_kind == "planck_pliklite":
            engine.set_target_binned_gaussian(self.binned, self.emulator, self.calib_index)
        elif self.like_kind == "gaussian":
            engine.set_target_gaussian(self.means[0], self.covs[0], self.normalized)
        else:
            engine.set_target_gaussian_mixture(self.means, self.covs, self.weights)
