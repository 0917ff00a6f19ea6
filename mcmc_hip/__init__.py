"""Top-level plugin module: makes `sampler: {mcmc_hip: {...}}` resolvable by Cobaya.

cobaya/component.py:598-795 (`get_component_class`) looks for an external module named like
the component on `sys.path` and takes the class returned by its module-level
`get_cobaya_class()` (component.py:677-678) or the class whose name matches case-insensitively
with underscores dropped (component.py:798-807: `mcmc_hip` <-> `MCMCHip`).

With Cobaya importable, the class below derives from `cobaya.samplers.mcmc.MCMC`, so it
inherits every default of mcmc.yaml through `HasDefaults.get_defaults` (component.py:321-326)
and only declares the new options as typed class attributes (unknown keys are rejected,
input.py:403-435).  Without Cobaya it is the standalone `cobaya_amd.MCMCHip`.
"""
from cobaya_amd.sampler import HIP_DEFAULTS
from cobaya_amd.sampler import MCMCHip as _Standalone

try:  # pragma: no cover - exercised only where Cobaya is installed
    from cobaya.samplers.mcmc import MCMC as _CobayaMCMC
except Exception:  # Cobaya absent (e.g. on the GPU box of this build): standalone class
    _CobayaMCMC = None

if _CobayaMCMC is None:
    MCMCHip = _Standalone
else:
    class MCMCHip(_CobayaMCMC):  # type: ignore[misc, valid-type]
        """Cobaya-hosted variant: Cobaya's `Sampler.__init__` (sampler.py:257-322) sets the
        options as attributes, seeds `_rng`, and calls `initialize()`; everything from there on
        is the engine-backed implementation."""

        file_base_name = "mcmc_hip"
        n_walkers: int = HIP_DEFAULTS["n_walkers"]
        group_size: int | None = HIP_DEFAULTS["group_size"]
        device: int | None = HIP_DEFAULTS["device"]
        steps_per_launch: int | str = HIP_DEFAULTS["steps_per_launch"]
        moments_every: int = HIP_DEFAULTS["moments_every"]
        emit: str = HIP_DEFAULTS["emit"]
        snapshot_every: int | None = HIP_DEFAULTS["snapshot_every"]
        max_rows: int = HIP_DEFAULTS["max_rows"]

        def initialize(self):
            from cobaya_amd.model import ProblemSpec, UnsupportedModel
            from cobaya.log import LoggedError
            try:
                self.spec = ProblemSpec.from_cobaya_model(self.model)
            except UnsupportedModel as e:
                raise LoggedError(self.log, "mcmc_hip cannot sample this model: %s", str(e))
            self._name = self.get_name()
            self.converged = False
            self.Rminus1_last = float("inf")
            self.engine = None
            _Standalone.initialize(self)

        _resume = False

    # engine-backed implementation: every method/property of the standalone class that the
    # Cobaya base does not have to keep (life-cycle hooks above excepted) is shared verbatim
    for _name, _attr in vars(_Standalone).items():
        if _name.startswith("__") or _name in ("initialize", "file_base_name", "info",
                                                "get_name"):
            continue
        setattr(MCMCHip, _name, _attr)
    del _name, _attr


def get_cobaya_class():
    return MCMCHip
