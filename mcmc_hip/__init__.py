"""Top-level plugin module: makes `sampler: {mcmc_hip: {...}}` resolvable by Cobaya.

cobaya/component.py:598-795 (`get_component_class`) looks for an external module named like
the component on `sys.path` and takes the class returned by its module-level
`get_cobaya_class()` (component.py:677-678) or the class whose name matches case-insensitively
with underscores dropped (component.py:798-807: `mcmc_hip` <-> `MCMCHip`).

With Cobaya importable, the class below is `EnsembleMCMC` (all of the engine-backed logic,
`cobaya_amd/sampler.py`) in front of `cobaya.samplers.mcmc.MCMC` in the MRO: Cobaya's own
`Sampler.__init__` (sampler.py:257-322) sets the options, the model and the `Output` object,
loads the checkpoint info when resuming and calls `initialize()`; every life-cycle method
(`initialize`, `run`, `products`, `samples`, `write_checkpoint`, `info`, ...) then resolves to
the ensemble implementation, while the class-level plumbing -- defaults of mcmc.yaml through
`HasDefaults.get_defaults` (component.py:321-326), `check_force_resume` (sampler.py:417-458),
`_at_resume_prefer_new/old`, `get_version`, `_get_desc` -- stays Cobaya's.  Unknown option
keys are rejected by Cobaya (input.py:403-435), hence the typed class attributes for the new
options.  Without Cobaya the module exports the standalone `cobaya_amd.MCMCHip`.
"""
import re

from cobaya_amd.sampler import HIP_DEFAULTS, EnsembleMCMC
from cobaya_amd.sampler import MCMCHip as _Standalone

try:
    from cobaya.samplers.mcmc import MCMC as _CobayaMCMC
except ImportError:  # Cobaya absent (e.g. on the GPU box of this build): standalone class
    _CobayaMCMC = None

if _CobayaMCMC is None:
    MCMCHip = _Standalone
else:
    import pandas as _pd
    from cobaya.collection import SampleCollection as _CobayaCollection
    from cobaya.log import LoggedError as _CobayaLoggedError

    class MCMCHip(EnsembleMCMC, _CobayaMCMC):  # type: ignore[misc, valid-type]
        """`sampler: mcmc_hip` inside a real Cobaya."""

        file_base_name = "mcmc_hip"
        _LoggedError = _CobayaLoggedError
        n_walkers: int = HIP_DEFAULTS["n_walkers"]
        group_size: int | None = HIP_DEFAULTS["group_size"]
        device: int | None = HIP_DEFAULTS["device"]
        steps_per_launch: int | str | None = HIP_DEFAULTS["steps_per_launch"]
        moments_every: int = HIP_DEFAULTS["moments_every"]
        emit: str = HIP_DEFAULTS["emit"]
        snapshot_every: int | None = HIP_DEFAULTS["snapshot_every"]
        max_rows: int = HIP_DEFAULTS["max_rows"]
        bounds_snapshots: int = HIP_DEFAULTS["bounds_snapshots"]
        drain_copy: bool = HIP_DEFAULTS["drain_copy"]
        row_buffer_bytes: int = HIP_DEFAULTS["row_buffer_bytes"]
        drain_ring_bytes: int = HIP_DEFAULTS["drain_ring_bytes"]
        device_checkpoint: bool | str | None = HIP_DEFAULTS["device_checkpoint"]
        shared_basis: bool = HIP_DEFAULTS["shared_basis"]
        evaluation: str = HIP_DEFAULTS["evaluation"]
        basis_group_size: int | None = HIP_DEFAULTS["basis_group_size"]
        checkpoint_lag: int | None = HIP_DEFAULTS["checkpoint_lag"]
        emit_thin: int | None = HIP_DEFAULTS["emit_thin"]

        def _export_collection(self, coll):
            """Our table -> `cobaya.collection.SampleCollection` (same columns,
            collection.py:154-161), so that `products()["sample"]` offers the whole reference
            API (`mean`, `cov`, `to_getdist`, slicing, ...).  It is detached from the output
            driver: the chain file is written by `_flush_rows`."""
            out = _CobayaCollection(self.model, None, name=str(1 + self.rank),
                                    temperature=self.temperature, sample_type="mcmc")
            if list(out.columns) != list(coll.columns):
                self._fail("column contract broken: Cobaya expects %r, mcmc_hip has %r",
                           list(out.columns), list(coll.columns))
            if not len(coll):
                return out
            data = _pd.DataFrame(coll.data.to_numpy(dtype=float), columns=out.columns)
            return out._copy(data=data)

        @classmethod
        def output_files_regexps(cls, output, info=None, minimal=False):
            """mcmc.py:1186-1198 plus the per-process ensemble state files, so that `force`
            cleans them and a stale state never survives a fresh start."""
            regexps = _CobayaMCMC.output_files_regexps(output, info=info, minimal=minimal)
            if not minimal:
                regexps.append((re.compile(output.prefix_regexp_str
                                           + r"\d+\.(state\.npz|bounds\.npy|bounds_tags\.npy)$"),
                                None))
            return regexps


def get_cobaya_class():
    return MCMCHip
