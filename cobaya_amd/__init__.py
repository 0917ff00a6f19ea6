"""cobaya_amd: the MI355X-native walker-ensemble Metropolis path behind Cobaya's sampler API.

    from cobaya_amd import run            # run(info) -> (updated_info, sampler)
    from cobaya_amd import MCMCHip        # the sampler class (registered as `mcmc_hip`)
    from cobaya_amd.engine import Engine  # thin wrapper over the C ABI (include/mcmc_hip.h)
"""
from .model import ProblemSpec, UnsupportedModel  # noqa: F401
from .run import run  # noqa: F401
from .sampler import LoggedError, MCMCHip  # noqa: F401

__version__ = "0.1.0"
