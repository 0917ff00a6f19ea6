"""pytest configuration: markers and import paths.

`gpu` marks tests that need a real MI355X (run by the driver with `-m gpu`);
everything else must pass on a CPU-only box (`-m "not gpu"`).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def pytest_sessionstart(session):
    """Developer aid: MCMC_TEST_ON_ORACLE=1 serves the sampler's ctypes seam with the
    oracle-backed double (tests/oracle_engine.py), so that the small `-m gpu` sampler tests can
    be debugged in the CPU-only container (the oracle is the kernels' bit-exact specification).
    Never set by the driver: on the GPU box the real engine runs."""
    if os.environ.get("MCMC_TEST_ON_ORACLE"):
        from cobaya_amd.sampler import EnsembleMCMC
        from tests.oracle_engine import OracleEngine
        EnsembleMCMC._engine_factory = staticmethod(OracleEngine)
