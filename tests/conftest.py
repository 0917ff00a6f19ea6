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
        if os.path.exists("/dev/kfd"):
            # a GPU box: the double must never stand in for the device
            pytest.exit("MCMC_TEST_ON_ORACLE is a CPU-container developer aid; unset it on a GPU "
                        "box -- the -m gpu tests must run the HIP kernels", returncode=4)
        from cobaya_amd.sampler import EnsembleMCMC
        from tests.oracle_engine import OracleEngine
        EnsembleMCMC._engine_factory = staticmethod(OracleEngine)


@pytest.fixture(autouse=True)
def _gpu_tests_run_hip_kernels(request, monkeypatch):
    """Every sampler a `-m gpu` test drives must sit on the ctypes engine of libmcmc_hip.so and
    its first launch must report a HIP kernel by name (`mcmc_hip_last_step_kernel`): a sampler-level
    GPU test can then not pass on a stand-in engine, whatever the environment says."""
    if "gpu" not in request.keywords or os.environ.get("MCMC_TEST_ON_ORACLE"):
        yield
        return
    from cobaya_amd.engine import Engine
    from cobaya_amd.sampler import EnsembleMCMC
    advance = EnsembleMCMC.advance

    def checked_advance(self):
        advance(self)
        if not getattr(self, "_hip_kernel_checked", False):
            assert isinstance(self.engine, Engine), \
                f"a -m gpu test ran on {type(self.engine).__name__}, not on the HIP engine"
            name = self.engine.last_step_kernel()
            assert "kernel" in name and "mcmc::" in name, f"no HIP step kernel reported: {name!r}"
            self._hip_kernel_checked = True
    monkeypatch.setattr(EnsembleMCMC, "advance", checked_advance)
    yield
