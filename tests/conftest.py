import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(autouse=True)
def _forget_forced_timeouts():
    """Tests that force a cooperative timeout (GPFQ_COOP_SPIN_LIMIT=0) throw the device's launch-API switch
    (gpfq_capi.hip note_contention): clear it behind every test so that each starts from the undisturbed default."""
    yield
    mod = sys.modules.get("quantized_neural_nets_amd._lib")
    if mod is not None:
        mod.lib.gpfq_clear_contention()
