import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the test suite wants the whole build: a missing llvm-objdump (isa_check skipped) or test library is an error here, not a printed note
os.environ.setdefault("CANONSWAP_STRICT_BUILD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def state_dicts_np():
    from canonswap_amd import synth
    return synth.make_state_dicts(0)


@pytest.fixture(scope="session")
def state_dicts(state_dicts_np):
    from canonswap_amd import synth
    return synth.to_torch(state_dicts_np)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load
