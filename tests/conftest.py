import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The parity tests compare with states recorded inside the reference, whose prompt sampler draws through the host generator: they need
# the reference's random stream (univs_amd/switches.py: sampler).  The library default on the GPU is the device-side sampler ("auto");
# tests of that mode select it per encoder.  Set before univs_amd.switches is imported.
os.environ.setdefault("UNIVS_SAMPLER", "reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu test started without a visible GPU")
    return torch.device("cuda:0")


@pytest.fixture
def linear_terms():
    """`linear_terms(3)`: select the fp16 three-product Linear (6: the bf16 six-product one) for the rest of the test."""
    from univs_amd import ops
    prev = {}

    def select(terms):
        if not prev:
            prev.update(ops.get_config())
        ops.configure(linear_terms=int(terms))
    yield select
    if prev:
        ops.configure(**prev)
