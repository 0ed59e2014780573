import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import s2vt_oracle
    s2vt_oracle.lib()
    return s2vt_oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import s2vt_amd
    s2vt_amd.lib()          # raises loudly if libs2vt_hip.so is missing: there is no fallback
    from s2vt_amd import ops
    return ops
