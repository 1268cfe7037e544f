import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# puts nerf-navigation_amd/ on sys.path so `import raymarching`, `import gridencoder`, ... resolve to the drop-ins
importlib.import_module("nerf-navigation_amd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ngp_oracle
    ngp_oracle.build()
    return ngp_oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ngp_hip
    ngp_hip.lib()          # raises if libngp_hip.so is missing: GPU tests never run on a fallback
    return torch.device("cuda:0")
