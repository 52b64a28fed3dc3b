import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library on a gfx950 device.  Fails loudly (never skips to a fallback) under -m gpu."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    import openvr_fsr_amd
    assert openvr_fsr_amd.have_library(), "libopenvr_fsr_amd.so is not built"
    return torch
