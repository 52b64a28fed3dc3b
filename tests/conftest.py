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


def pytest_sessionfinish(session, exitstatus):
    """When the suite runs against a CHECKED build (OVRFSR_LIB=ab/bounds.so: csrc/fsr_bounds.h), read the device-side accessor counters the
    whole session accumulated and write them to gpurun_out/bounds_suite.json -- every GPU test then doubles as a bounds campaign."""
    import ctypes
    import json
    lib_path = os.environ.get("OVRFSR_LIB", "")
    if not lib_path.endswith("bounds.so"):
        return
    try:
        import openvr_fsr_amd as A
        lib = A.library()
        n = lib.ovrfsr_debug_bounds_slots()
        buf = (ctypes.c_ulonglong * n)()
        lib.ovrfsr_debug_bounds.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
        assert lib.ovrfsr_debug_bounds(buf, n, 0) == 0
        nk = (n - 5) // 3
        v = list(buf)
        rec = {"out_of_bounds": sum(v[:nk]), "declared_pad_accesses": sum(v[nk:2 * nk]), "checked_accesses": sum(v[2 * nk:3 * nk]),
               "per_kind_out_of_bounds": v[:nk], "per_kind_pad": v[nk:2 * nk], "per_kind_checked": v[2 * nk:3 * nk], "first_record": v[3 * nk:],
               "pytest_exitstatus": int(exitstatus), "tests_collected": session.testscollected}
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, "bounds_suite.json"), "w"), indent=1)
        print("\n[checked build] %d checked accesses, %d out of bounds, %d declared-pad accesses" % (rec["checked_accesses"], rec["out_of_bounds"], rec["declared_pad_accesses"]))
    except Exception as e:  # noqa: BLE001
        print("\n[checked build] counters unavailable: %r" % (e,))
