"""The CHECKED build ("LDS-bounds asserts in debug kernels"; VERDICT r5 next #1a) on the GPU.

-DOVRFSR_BOUNDS (openvr_fsr_amd/csrc/fsr_bounds.h -> ab/bounds.so, built by __graft_entry__.build()) routes every LDS index, every image
byte offset and every device-table index of every kernel through an accessor that knows the extent of what it points into; a violation is
counted on the device instead of faulting.  tools/debug/bounds_campaign.py drives that build over the fuzz seeds of test_gpu_fuzz.py, ragged /
tiny / minification shapes, every format pair, batches with stride gaps, shared textures, pair_submit, BASELINE C1-C5 at full size and the
natural-content fixtures: the claim is 0 out-of-bounds accesses, and pad accesses only where the design declares a pad.
The product library is the same source with the accessors expanded to plain pointers -- identical machine code (tools/isa_fingerprint.py)."""
import os
import re
import subprocess
import sys

import pytest

from tests.variants import ROOT, variant

pytestmark = pytest.mark.gpu


def test_checked_build_finds_no_out_of_bounds_access(gpu):
    lib = variant("bounds", "-DOVRFSR_BOUNDS")
    env = dict(os.environ, OVRFSR_LIB=lib, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "bounds_campaign.py")], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    m = re.search(r"TOTAL configurations (\d+), checked accesses (\d+), OUT OF BOUNDS (\d+), undeclared pad accesses (\d+), declared pad accesses (\d+), selftest (\w+)", r.stdout)
    assert m, (r.stdout[-1500:], r.stderr[-800:])
    configs, checked, oob, badpad, pad, selftest = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)
    # the self-test launch commits every kind of violation once and must be COUNTED exactly: a zero below is then a finding, not silence
    assert selftest == "ok", r.stdout[:600]
    assert r.returncode == 0 and oob == 0 and badpad == 0, r.stdout[-3000:]
    assert configs > 2500 and checked > 5e9, (configs, checked)
    # the canary: the analysis sweep of EASU does read its declared luma pad rows (fsr_params.h kLumPadRows) -- the accessors are live in the
    # kernels, not only in the self-test -- and nothing else than the declared kinds shows pad accesses
    rows = {k: (int(a), int(b), int(c)) for k, a, b, c in re.findall(r"^(K_[A-Z0-9_]+)\s+(\d+)\s+(\d+)\s+(\d+)", r.stdout, re.M)}
    assert rows["K_EASU_LUM"][2] > 0 and rows["K_EASU_LUM"][1] == 0
    for kind in ("K_IMAGE_IN", "K_IMAGE_OUT", "K_EASU_COL", "K_EASU_ANA", "K_FUSED_MID", "K_OUTSIDE_TEX", "K_NIS_Y255", "K_NIS_EDGE", "K_NIS_RAW", "K_NIS_SHARPEN_Y",
                 "K_RCAS_TILE", "K_TILE_LIST", "K_TILE_REC", "K_SPAN_REC", "K_BIL_X", "K_BIL_Y", "K_NIS_COEF", "K_TIE_LIST"):
        assert rows[kind][0] > 0, "no checked access of kind %s: a kernel family did not run" % kind
        assert rows[kind][1] == 0, kind


def test_fuzz_seeds_pass_against_the_checked_build(gpu):
    """The fuzz tests THEMSELVES (tests/test_gpu_fuzz.py: oracle parity on random sizes / scales / masks / pitches / formats, ctx life-cycle
    stress) on 120 fresh seeds against ab/bounds.so: every instance must pass its parity asserts -- a checked access that is wrongly flagged is
    redirected to the plane base and corrupts the checked build's pixels, which is how the checker's one false positive was found (a 3-element
    vector load counted as 16 bytes) -- and the device counters must show 0 out-of-bounds accesses at the end."""
    lib = variant("bounds", "-DOVRFSR_BOUNDS")
    env = dict(os.environ, OVRFSR_LIB=lib, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "debug", "fuzz_campaign.py"), "170000", "120"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    m = re.search(r"seeds 170000\.\.170119: (\d+) failures", r.stdout)
    assert m and int(m.group(1)) == 0, (r.stdout[-2500:], r.stderr[-500:])
    m = re.search(r"checked build: (\d+) checked accesses, (\d+) OUT OF BOUNDS, (\d+) declared-pad accesses", r.stdout)
    assert m and int(m.group(1)) > 1e8 and int(m.group(2)) == 0, r.stdout[-600:]


def test_injected_resource_failures_leave_outputs_untouched(gpu):
    """FAULT INJECTION (failure detection / recovery: "no fault injection" in the reference).  The checked build can make
    the n-th device allocation / stream / event creation of the launch manager fail; tools/debug/fault_campaign.py walks n over ten pipeline
    configurations and checks the header's promises: a status and an error text, the caller's output image untouched, a failed (re)build
    leaves the ctx DISABLED (PostProcessor.cpp:145-152) and the next apply touches nothing, failures the library can absorb (no auxiliary
    stream: in-order launches) still give correct pixels, and reset restores a ctx that produces exactly the un-failed result."""
    lib = variant("bounds", "-DOVRFSR_BOUNDS")
    env = dict(os.environ, OVRFSR_LIB=lib, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "fault_campaign.py")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    m = re.search(r"TOTAL (\d+) injected failures over (\d+) configurations, (\d+) violations", r.stdout)
    assert m, (r.stdout[-1500:], r.stderr[-800:])
    assert r.returncode == 0 and int(m.group(3)) == 0 and int(m.group(1)) >= 20, r.stdout[-2500:]
