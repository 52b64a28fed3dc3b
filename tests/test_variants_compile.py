"""Every measurement macro the Makefile documents, and every variant tools/variants/ can rebuild, still type-checks.

The records under profiles/ cite these builds; a rename that breaks one (round 4: `half2_t` -> `hale2_t` inside the
OVRFSR_HALF_ACC block) makes its measurement unreproducible without anybody noticing.  Syntax-only hipcc passes
(templates are instantiated, no code generation): ~4 s each, run four at a time.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openvr_fsr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# macros that live in the product sources (Makefile header)
IN_TREE = [
    ("fsr_kernels.hip", "-DOVRFSR_TIE_OFF"),
    ("fsr_kernels.hip", "-DOVRFSR_TIE_NOPASS"),
    ("fsr_kernels.hip", "-DOVRFSR_TIE_BITS=8 -DOVRFSR_TIE_HALF_BITS=5"),
    ("fsr_kernels.hip", "-DOVRFSR_EASU_DIR_CONTRACTED"),
    ("fsr_kernels.hip", "-DOVRFSR_FUSED_NT=512"),
    ("fsr_kernels.hip", "-DOVRFSR_TIE_AUDIT"),
    ("postprocessor.cpp", "-DOVRFSR_MUTATE_NO_JOIN"),
]
# variants kept as patches (tools/variants/build.sh)
PATCHED = [
    ("scalar", "-DOVRFSR_EASU_SCALAR"),
    ("px1", "-DOVRFSR_EASU_1PX"),
    ("hacc", "-DOVRFSR_HALF_ACC"),
    ("items1", "-DOVRFSR_FUSED_ITEMS=1"),
    ("items2", "-DOVRFSR_FUSED_ITEMS=2 -DOVRFSR_FUSED_SKIP_VRING"),
    ("narrow", "-DOVRFSR_FUSED_NARROW=1"),
    ("narrow_items", "-DOVRFSR_FUSED_NARROW=1 -DOVRFSR_FUSED_ITEMS=1"),
    ("compact", "-DOVRFSR_NIS_COMPACT"),
    ("nishalf", "-DOVRFSR_NIS_HALF_LDS"),
    ("plain", ""),
    # round 5's scheduling experiments (profiles/r05_sched_ab.txt): each its own patch
    ("soa", "-DOVRFSR_EASU_SOA", "easu_soa"),
    ("fsb", "-DOVRFSR_EASU_FS_BUNDLE -DOVRFSR_EASU_OCC5", "easu_fs_bundle"),
    ("mme", "-DOVRFSR_EASU_MM_EARLY", "easu_fs_bundle"),
    ("rpipe", "-DOVRFSR_RCAS_PIPE", "rcas_pipe"),
    ("px2", "-DOVRFSR_RCAS_PX2", "rcas_px2"),
    ("rlds", "", "rcas_lds_cap"),
    # round 6: the fused kernel as a persistent workgroup prefetching the next tile's texels (profiles/r06_fused_prefetch.txt)
    ("pf", "-DOVRFSR_FUSED_PF_DEFAULT=1 -DOVRFSR_FUSED_PF_WAVES=6", "fused_prefetch"),
]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _syntax(tu, flags):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=on"]
    cmd += ["-x", "hip"] if tu.endswith(".cpp") else ["-fno-slp-vectorize"]
    cmd += flags.split() + ["-fsyntax-only", tu]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    return (tu, flags, r.returncode, r.stderr[-2000:])


def _patched(name, flags, patches=None):
    env = dict(os.environ, PATCHES=patches) if patches else None
    r = subprocess.run([os.path.join(ROOT, "tools", "variants", "build.sh"), name, flags, "--syntax-only"], capture_output=True, text=True, env=env)
    return (name, flags, r.returncode, (r.stdout + r.stderr)[-2000:])


def test_in_tree_macros_compile():
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda a: _syntax(*a), IN_TREE))
    bad = [r for r in res if r[2] != 0]
    assert not bad, bad


@pytest.mark.skipif(shutil.which("patch") is None, reason="patch(1) not installed")
def test_patched_variants_apply_and_compile():
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda a: _patched(*a), PATCHED))
    bad = [r for r in res if r[2] != 0]
    assert not bad, bad
