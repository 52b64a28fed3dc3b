"""Checked / audit variants of the library (ab/*.so, never shipped): find a FRESH one or build it.

The object directories do not travel to the GPU box (.gpurunignore), so freshness is a hash stamp written by the build
(tools/variant_fresh.py): a variant built from other sources than the ones next to it proves nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def variant(name, flags, timeout=1500):
    lib = os.path.join(ROOT, "ab", name + ".so")
    fresh = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_fresh.py"), name, flags]).returncode == 0
    if not fresh:
        r = subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), name, flags], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0 or not os.path.exists(lib):
            pytest.skip("%s build unavailable here: %s" % (name, (r.stderr or r.stdout)[-300:]))
    return lib
