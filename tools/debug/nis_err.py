#!/usr/bin/env python3
"""tools/debug/nis_err.py REF.so VAR.so -- NVScaler (C3 shape, structured and random content): the product build of VAR.so against
the STRICT build of REF.so (bit-identical to the oracle), float and UNORM8 outputs.  Each library runs in its own process."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "dump":
    from tests import synth
    from tests.util import run_gpu
    prec, out = int(sys.argv[2]), sys.argv[3]
    res = {}
    for name, gen in (("structured", synth.structured_u8), ("random", synth.random_u8)):
        img = gen(1683, 1869, 0x5EED0002)
        res[name + "_f"] = run_gpu(img, 2244, 2492, np.float32, precision=prec, use_nis=1, sharpness=0.9)
        res[name + "_u"] = run_gpu(img, 2244, 2492, np.uint8, precision=prec, use_nis=1, sharpness=0.9)
    np.savez(out, **res)
    sys.exit(0)

ref, var = sys.argv[1], sys.argv[2]
for lib, prec, out in ((ref, 2, "/tmp/nis_ref.npz"), (var, 0, "/tmp/nis_var.npz")):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "dump", str(prec), out], env=dict(os.environ, OVRFSR_LIB=os.path.abspath(lib)))
a, b = np.load("/tmp/nis_ref.npz"), np.load("/tmp/nis_var.npz")
print("product build of %s vs strict build of %s (NVScaler 1683x1869 -> 2244x2492, sharpness 0.9)" % (var, ref))
for name in ("structured", "random"):
    d = np.abs(a[name + "_f"][..., :3].astype(np.float64) - b[name + "_f"][..., :3])
    e = np.abs(a[name + "_u"].astype(np.int16) - b[name + "_u"].astype(np.int16))
    print("%-10s float max-abs %.3e, values > 1e-3: %d of %d;  unorm8 max LSB %d, bytes differing %d" %
          (name, d.max(), int((d > 1e-3).sum()), d.size, int(e.max()), int((e != 0).sum())))
