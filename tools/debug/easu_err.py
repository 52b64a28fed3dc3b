#!/usr/bin/env python3
"""tools/debug/easu_err.py -- product build vs strict build of the library named by OVRFSR_LIB, on the GPU (no oracle: the
strict build is bit-identical to it, tests/test_gpu_parity*.py).  Prints, per image:
  EASU float output:   max |product - strict| in bytes (x255) and how many values exceed 2^-k of a byte, k = 6..12
  EASU UNORM8 output:  bytes that differ, max LSB
  EASU->RCAS UNORM8:   bytes that differ, max LSB, bytes off by > 1
Usage: OVRFSR_LIB=ab/k8.so python tools/debug/easu_err.py [--quick]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402
from tests.util import run_gpu  # noqa: E402

STRICT, FP32 = 2, 0
quick = "--quick" in sys.argv
cases = [("C2 structured", synth.structured_u8, 1683, 1869, 2244, 2492, 0x5EED0001),
         ("C2 random", synth.random_u8, 1683, 1869, 2244, 2492, 0x5EED0001),
         ("C2 extremes", synth.extremes_u8, 1683, 1869, 2244, 2492, 7),
         ("C4 structured", synth.structured_u8, 2244, 2492, 2916, 3240, 0x5EED0003),
         ("1.5x random", synth.random_u8, 1200, 1000, 1800, 1500, 11),
         ("2x structured", synth.structured_u8, 1000, 900, 2000, 1800, 12)]
if quick:
    cases = cases[:2]
print("lib:", os.environ.get("OVRFSR_LIB", "(default)"))
for name, gen, iw, ih, ow, oh, seed in cases:
    img = gen(iw, ih, seed)
    fs = run_gpu(img, ow, oh, np.float32, precision=STRICT, stage_mask=1)
    fp = run_gpu(img, ow, oh, np.float32, precision=FP32, stage_mask=1)
    d = np.abs(fp[..., :3].astype(np.float64) - fs[..., :3].astype(np.float64)) * 255.0
    hist = " ".join("2^-%d:%d" % (k, int((d > 2.0 ** -k).sum())) for k in range(6, 13))
    print("%-14s EASU float  max %.3e byte (%.3e unit) of %d values; > %s" % (name, d.max(), d.max() / 255.0, d.size, hist))
    us = run_gpu(img, ow, oh, np.uint8, precision=STRICT, stage_mask=1)
    up = run_gpu(img, ow, oh, np.uint8, precision=FP32, stage_mask=1)
    e = np.abs(us.astype(np.int16) - up.astype(np.int16))
    print("%-14s EASU unorm8 n_diff %d max_lsb %d" % (name, int((e != 0).sum()), int(e.max())))
    ps = run_gpu(img, ow, oh, np.uint8, precision=STRICT, sharpness=0.9)
    pp = run_gpu(img, ow, oh, np.uint8, precision=FP32, sharpness=0.9)
    e = np.abs(ps.astype(np.int16) - pp.astype(np.int16))
    print("%-14s pipeline u8 n_diff %d max_lsb %d n_gt1 %d" % (name, int((e != 0).sum()), int(e.max()), int((e > 1).sum())))
    sys.stdout.flush()
