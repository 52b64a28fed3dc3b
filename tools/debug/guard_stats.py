#!/usr/bin/env python3
"""tools/debug/guard_stats.py -- which branch of the near-tie pass does a content distribution take?

The product EASU kernel lists, per 32x32 tile, the pixels with a channel within 2^-9 byte of a UNORM8 rounding boundary and
re-resolves them in the reference's operator order: four lanes per pixel when the tile lists <= 32 pixels
(fsr_kernels.inc, `total <= 32u`), one lane per pixel otherwise.  This tool evaluates the same test on the product build's
FLOAT EASU output (x255; the float store is the value the test sees, up to its 1/255 scaling rounding) and prints the
list-length distribution per tile for structured and uniform-random C2 content.  Run on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402
from tests.util import run_gpu  # noqa: E402

K = 9
iw, ih, ow, oh = 1683, 1869, 2244, 2492
for name, gen in (("structured", synth.structured_u8), ("uniform random", synth.random_u8), ("extremes 0/255", synth.extremes_u8)):
    img = gen(iw, ih, 0x5EED0000)
    f = run_gpu(img, ow, oh, np.float32, precision=0, stage_mask=1)[..., :3].astype(np.float64) * 255.0
    d = np.abs(f - (np.floor(f) + 0.5))
    tie = (d < 2.0 ** -K).any(axis=2)
    th, tw = (oh + 31) // 32, (ow + 31) // 32
    pad = np.zeros((th * 32, tw * 32), bool)
    pad[:oh, :ow] = tie
    per_tile = pad.reshape(th, 32, tw, 32).sum(axis=(1, 3)).ravel()
    hist = np.bincount(np.minimum(per_tile, 64), minlength=65)
    print("%-16s near-tie pixels %.3f %% of %d; per 32x32 tile: mean %.1f, median %d, max %d; tiles with 0: %.1f %%, 1..16: %.1f %%, "
          "17..32: %.1f %%, > 32 (one-lane dense branch): %.1f %%"
          % (name, 100.0 * tie.mean(), tie.size, per_tile.mean(), int(np.median(per_tile)), int(per_tile.max()),
             100.0 * hist[0] / per_tile.size, 100.0 * hist[1:17].sum() / per_tile.size, 100.0 * hist[17:33].sum() / per_tile.size,
             100.0 * hist[33:].sum() / per_tile.size))
    sys.stdout.flush()
