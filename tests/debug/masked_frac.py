# tests/debug/masked_frac.py -- fraction of bytes of a radius-masked product-build frame that differ from the oracle
import sys; sys.path.insert(0, '.')
import numpy as np
from tests import synth
from tests.util import run_gpu, lsb_stats
from oracle import oracle as O
for (iw, ih, ow, oh) in ((96, 80, 128, 107), (330, 250, 440, 333), (600, 450, 800, 600)):
    for gen in (synth.structured_u8, synth.random_u8):
        img = gen(iw, ih, 11)
        for fused in (0, -1):
            want = O.fsr_pipeline_u8(img, ow, oh, sharpness=0.9, radius=0.6, proj=(0.4, 0.5, 0.6, 0.5), eye=0)
            got = run_gpu(img, ow, oh, np.uint8, eye=0, radius=0.6, sharpness=0.9, proj_centre=(0.4, 0.5, 0.6, 0.5), fused=fused)
            mx, frac = lsb_stats(got, want)
            print("%dx%d->%dx%d %-14s fused=%2d  max %d LSB, frac %.5f" % (iw, ih, ow, oh, gen.__name__, fused, mx, frac))
