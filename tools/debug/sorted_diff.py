import sys; sys.path.insert(0, '.')
import numpy as np
from tests import synth
from tests.util import run_gpu
iw, ih, ow, oh = 330, 250, 440, 333
img8 = synth.structured_u8(iw, ih, 91)
proj = (0.45, 0.5, 0.55, 0.5)
kw = dict(eye=1, precision=0, sharpness=0.7, radius=0.55, proj_centre=proj, debug_mode=0)
a = run_gpu(img8, ow, oh, np.uint8, **kw)
b = run_gpu(img8, ow, oh, np.uint8, fused=0, **kw)
c = run_gpu(img8, ow, oh, np.uint8, fused=1, **kw)
for name, x, y in (("auto vs fused0", a, b), ("auto vs fused1", a, c), ("fused0 vs fused1", b, c)):
    d = (x != y).any(axis=2)
    ys, xs = np.nonzero(d)
    print(name, "differing px:", d.sum(), "tiles:", sorted(set(zip((ys // 32).tolist(), (xs // 32).tolist())))[:20])
    if d.sum():
        print("  first:", [(int(y_), int(x_), x[y_, x_].tolist(), y[y_, x_].tolist()) for y_, x_ in list(zip(ys, xs))[:6]])
