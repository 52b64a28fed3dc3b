import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from tests import synth
from tests.util import run_gpu
for (iw,ih,ow,oh) in [(1185,1185,1580,1580),(2370,2370,3160,3160)]:
    img8 = synth.structured_u8(iw, ih, 77)
    imgh = (img8.astype(np.float32) / 255.0).astype(np.float16)
    centre, rad = O.mask_constants(ow, oh, 0.5)
    e = O.easu(imgh.astype(np.float32), ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
    ge = run_gpu(imgh, ow, oh, np.float32, precision=2, stage_mask=1, radius=0.5)
    print("easu f32 equal", np.array_equal(ge.view(np.uint32), e.view(np.uint32)), np.abs(ge-e).max())
    ge16 = run_gpu(imgh, ow, oh, np.float16, precision=2, stage_mask=1, radius=0.5)
    print("easu f16 equal", np.array_equal(ge16, e.astype(np.float16)), np.isnan(ge16.astype(np.float32)).sum())
    e16 = e.astype(np.float16).astype(np.float32)
    want = O.rcas(e16, O.rcas_con(0.9), centre, rad)
    got = run_gpu(imgh, ow, oh, np.float16, precision=2, sharpness=0.9, radius=0.5)
    d = np.abs(got.astype(np.float32)-want.astype(np.float16).astype(np.float32))
    print("pipe f16", np.array_equal(got, want.astype(np.float16)), np.nanmax(d), int((d>0).sum()), np.isnan(got.astype(np.float32)).sum(), np.isnan(want).sum())
    idx = np.argwhere(d>0)[:5]; print(idx)
    for (y,x,c) in idx: print(y,x,c, got[y,x,c], want[y,x,c], e16[y,x,c])
