import sys; sys.path.insert(0, '.')
import numpy as np, torch
import openvr_fsr_amd as A
from oracle import oracle as O
from tests import synth
from tests.test_gpu_fuzz import _outside_px
for seed in (0, 1, 6):
    rng = np.random.default_rng(3000 + seed)
    iw, ih = int(rng.integers(20, 330)), int(rng.integers(20, 330))
    s = float(rng.choice([0.5, 0.501, 0.67, 0.75, 0.77, 0.9, 0.99, rng.uniform(0.5, 1.0), 1.15]))
    ow, oh = max(8, int(iw / s)), max(8, int(ih / s))
    if s < 1: ow, oh = max(ow, iw + 1), max(oh, ih + 1)
    radius = float(rng.uniform(0.1, 0.9)); proj = tuple(float(x) for x in rng.uniform(0.3, 0.7, 4))
    eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
    img8 = [synth.structured_u8, synth.random_u8][seed % 2](iw, ih, seed)
    want = O.fsr_pipeline_u8(img8, ow, oh, sharpness=sharp, radius=radius, proj=proj, eye=eye, debug=debug)
    centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
    outside = _outside_px(ow, oh, centre, rad[1], 16, 16)
    print("seed", seed, (iw, ih, ow, oh), "s", s, "radius", radius, "debug", debug, "outside frac", outside.mean())
    for fused in (-1, 0, 1):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=sharp, radius=radius, proj_centre=proj, debug_mode=debug, precision=0, fused=fused)
        got = pp.apply(eye, torch.from_numpy(img8).cuda()).cpu().numpy(); pp.close()
        bad = (got != want).any(axis=2) & outside
        ys, xs = np.nonzero(bad)
        print("  fused", fused, "bad outside px", bad.sum(), [(int(y), int(x), got[y, x].tolist(), want[y, x].tolist()) for y, x in list(zip(ys, xs))[:4]])
        if bad.sum():
            tiles = sorted(set(zip((ys // 32).tolist(), (xs // 32).tolist())))
            # is the tile fully outside?
            print("   tiles:", [(t, bool(outside[t[0]*32:(t[0]+1)*32, t[1]*32:(t[1]+1)*32].all())) for t in tiles[:8]])
