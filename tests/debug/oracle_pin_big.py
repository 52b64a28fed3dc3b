"""tests/debug/oracle_pin_big.py -- the oracle against oracle/_ref at FULL BASELINE sizes and at the largest coordinates (CPU, ~3 minutes): C2, C4, the C5
shape, 16384-wide and 16384-high strips, a x1.0001 and a x2 ratio; natural content unmasked, uniform noise with the mask (right eye, shifted centres);
EASU, RCAS behind the UNORM8 intermediate, NVScaler -- bit for bit.  The random campaign (oracle_pin_campaign.py) stays below 260 texels a side; this is where
the restated sampler (coordinates snapped to 8 sub-texel bits after an fp32 multiply by up to 16384) would show a size-dependent slip."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'./tests/debug')
import numpy as np
from oracle import oracle as O
from tests import synth, natural
import oracle_pin_campaign as P
cs, cu = O.ref_nis_coefs()
cases = [("C2", 1683, 1869, 2244, 2492), ("C4", 2244, 2492, 2916, 3240), ("C5 shape", 2370, 2370, 3160, 3160), ("strip 16384 wide", 12288, 48, 16384, 64), ("strip 16384 high", 48, 12288, 64, 16384),
         ("x1.0001", 4000, 300, 4001, 301), ("x2 exactly", 1500, 1100, 3000, 2200)]
for name, iw, ih, ow, oh in cases:
    t0 = time.time()
    for radius in (2.0, 0.5):
        img = O.unorm8_to_float(natural.tiled_u8(iw, ih, 3) if radius == 2.0 else synth.random_u8(iw, ih, 9))
        centre, rad = O.mask_constants(ow, oh, radius, (0.45, 0.5, 0.55, 0.5), True, 1)
        con = O.easu_con(iw, ih, ow, oh)
        a = O.easu(img, ow, oh, con, centre, rad)
        e1 = P.same_bits(a, O.ref_easu(img, ow, oh, con, centre, rad))
        q = O.unorm8_to_float(O.float_to_unorm8(a)); rcon = O.rcas_con(0.9, 0)
        e2 = P.same_bits(O.rcas(q, rcon, centre, rad), O.ref_rcas(q, rcon, centre, rad))
        e3 = None
        if ow <= 2 * iw and oh <= 2 * ih:
            ok, cfg = O.ref_nis_scaler_config(0.9, iw, ih, ow, oh)
            blk = O.nis_block(cfg, centre, rad, 0)
            e3 = P.same_bits(O.nis_upscale(img, ow, oh, blk, cs, cu), O.ref_nis_upscale(img, ow, oh, blk, cs, cu))
        print("%-18s %5dx%-5d -> %5dx%-5d radius %.1f: EASU %s  RCAS %s  NVScaler %s   [%.0f s]" % (name, iw, ih, ow, oh, radius, e1, e2, e3, time.time() - t0), flush=True)
