#!/usr/bin/env python3
"""tests/debug/sampler_exposure.py -- how much of the output depends on the ONE thing nothing in the reference can pin: the D3D11 sampler.

78 % of the pixels at the reference's shipped radius 0.5 go through `Bilinear` (fsr_easu.hlsl:33-36) / `DirectCopy` (NIS_Upscale.hlsl:77-90):
one SampleLevel through the default linear-clamp sampler.  oracle/hlsl_shim.hpp restates that sampler from the D3D11 functional spec -- texel
coordinate snapped to 8 fractional bits, round to nearest -- and the C oracle, oracle/_ref and the kernels all share that reading, so no test
can see an error in it, and no D3D runtime exists here to check it against.  What CAN be stated is the exposure: this script re-runs the CPU
oracle with other sampler models (6 / 10 / 12 fractional bits, a truncating snap, exact float weights) and reports how far the final UNORM8
output moves -- per configuration, over the pixels outside the radius and (NVScaler's chroma tap goes through the same sampler) inside it.

    python tests/debug/sampler_exposure.py [--quick]        (CPU only; ~2 min at full size on 8 cores)   -> profiles/r06_sampler_exposure.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import openvr_fsr_amd as A  # noqa: E402  (host constants only: NIS coefficient banks and NISConfig -- no device is touched)
from oracle import oracle as O  # noqa: E402
from tests import natural, synth  # noqa: E402

MODELS = [("8 bits, round (D3D11 spec: the shipped reading)", 8, 0), ("6 bits, round", 6, 0), ("10 bits, round", 10, 0), ("12 bits, round", 12, 0),
          ("8 bits, truncate", 8, 1), ("exact float weights (no snap)", 0, 0)]


def set_model(bits, trunc):
    fn = O.lib().ovo_set_sampler_model
    fn.argtypes = [ctypes.c_int, ctypes.c_int]
    fn.restype = None
    fn(bits, trunc)


def outside_mask(ow, oh, centre, r2, gw, gh):
    gx = (np.arange(ow, dtype=np.uint32) // gw) * np.uint32(gw) + np.uint32(gw // 2)
    gy = (np.arange(oh, dtype=np.uint32) // gh) * np.uint32(gh) + np.uint32(gh // 2)
    c = np.asarray(centre, np.uint32)
    with np.errstate(over="ignore"):
        d1 = (c[0] - gx)[None, :] ** 2 + ((c[1] - gy) ** 2)[:, None]
        d2 = (c[2] - gx)[None, :] ** 2 + ((c[3] - gy) ** 2)[:, None]
    return ~((d1 <= np.uint32(r2)) | (d2 <= np.uint32(r2)))


def fsr_u8(img8, ow, oh, radius):
    return O.fsr_pipeline_u8(img8, ow, oh, sharpness=0.9, radius=radius)


def nis_u8(img8, ow, oh, radius):
    ih, iw = img8.shape[:2]
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(0.9, iw, ih, ow, oh)
    centre, rad = O.mask_constants(ow, oh, radius)
    return O.float_to_unorm8(O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, 0), cs, cu))


def main():
    quick = "--quick" in sys.argv
    # two shapes: BASELINE C2/C3 -- the scale is EXACTLY 3/4, every texel coordinate a multiple of 1/4 that any snap of >= 2 bits leaves alone --
    # and BASELINE C4 (x1.3: 2244x2492 -> 2916x3240), a generic ratio where the number of fractional bits matters
    shapes = [("C2/C3 shape, scale 3/4 exactly", (421, 467, 561, 623) if quick else (1683, 1869, 2244, 2492)),
              ("C4 shape, x1.3", (561, 623, 729, 810) if quick else (2244, 2492, 2916, 3240))]
    configs = [("%sr EASU+RCAS, radius 0.5", fsr_u8, 0.5, 16, 16), ("%sr NVScaler, radius 0.5", nis_u8, 0.5, 32, 24), ("%s  NVScaler, mask off (chroma tap only)", nis_u8, 2.0, 32, 24)]
    print("sampler exposure: sharpness 0.9; differences of the final UNORM8 output against the shipped sampler reading (8 bits, rounding)")
    print("%-44s %-24s %-34s %11s %9s %11s %9s" % ("configuration", "content", "sampler model", "outside: %", "max LSB", "inside: %", "max LSB"))
    for sname, (iw, ih, ow, oh) in shapes:
        print("\n== %s: %dx%d -> %dx%d" % (sname, iw, ih, ow, oh))
        contents = [("structured", synth.structured_u8(iw, ih, synth.seed_for(0, 0))), ("uniform random", synth.random_u8(iw, ih, synth.seed_for(0, 1))),
                    ("natural: rendered art", natural.tiled_u8(iw, ih, 0)), ("natural: rendered UI", natural.tiled_u8(iw, ih, 1)), ("natural: photograph", natural.tiled_u8(iw, ih, 2))]
        worst = {}
        short = "C2" if sname.startswith("C2") else "C4"
        for cfmt, fn, radius, gw, gh in configs:
            cname = cfmt % (short if fn is fsr_u8 else ("C3" if short == "C2" else "C4-NIS"))
            centre, rad = O.mask_constants(ow, oh, radius)
            outside = outside_mask(ow, oh, centre, rad[1], gw, gh)
            for tag, img8 in contents:
                set_model(8, 0)
                base = fn(img8, ow, oh, radius)
                for mname, bits, trunc in MODELS[1:]:
                    set_model(bits, trunc)
                    got = fn(img8, ow, oh, radius)
                    d = np.abs(got[..., :3].astype(np.int16) - base[..., :3].astype(np.int16))
                    o, i = d[outside], d[~outside]
                    fo = 100.0 * (o != 0).mean() if o.size else 0.0
                    fi = 100.0 * (i != 0).mean() if i.size else 0.0
                    mo, mi = (int(o.max()) if o.size else 0), (int(i.max()) if i.size else 0)
                    print("%-44s %-24s %-34s %11.4f %9d %11.4f %9d" % (cname, tag, mname, fo, mo, fi, mi), flush=True)
                    w = worst.setdefault(mname, [0.0, 0, 0.0, 0])
                    w[0], w[1], w[2], w[3] = max(w[0], fo), max(w[1], mo), max(w[2], fi), max(w[3], mi)
        set_model(8, 0)
        print("worst case at the %s over configurations and contents:" % sname)
        for mname, (fo, mo, fi, mi) in worst.items():
            print("  %-34s outside the radius: %.3f %% of the bytes change, by at most %d LSB; inside: %.4f %%, at most %d LSB" % (mname, fo, mo, fi, mi))


if __name__ == "__main__":
    main()
