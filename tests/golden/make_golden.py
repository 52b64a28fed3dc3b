#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the committed golden fixtures FROM THE REFERENCE ITSELF.

Needs oracle/_ref/libovrfsr_ref.so (python oracle/build_ref.py, only possible where /root/reference
exists).  Everything written here is an output of reference code:
  ref_consts.json  FsrEasuCon / FsrRcasCon / f32->f16 / NVScalerUpdateConfig / coefficient banks
                   (compiled with #define A_CPU exactly as PostProcessor.cpp:7-11 does)
  nis_vectors.npz  the same for nis/NIS_Scaler.h (NVScaler, NVSharpen, DirectCopy, radius mask)
  fsr_vectors.npz  small RGBA8 inputs and the float outputs of the reference's own fsr_easu.hlsl /
                   fsr_rcas.hlsl entry points (FsrEasuF / FsrRcasF bodies + radius mask) compiled
                   through oracle/hlsl_shim.hpp
The GPU box has no /root/reference: tests there compare against these files.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import synth  # noqa: E402

SHAPES = [(1683, 1869, 2244, 2492), (2244, 2492, 2916, 3240), (2370, 2370, 3160, 3160), (96, 80, 128, 107),
          (640, 360, 1280, 720), (1280, 720, 1920, 1080), (1000, 1000, 1299, 1299), (17, 33, 33, 64)]
SHARPNESS = [0.0, 0.25, 0.5, 0.75, 0.9, 1.0, 1.5, -0.5]


def hexs(a):
    return ["0x%08x" % int(x) for x in np.asarray(a, np.uint32).ravel()]


def consts():
    R = O.ref()
    out = {"easu_con": [], "rcas_con": [], "f32_to_f16": [], "nis_scaler": [], "nis_sharpen": []}
    for (iw, ih, ow, oh) in SHAPES:
        con = np.zeros(16, np.uint32)
        R.ref_easu_con(con.ctypes.data_as(O.u32p), iw, ih, iw, ih, ow, oh)
        out["easu_con"].append({"in": [iw, ih], "out": [ow, oh], "con": hexs(con)})
    for s in SHARPNESS:
        clamped = R.ref_clamp_f1(s, 0.0, 1.0)
        stops = float(np.float32(2.0) - np.float32(2.0) * np.float32(clamped))
        con = np.zeros(4, np.uint32)
        R.ref_rcas_con(con.ctypes.data_as(O.u32p), stops)
        out["rcas_con"].append({"sharpness": s, "stops_bits": hexs(np.array([stops], np.float32).view(np.uint32))[0],
                                "con": hexs(con)})
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        np.array([0.0, -0.0, 1.0, -1.0, 0.870550513, 65504.0, 65520.0, 65536.0, 1e-8, 6e-8, 6.1e-5, 5.96e-8, 1e10,
                  np.inf, -np.inf, np.nan, 0.333333, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25], np.float32),
        rng.uniform(-2, 2, 64).astype(np.float32),
        (rng.uniform(1, 2, 64) * 2.0 ** rng.integers(-30, 20, 64)).astype(np.float32)])
    for v in vals:
        out["f32_to_f16"].append([hexs(np.array([v], np.float32).view(np.uint32))[0], "0x%04x" % R.ref_f32_to_f16(float(v))])
    assert R.ref_nis_config_size() == 256
    for s in [0.0, 0.3, 0.5, 0.9, 1.0]:
        for (iw, ih, ow, oh) in [(1683, 1869, 2244, 2492), (2244, 2492, 2916, 3240), (960, 540, 1920, 1080), (400, 400, 1000, 1000)]:
            buf = np.zeros(64, np.uint32)
            ok = R.ref_nis_scaler_config(buf.ctypes.data, s, iw, ih, ow, oh)
            out["nis_scaler"].append({"sharpness": s, "in": [iw, ih], "out": [ow, oh], "ok": int(ok), "cfg": hexs(buf)})
        buf = np.zeros(64, np.uint32)
        ok = R.ref_nis_sharpen_config(buf.ctypes.data, s, 2244, 2492)
        out["nis_sharpen"].append({"sharpness": s, "in": [2244, 2492], "ok": int(ok), "cfg": hexs(buf)})
    sc, us = np.zeros(512, np.float32), np.zeros(512, np.float32)
    R.ref_nis_coefs(sc.ctypes.data_as(O.f32p), us.ctypes.data_as(O.f32p))
    out["nis_coef_scale"] = hexs(sc.view(np.uint32))
    out["nis_coef_usm"] = hexs(us.view(np.uint32))
    out["rmp8x8"] = []
    for lane in range(64):
        xy = np.zeros(2, np.uint32)
        R.ref_rmp8x8(lane, xy.ctypes.data_as(O.u32p))
        out["rmp8x8"].append([int(xy[0]), int(xy[1])])
    with open(os.path.join(HERE, "ref_consts.json"), "w") as f:
        json.dump(out, f, indent=0)


CASES = [
    # name, inW, inH, outW, outH, generator, seed, radius, proj, eye, sharpness, debug
    ("structured_nomask", 48, 40, 64, 53, "structured_u8", 1, 2.0, (0.5, 0.5, 0.5, 0.5), 0, 0.9, 0),
    ("random_mask", 61, 47, 80, 63, "random_u8", 2, 0.5, (0.45, 0.55, 0.6, 0.4), 1, 0.75, 1),
    ("extremes", 32, 32, 41, 41, "extremes_u8", 3, 0.7, (0.5, 0.5, 0.5, 0.5), 0, 1.0, 0),
    ("scale13", 60, 50, 78, 65, "structured_u8", 4, 2.0, (0.5, 0.5, 0.5, 0.5), 0, 0.2, 0),
]


def vectors():
    data = {}
    meta = []
    for (name, iw, ih, ow, oh, gen, seed, radius, proj, eye, sharp, dbg) in CASES:
        img8 = getattr(synth, gen)(iw, ih, seed)
        con = O.easu_con(iw, ih, ow, oh)
        centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
        easu_f = O.ref_easu(O.unorm8_to_float(img8), ow, oh, con, centre, rad)
        mid8 = O.float_to_unorm8(easu_f)
        rcon = O.rcas_con(sharp, dbg)
        rcas_f = O.ref_rcas(O.unorm8_to_float(mid8), rcon, centre, rad)
        data[name + "_in"] = img8
        data[name + "_easu"] = easu_f
        data[name + "_rcas"] = rcas_f
        data[name + "_consts"] = np.concatenate([con, rcon, centre, rad]).astype(np.uint32)
        meta.append({"name": name, "in": [iw, ih], "out": [ow, oh], "radius": radius, "proj": list(proj), "eye": eye,
                     "sharpness": sharp, "debug": dbg})
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "fsr_vectors.npz"), **data)


NIS_CASES = [
    # name, inW, inH, outW, outH, generator, seed, radius, proj, eye, sharpness, debug
    ("nis_structured", 48, 40, 64, 53, "structured_u8", 11, 2.0, (0.5, 0.5, 0.5, 0.5), 0, 0.9, 0),
    ("nis_random_mask", 61, 47, 80, 63, "random_u8", 12, 0.5, (0.45, 0.55, 0.6, 0.4), 1, 0.5, 1),
    ("nis_extremes_2x", 30, 26, 60, 52, "extremes_u8", 13, 0.8, (0.5, 0.5, 0.5, 0.5), 0, 0.2, 0),
]


def nis_vectors():
    """Outputs of the reference's own NIS_Scaler.h (NVScaler / NVSharpen + DirectCopy + mask) via oracle/_ref."""
    cs, cu = O.ref_nis_coefs()
    data, meta = {}, []
    for (name, iw, ih, ow, oh, gen, seed, radius, proj, eye, sharp, dbg) in NIS_CASES:
        img8 = getattr(synth, gen)(iw, ih, seed)
        ok, cfg = O.ref_nis_scaler_config(sharp, iw, ih, ow, oh)
        centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
        blk = O.nis_block(cfg, centre, rad, dbg)
        up = O.ref_nis_upscale(O.unorm8_to_float(img8), ow, oh, blk, cs, cu)
        # sharpen-only (renderScale == 1) on the same input
        ok2, cfg2 = O.ref_nis_scaler_config(sharp, iw, ih, iw, ih)
        c2, r2 = O.mask_constants(iw, ih, radius, proj, True, eye)
        blk2 = O.nis_block(cfg2, c2, r2, dbg)
        sh = O.ref_nis_sharpen(O.unorm8_to_float(img8), blk2, cs, cu)
        data[name + "_in"] = img8
        data[name + "_upscale"] = up
        data[name + "_sharpen"] = sh
        data[name + "_blk_upscale"] = blk
        data[name + "_blk_sharpen"] = blk2
        meta.append({"name": name, "in": [iw, ih], "out": [ow, oh], "radius": radius, "proj": list(proj), "eye": eye,
                     "sharpness": sharp, "debug": dbg, "ok": int(ok)})
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "nis_vectors.npz"), **data)


if __name__ == "__main__":
    if not O.have_ref():
        sys.exit("oracle/_ref missing: run python oracle/build_ref.py where /root/reference exists")
    consts()
    vectors()
    nis_vectors()
    print("golden fixtures written")
