#!/usr/bin/env python3
"""tools/debug/bounds_campaign.py -- drive a CHECKED build of the library (tools/build_variant.sh bounds "-DOVRFSR_BOUNDS") over every
configuration, format and odd shape the GPU tests use, and read the checked accessors' counters (openvr_fsr_amd/csrc/fsr_bounds.h).

In a checked build every LDS index, every image byte offset and every device-table index of every kernel goes through an accessor that
knows the extent of what it points into (including the DECLARED pads: allocated bytes a kernel may read and never relies on).  An access
outside is counted per kind instead of faulting; the claim is: 0 out-of-bounds accesses, and pad accesses only of the kinds the design
declares (the luma rows behind EASU's 4-rows-per-lane analysis sweep; the tap-table quads behind the last output column).

    OVRFSR_LIB=$PWD/ab/bounds.so python tools/debug/bounds_campaign.py [--full | --quick] [--seed-offset N]      (GPU; --seed-offset: only the
    random-shape generators, on draws no earlier campaign saw)

Campaign: (0) the self-test launch -- every kind of violation once, exact counts expected: a zero below means "nothing out of bounds",
not "nothing checked"; (1) the seeds of tests/test_gpu_fuzz.py (random sizes 5..330, scales 0.5..1.15, masks, projection centres, padded
row pitches) through every pipeline form; (2) ragged / tiny / minification shapes (1x1 .. 33x17, 16x16, one-tile, one-pixel-over-a-tile);
(3) every format pair; (4) shared side-by-side textures, pair_submit, batches with stride gaps; (5) BASELINE C1-C5 at full size
(2 images each; --full: 8) incl. C2r / C3r / C2s / C3s; (6) natural-content fixtures (tests/golden/natural_*.png) when present."""
import ctypes
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import openvr_fsr_amd as A  # noqa: E402
from tests import synth  # noqa: E402

DEV = torch.device("cuda")
STRICT, FP32 = 2, 0


def kinds():
    """kind names in counter order, parsed from the header (the enum is the single source)"""
    text = open(os.path.join(ROOT, "openvr_fsr_amd", "csrc", "fsr_bounds.h")).read()
    body = text[text.index("enum Kind"):text.index("K_COUNT")]
    return re.findall(r"^\s*(K_[A-Z0-9_]+)", body, re.M)


KINDS = kinds()
NK = len(KINDS)
# kinds whose declared pad a kernel is allowed to touch (fsr_bounds.h); everything else must show pad == 0 too
PAD_ALLOWED = {"K_EASU_LUM", "K_BIL_X", "K_BIL_Y"}


def checked_build():
    return hasattr(A.library(), "ovrfsr_debug_bounds")


def counters(reset=False):
    lib = A.library()
    if not checked_build():
        # not a checked build: the campaign still DRIVES every configuration -- what the host-side sanitizer build wants (OVRFSR_LIB=ab/asan_gcc.so:
        # tile lists, tap tables and argument blocks of thousands of shapes built under AddressSanitizer); every count reads 0
        return {"oob": dict.fromkeys(KINDS, 0), "pad": dict.fromkeys(KINDS, 0), "checked": dict.fromkeys(KINDS, 0), "first": None}
    fn = lib.ovrfsr_debug_bounds
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
    n = lib.ovrfsr_debug_bounds_slots()
    assert n == 3 * NK + 5, (n, NK)
    buf = (ctypes.c_ulonglong * n)()
    assert fn(buf, n, 1 if reset else 0) == 0
    v = list(buf)
    first = None
    if v[3 * NK]:
        off = v[3 * NK + 1] if v[3 * NK + 1] < 2 ** 63 else v[3 * NK + 1] - 2 ** 64
        first = {"kind": KINDS[v[3 * NK] - 1], "byte_offset": off, "access_bytes": v[3 * NK + 2], "plane_bytes": v[3 * NK + 3],
                 "block": v[3 * NK + 4] & 0xffffffff, "image": v[3 * NK + 4] >> 32}
    return {"oob": dict(zip(KINDS, v[:NK])), "pad": dict(zip(KINDS, v[NK:2 * NK])), "checked": dict(zip(KINDS, v[2 * NK:3 * NK])), "first": first}


TOTAL = {"oob": dict.fromkeys(KINDS, 0), "pad": dict.fromkeys(KINDS, 0), "checked": dict.fromkeys(KINDS, 0), "launch_groups": 0, "first": None}
FAIL = []


def collect(tag):
    torch.cuda.synchronize()
    c = counters(reset=True)
    for k in KINDS:
        TOTAL["oob"][k] += c["oob"][k]; TOTAL["pad"][k] += c["pad"][k]; TOTAL["checked"][k] += c["checked"][k]
    TOTAL["launch_groups"] += 1
    bad = {k: v for k, v in c["oob"].items() if v}
    badpad = {k: v for k, v in c["pad"].items() if v and k not in PAD_ALLOWED}
    if bad or badpad:
        FAIL.append((tag, bad, badpad, c["first"]))
        if TOTAL["first"] is None:
            TOTAL["first"] = (tag, c["first"])
        print("  !! %s: out of bounds %s, undeclared pad reads %s, first %s" % (tag, bad, badpad, c["first"]), flush=True)
    return c


def selftest():
    lib = A.library()
    if not checked_build():
        print("selftest: skipped (not a checked build: configurations are driven, nothing is counted)", flush=True)
        return True
    counters(reset=True)
    assert lib.ovrfsr_debug_bounds_selftest() == 0
    c = counters(reset=True)
    want_oob = {"K_SELFTEST": 2, "K_IMAGE_IN": 3, "K_LDS_ALLOC": 1}
    want_pad = {"K_SELFTEST": 1}
    want_chk = {"K_SELFTEST": 64 + 3, "K_IMAGE_IN": 4}
    ok = all(c["oob"][k] == want_oob.get(k, 0) for k in KINDS) and all(c["pad"][k] == want_pad.get(k, 0) for k in KINDS) and \
        all(c["checked"][k] == want_chk.get(k, 0) for k in KINDS) and c["first"] is not None
    print("selftest: oob %s pad %s checked %s first %s -> %s" % ({k: v for k, v in c["oob"].items() if v}, {k: v for k, v in c["pad"].items() if v},
                                                                  {k: v for k, v in c["checked"].items() if v}, c["first"], "ok" if ok else "MISMATCH"), flush=True)
    return ok


def to_dev(img, fmt):
    """numpy RGBA8 [H,W,4] -> device tensor of the requested input format"""
    t = torch.from_numpy(np.ascontiguousarray(img)).to(DEV)
    if fmt == "u8":
        return t, None
    if fmt == "bgra8":
        return t[..., [2, 1, 0, 3]].contiguous(), A.FORMAT_BGRA8
    if fmt == "f16":
        return (t.float() / 255.0).to(torch.float16), None
    if fmt == "f32":
        return t.float() / 255.0, None
    if fmt == "hdr16":
        x = (t.float() * (6.0 / 255.0)).to(torch.float16); x[..., 3] = 1.0
        return x, None
    if fmt in ("wild32", "wild16"):   # texels no colour image holds: NaN, +-Inf, 1e30 / 65504, denormals, negative values, zeros of both signs
        rng = np.random.default_rng(int(img.sum()) & 0xffff)
        big = 1e30 if fmt == "wild32" else 65504.0
        x = rng.standard_normal(img.shape).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 3))
        sel = rng.random(img.shape)
        for lo, v in ((0.00, 0.0), (0.04, -0.0), (0.06, 1e-41 if fmt == "wild32" else 6e-8), (0.09, big), (0.11, -big), (0.13, np.nan), (0.14, np.inf), (0.15, -np.inf)):
            x[(sel >= lo) & (sel < lo + 0.02)] = np.float32(v)
        x = torch.from_numpy(x).to(DEV)
        return (x if fmt == "wild32" else x.to(torch.float16)), None
    if fmt == "rgb10":
        v = (t[..., :3].to(torch.int64) * 1023 + 127) // 255
        return ((v[..., 0] | (v[..., 1] << 10) | (v[..., 2] << 20) | (3 << 30)) & 0xffffffff).to(torch.int64).to(torch.int32), None
    raise ValueError(fmt)


OUT_DT = {"u8": torch.uint8, "f16": torch.float16, "f32": torch.float32, "rgb10": torch.int32}


def padded(t, pad):
    """the same image inside a buffer with `pad` extra texels per row (row pitch > width * texel)"""
    if pad == 0:
        return t
    if t.dim() == 2:
        big = torch.zeros((t.shape[0], t.shape[1] + pad), dtype=t.dtype, device=DEV)
        big[:, :t.shape[1]] = t
        return big[:, :t.shape[1]]
    big = torch.zeros((t.shape[0], t.shape[1] + pad, 4), dtype=t.dtype, device=DEV)
    big[:, :t.shape[1]] = t
    return big[:, :t.shape[1]]


def one(tag, img8, ow, oh, in_fmt="u8", out_fmt="u8", pad_in=0, pad_out=0, eye=0, bounds=None, **cfg):
    tex, fmt = to_dev(img8, in_fmt)
    tex = padded(tex, pad_in)
    odt = OUT_DT[out_fmt]
    shape = (oh, ow + pad_out) if odt == torch.int32 else (oh, ow + pad_out, 4)
    big = torch.zeros(shape, dtype=odt, device=DEV)
    out = big[:, :ow]
    kw = dict(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9, radius=2.0)
    kw.update(cfg)
    try:
        pp = A.PostProcessor(**kw)
        pp.apply(eye, tex, out=out, in_format=fmt, bounds=bounds)
        torch.cuda.synchronize()
        pp.close()
    except A.OvrFsrError as e:
        if e.status in (1, 2):  # a configuration the library refuses (fused kernel's LDS fit, NIS 1x..2x ...): nothing launched
            counters(reset=True)
            return None
        raise
    return collect(tag)


def batch(tag, imgs8, ow, oh, in_fmt="u8", out_fmt="u8", gap_rows=0, pad_in=0, pad_out=0, shared=False, first_eye=0, **cfg):
    n = len(imgs8)
    texs = [to_dev(i, in_fmt)[0] for i in imgs8]
    ih, iw = texs[0].shape[:2]
    buf_in = torch.zeros((n, ih + gap_rows, iw + pad_in) + tuple(texs[0].shape[2:]), dtype=texs[0].dtype, device=DEV)
    for i, t in enumerate(texs):
        buf_in[i, :ih, :iw] = t
    odt = OUT_DT[out_fmt]
    buf_out = torch.zeros((n, oh + gap_rows, ow + pad_out, 4), dtype=odt, device=DEV)
    kw = dict(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9, radius=2.0)
    kw.update(cfg)
    try:
        pp = A.PostProcessor(**kw)
        pp.apply_batch(buf_in[:, :ih, :iw], buf_out[:, :oh, :ow], first_eye=first_eye, alternate_eyes=True, shared=shared)
        torch.cuda.synchronize()
        pp.close()
    except A.OvrFsrError as e:
        if e.status in (1, 2):
            counters(reset=True)
            return None
        raise
    return collect(tag)


FORMS = (("two-pass", dict(fused=0)), ("auto", dict(fused=-1)), ("fused", dict(fused=1)), ("easu-only", dict(stage_mask=1)))


def fsr_forms(tag, img8, ow, oh, radius, proj, eye, debug, sharp, pad_in, pad_out, precisions=(FP32, STRICT), forms=FORMS, **extra):
    for prec in precisions:
        for name, form in forms:
            kw = dict(radius=radius, proj_centre=proj, debug_mode=debug, sharpness=sharp, precision=prec)
            kw.update(form); kw.update(extra)
            one("%s %s %s" % (tag, name, "strict" if prec == STRICT else "product"), img8, ow, oh, pad_in=pad_in, pad_out=pad_out, eye=eye, **kw)


SEED_OFF = 0   # --seed-offset N: the same generators on draws no earlier campaign saw (N a multiple of 10 000 keeps the three families apart)


def fuzz_seeds():
    """the instance generators of tests/test_gpu_fuzz.py, same seeds"""
    for seed in range(24):
        rng = np.random.default_rng(1000 + seed + SEED_OFF)
        iw, ih = int(rng.integers(5, 150)), int(rng.integers(5, 150))
        s = float(rng.choice([0.5, 0.59, 0.67, 0.75, 0.77, 0.9, 0.97, rng.uniform(0.5, 1.0)]))
        ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
        radius = float(rng.choice([2.0, 2.0, rng.uniform(0.15, 1.3)]))
        proj = tuple(float(x) for x in rng.uniform(0.25, 0.75, 4))
        eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
        img8 = [synth.structured_u8, synth.random_u8, synth.extremes_u8][seed % 3](iw, ih, seed + SEED_OFF)
        pad_in, pad_out = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        fsr_forms("fuzz-fsr[%d] %dx%d->%dx%d r%.2f" % (seed, iw, ih, ow, oh, radius), img8, ow, oh, radius, proj, eye, debug, sharp, pad_in, pad_out)
        one("fuzz-fsr[%d] rcas-only" % seed, img8, iw, ih, pad_in=pad_in, pad_out=pad_out, eye=eye, radius=radius, proj_centre=proj, debug_mode=debug, stage_mask=2)
    for seed in range(12):
        rng = np.random.default_rng(2000 + seed + SEED_OFF)
        iw, ih = int(rng.integers(5, 150)), int(rng.integers(5, 150))
        s = float(rng.choice([0.5, 0.59, 0.67, 0.75, 0.77, 0.9, 0.97, rng.uniform(0.5, 1.0)]))
        ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
        radius = float(rng.choice([2.0, 2.0, rng.uniform(0.15, 1.3)]))
        proj = tuple(float(x) for x in rng.uniform(0.25, 0.75, 4))
        eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
        ow, oh = min(ow, 2 * iw), min(oh, 2 * ih)
        img8 = [synth.structured_u8, synth.random_u8, synth.extremes_u8][seed % 3](iw, ih, seed + SEED_OFF)
        pad_in = int(rng.integers(0, 9))
        for prec in (FP32, STRICT):
            for of in ("u8", "f32"):
                one("fuzz-nis[%d] %dx%d->%dx%d r%.2f %s %s" % (seed, iw, ih, ow, oh, radius, of, "strict" if prec == STRICT else "product"), img8, ow, oh,
                    out_fmt=of, pad_in=pad_in, eye=eye, use_nis=1, radius=radius, proj_centre=proj, debug_mode=debug, sharpness=sharp, precision=prec)
            one("fuzz-nis[%d] sharpen" % seed, img8, iw, ih, pad_in=pad_in, eye=eye, use_nis=1, radius=radius, proj_centre=proj, debug_mode=debug, sharpness=sharp, precision=prec)
    for seed in range(240):   # (the suite runs 16 of these; the rest widen the campaign: widths of 32 k + 1 texels with a mask need many draws)
        rng = np.random.default_rng(3000 + seed + SEED_OFF)
        iw, ih = int(rng.integers(20, 330)), int(rng.integers(20, 330))
        s = float(rng.choice([0.5, 0.501, 0.67, 0.75, 0.77, 0.9, 0.99, rng.uniform(0.5, 1.0), 1.15]))
        ow, oh = max(8, int(iw / s)), max(8, int(ih / s))
        if s < 1:
            ow, oh = max(ow, iw + 1), max(oh, ih + 1)
        radius = float(rng.uniform(0.1, 0.9))
        proj = tuple(float(x) for x in rng.uniform(0.3, 0.7, 4))
        eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
        img8 = [synth.structured_u8, synth.random_u8][seed % 2](iw, ih, seed + SEED_OFF)
        pad_in, pad_out = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        fsr_forms("fuzz-masked[%d] %dx%d->%dx%d r%.2f" % (seed, iw, ih, ow, oh, radius), img8, ow, oh, radius, proj, eye, debug, sharp, pad_in, pad_out, precisions=(FP32,))
        if ow <= 2 * iw and oh <= 2 * ih and ow >= iw and oh >= ih:
            one("fuzz-masked[%d] nis" % seed, img8, ow, oh, pad_in=pad_in, pad_out=pad_out, eye=eye, use_nis=1, radius=radius, proj_centre=proj, debug_mode=debug)


def ragged():
    shapes = [(1, 1, 2, 2), (1, 1, 1, 1), (2, 3, 3, 5), (5, 4, 7, 6), (16, 16, 21, 21), (16, 16, 32, 32), (12, 12, 16, 16), (24, 24, 32, 32), (24, 24, 33, 33),
              (31, 17, 33, 18), (33, 17, 64, 33), (47, 13, 63, 17), (13, 47, 17, 63), (96, 80, 128, 107), (100, 100, 133, 133), (64, 64, 65, 65),
              (200, 9, 267, 12), (9, 200, 12, 267), (120, 100, 100, 84), (50, 60, 25, 30), (300, 200, 301, 201), (255, 255, 340, 340), (128, 128, 256, 256),
              # output widths of 32 k + 1: the unchecked 12-byte row loads of rcas_column4 end flush with the row (the checker's own false positive of
              # round 6 -- a 3-vector is 16 bytes to sizeof, 12 to the load -- was found on 49x187 -> 97x373)
              (49, 187, 97, 373), (33, 40, 65, 80), (97, 97, 129, 129), (65, 33, 129, 65),
              # a single tile row of odd height: the discarded second pixel of the last row pair (found by fuzz seed 162312 under the checked build)
              (165, 32, 143, 27), (60, 40, 45, 29), (64, 24, 85, 31), (80, 20, 107, 27), (40, 23, 80, 31)]
    for i, (iw, ih, ow, oh) in enumerate(shapes):
        img8 = [synth.structured_u8, synth.random_u8, synth.extremes_u8][i % 3](iw, ih, 70 + i)
        for radius in (2.0, 0.84, 0.7, 0.6, 0.2):
            fsr_forms("ragged %dx%d->%dx%d r%.1f" % (iw, ih, ow, oh, radius), img8, ow, oh, radius, (0.45, 0.55, 0.52, 0.47), i & 1, (i >> 1) & 1, 0.9, i % 4, (i + 1) % 3)
            if iw <= ow <= 2 * iw and ih <= oh <= 2 * ih:
                for prec in (FP32, STRICT):
                    one("ragged nis %dx%d->%dx%d r%.1f %d" % (iw, ih, ow, oh, radius, prec), img8, ow, oh, pad_in=i % 4, pad_out=(i + 1) % 3, use_nis=1, radius=radius, precision=prec)
        for radius in (2.0, 0.5):
            for prec in (FP32, STRICT):
                one("ragged rcas-only %dx%d r%.1f %d" % (iw, ih, radius, prec), img8, iw, ih, pad_in=i % 4, radius=radius, stage_mask=2, precision=prec, debug_mode=i & 1)
                one("ragged nis-sharpen %dx%d r%.1f %d" % (iw, ih, radius, prec), img8, iw, ih, pad_in=i % 4, use_nis=1, radius=radius, precision=prec, debug_mode=i & 1)


def formats():
    iw, ih, ow, oh = 150, 110, 200, 147
    img8 = synth.structured_u8(iw, ih, 5)
    pairs = [("u8", "u8"), ("u8", "f16"), ("u8", "f32"), ("f16", "u8"), ("f16", "f16"), ("f16", "f32"), ("f32", "u8"), ("f32", "f16"), ("f32", "f32"),
             ("hdr16", "f16"), ("rgb10", "rgb10"), ("rgb10", "f32"), ("bgra8", "u8"),
             ("wild32", "f32"), ("wild32", "u8"), ("wild16", "f16"), ("wild16", "u8")]   # (non-finite / extreme texels: outside the parity contract, inside the memory-safety one)
    for fi, fo in pairs:
        for radius in (2.0, 0.5):
            for prec in (FP32, STRICT):
                forms = FORMS if not fi.startswith("rgb10") else (("two-pass", dict(fused=0)), ("easu-only", dict(stage_mask=1)))
                for name, form in forms:
                    for qi in (1, 0):
                        one("fmt %s->%s %s r%.1f q%d p%d" % (fi, fo, name, radius, qi, prec), img8, ow, oh, in_fmt=fi, out_fmt=fo, pad_in=3, pad_out=2,
                            radius=radius, precision=prec, quantize_intermediate=qi, **form)
                if not fi.startswith("rgb10"):
                    one("fmt %s->%s nis r%.1f p%d" % (fi, fo, radius, prec), img8, ow, oh, in_fmt=fi, out_fmt=fo, pad_in=3, use_nis=1, radius=radius, precision=prec)
                    one("fmt %s->%s nis-sharpen r%.1f p%d" % (fi, fo, radius, prec), img8, iw, ih, in_fmt=fi, out_fmt=fo, pad_in=3, use_nis=1, radius=radius, precision=prec)
                one("fmt %s->%s rcas-only r%.1f p%d" % (fi, fo, radius, prec), img8, iw, ih, in_fmt=fi, out_fmt=fo, pad_in=3, radius=radius, precision=prec, stage_mask=2)


def batches():
    iw, ih, ow, oh = 70, 52, 93, 69
    imgs = [synth.structured_u8(iw, ih, 40 + i) for i in range(5)]
    proj = (0.45, 0.5, 0.55, 0.5)
    for prec in (FP32, STRICT):
        for radius in (2.0, 0.8, 0.3):
            for name, form in FORMS:
                batch("batch5 gaps %s r%.1f p%d" % (name, radius, prec), imgs, ow, oh, gap_rows=3, pad_in=2, pad_out=5, first_eye=1, radius=radius, proj_centre=proj, precision=prec, **form)
            batch("batch5 nis r%.1f p%d" % (radius, prec), imgs, ow, oh, gap_rows=3, pad_in=2, pad_out=5, use_nis=1, radius=radius, proj_centre=proj, precision=prec)
            batch("batch5 half r%.1f p%d" % (radius, prec), imgs, ow, oh, in_fmt="f16", out_fmt="f16", gap_rows=1, radius=radius, proj_centre=proj, precision=prec)
    # shared side-by-side textures: both eyes in one image, two mask centres
    sbs = [synth.structured_u8(2 * iw, ih, 60 + i) for i in range(3)]
    for radius in (2.0, 0.5):
        for name, form in FORMS:
            batch("shared sbs %s r%.1f" % (name, radius), sbs, 2 * ow, oh, shared=True, radius=radius, proj_centre=proj, **form)
        batch("shared sbs nis r%.1f" % radius, sbs, 2 * ow, oh, shared=True, use_nis=1, radius=radius, proj_centre=proj)
    # pair_submit: LEFT recorded, RIGHT launches both; either order
    for radius in (2.0, 0.5):
        for order in ((0, 1), (1, 0)):
            pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9, radius=radius, pair_submit=1)
            both = torch.stack([torch.from_numpy(imgs[0]).to(DEV), torch.from_numpy(imgs[1]).to(DEV)])
            outs = torch.zeros((2, oh, ow, 4), dtype=torch.uint8, device=DEV)
            for frame in range(3):
                for eye in order:
                    pp.apply(eye, both[eye], out=outs[eye])
            torch.cuda.synchronize()
            pp.close()
            collect("pair_submit r%.1f order %s" % (radius, order))


def baseline(n):
    C2 = (1683, 1869, 2244, 2492)
    C4 = (2244, 2492, 2916, 3240)
    C5 = (2370, 2370, 3160, 3160)

    def imgs(w, h, k):
        return [synth.structured_u8(w, h, synth.seed_for(i // 2, i & 1)) if i % 2 == 0 else synth.random_u8(w, h, synth.seed_for(i // 2, i & 1)) for i in range(k)]

    i2 = imgs(C2[0], C2[1], n)
    batch("C1 EASU-only fp32 out", i2[:2], C2[2], C2[3], out_fmt="f32", stage_mask=1)
    batch("C1 EASU-only strict", i2[:2], C2[2], C2[3], out_fmt="f32", stage_mask=1, precision=STRICT)
    batch("C2 EASU+RCAS", i2, C2[2], C2[3])
    batch("C2 fused", i2[:2], C2[2], C2[3], fused=1)
    batch("C2 strict", i2[:2], C2[2], C2[3], precision=STRICT)
    batch("C2r radius 0.5 (sorted)", i2, C2[2], C2[3], radius=0.5)
    batch("C2r radius 0.5 fused", i2[:2], C2[2], C2[3], radius=0.5, fused=1)
    batch("C2r radius 0.5 two-pass", i2[:2], C2[2], C2[3], radius=0.5, fused=0)
    one("C2 single apply", i2[0], C2[2], C2[3])
    one("C2r single apply", i2[0], C2[2], C2[3], radius=0.5)
    batch("C3 NVScaler", i2, C2[2], C2[3], use_nis=1)
    batch("C3 NVScaler strict", i2[:2], C2[2], C2[3], use_nis=1, precision=STRICT)
    batch("C3r NVScaler radius 0.5", i2, C2[2], C2[3], use_nis=1, radius=0.5)
    i2o = imgs(C2[2], C2[3], 2)
    batch("C2s RCAS only", i2o, C2[2], C2[3], stage_mask=2)
    batch("C2s RCAS only radius 0.5", i2o, C2[2], C2[3], stage_mask=2, radius=0.5)
    batch("C3s NVSharpen", i2o, C2[2], C2[3], use_nis=1)
    sbs = [np.concatenate([i2[0], i2[1]], axis=1)]
    batch("C2sbs shared", sbs, 2 * C2[2], C2[3], shared=True)
    batch("C2sbsr shared radius 0.5", sbs, 2 * C2[2], C2[3], shared=True, radius=0.5)
    i4 = imgs(C4[0], C4[1], 2)
    batch("C4 EASU+RCAS x1.3", i4, C4[2], C4[3])
    batch("C4 NVScaler x1.3", i4, C4[2], C4[3], use_nis=1)
    i5 = imgs(C5[0], C5[1], 2)
    batch("C5 masked half", i5, C5[2], C5[3], in_fmt="f16", out_fmt="f16", radius=0.5)
    batch("C5 masked half HDR x6", i5, C5[2], C5[3], in_fmt="hdr16", out_fmt="f16", radius=0.5)
    batch("C5 unmasked half two-pass", i5, C5[2], C5[3], in_fmt="f16", out_fmt="f16", radius=2.0, fused=0)
    batch("C5 masked half strict", i5[:1], C5[2], C5[3], in_fmt="f16", out_fmt="f16", radius=0.5, precision=STRICT)
    batch("C5 masked RGBA8", i5, C5[2], C5[3], radius=0.5)


def natural():
    gold = os.path.join(ROOT, "tests", "golden")
    names = sorted(f for f in os.listdir(gold) if f.startswith("natural_") and f.endswith(".npz"))
    for f in names:
        z = np.load(os.path.join(gold, f))
        img8 = z["rgba8"]
        ih, iw = img8.shape[:2]
        for (ow, oh) in ((iw * 4 // 3, ih * 4 // 3), (int(iw * 1.3), int(ih * 1.3))):
            for radius in (2.0, 0.5):
                fsr_forms("natural %s %dx%d->%dx%d r%.1f" % (f, iw, ih, ow, oh, radius), img8, ow, oh, radius, (0.5, 0.5, 0.5, 0.5), 0, 0, 0.9, 0, 0)
                one("natural %s nis r%.1f" % (f, radius), img8, ow, oh, use_nis=1, radius=radius)
                one("natural %s half r%.1f" % (f, radius), img8, ow, oh, in_fmt="hdr16", out_fmt="f16", radius=radius)
    return len(names)


def main():
    global SEED_OFF
    full = "--full" in sys.argv
    quick = "--quick" in sys.argv
    if "--seed-offset" in sys.argv:   # fresh random shapes only: the other sections are deterministic
        SEED_OFF = int(sys.argv[sys.argv.index("--seed-offset") + 1], 0)
    t0 = time.time()
    ok = selftest()
    sections = [("fuzz seeds", fuzz_seeds), ("ragged shapes", ragged), ("formats", formats), ("batches / shared / pair", batches)]
    if not quick:
        sections.append(("BASELINE C1-C5 full size", lambda: baseline(8 if full else 2)))
    sections.append(("natural content", natural))
    if SEED_OFF:
        sections = [("fuzz seeds + %d" % SEED_OFF, fuzz_seeds)]
    for name, fn in sections:
        before = TOTAL["launch_groups"]
        t1 = time.time()
        fn()
        print("%-28s %5d configurations   [%.0f s]" % (name, TOTAL["launch_groups"] - before, time.time() - t1), flush=True)
    print("\nkind                 checked accesses   out of bounds   declared-pad accesses")
    for k in KINDS:
        if TOTAL["checked"][k] or TOTAL["oob"][k] or TOTAL["pad"][k]:
            print("%-20s %16d %15d %15d%s" % (k, TOTAL["checked"][k], TOTAL["oob"][k], TOTAL["pad"][k], "" if not TOTAL["pad"][k] else ("   (declared)" if k in PAD_ALLOWED else "   UNDECLARED")))
    n_oob = sum(TOTAL["oob"].values())
    n_badpad = sum(v for k, v in TOTAL["pad"].items() if k not in PAD_ALLOWED)
    print("TOTAL configurations %d, checked accesses %d, OUT OF BOUNDS %d, undeclared pad accesses %d, declared pad accesses %d, selftest %s   [%.0f s]"
          % (TOTAL["launch_groups"], sum(TOTAL["checked"].values()), n_oob, n_badpad, sum(v for k, v in TOTAL["pad"].items() if k in PAD_ALLOWED),
             "ok" if ok else "MISMATCH", time.time() - t0))
    for f in FAIL[:40]:
        print("FAIL", f)
    sys.exit(0 if (ok and n_oob == 0 and n_badpad == 0) else 1)


if __name__ == "__main__":
    main()
