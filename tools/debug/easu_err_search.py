#!/usr/bin/env python3
"""tools/debug/easu_err_search.py -- ADVERSARIAL search for the largest distance between the product build's re-associated EASU
resolve and the reference-order evaluation (the strict build, bit-identical to the oracle), in bytes of the UNORM8 domain.

Why: the near-tie guard of the product kernels re-resolves, in reference order, every pixel whose fast result lies within
2^-9 byte (1.95e-3) of a rounding boundary; the UNORM8 EASU output is bit-identical to the oracle as long as
|fast - reference| stays below that band.  profiles/r03_easu_err.txt measured the distance on whole images (6e-4 byte at worst);
this tool LOOKS for a worse input: a (1 + lambda) evolution over 8x8-texel patches.  Every generation is one image made of a
grid of mutants of the current champion patch (single-texel edits, extremes, steps, fresh random / two-level / ramp patches);
the EASU float output of both builds is compared on the pixels whose 12-tap footprint and 4 analysis '+' patterns lie inside
their own patch, and the patch with the largest distance becomes the next champion.

NIS=1 in the environment searches NVScaler (use_nis, sharpness 0.9; 12x12 patches for its 6x6 support + interpolated edge map) instead;
there is no guard on that path: its contract is <= 1 LSB, and the distance shows how much of it re-association may use.
HALF=1 [HSCALE=s] searches EASU on RGBA16F texels half(b/255*s) against the HALF guard's relative band (outputs >= 0.5).
PIPE=1 searches the EASU -> RCAS pipeline's float output (10x10 patches): with the intermediate bit-identical, RCAS's own distance.

AUDIT=1 (with OVRFSR_LIB pointing at an audit build, tools/build_variant.sh audit "-DOVRFSR_TIE_AUDIT"): every generation's image is ALSO run
through the guarded store path (EASU-only UNORM8 output; RGBA16F output with the half guard on from 0.5 for HALF=1), where the audit build
re-resolves every pixel in reference order and counts FLIPS -- unlisted pixels whose stored value differs from the strict build's.  The
search's objective stays the distance: a flip needs |product - strict| > band at a pixel the guard did not list, so driving the distance up
IS the search for a flip; the audit says whether any candidate on the way produced one.

Usage: [NIS=1 | PIPE=1 | HALF=1 [HSCALE=s]] [AUDIT=1] python tools/debug/easu_err_search.py [generations=400] [seed=1] [scales=0,1,2,3]      (GPU; OVRFSR_LIB selects the library)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STRICT, FP32 = 2, 0
NIS = "pipe" if os.environ.get("PIPE", "0") == "1" else "half" if os.environ.get("HALF", "0") == "1" else os.environ.get("NIS", "0") == "1"
HSCALE = float(os.environ.get("HSCALE", "1.0"))   # HALF=1: texel = half(b / 255 * HSCALE)
AUDIT = os.environ.get("AUDIT", "0") == "1"
if AUDIT and NIS == "half":
    os.environ.setdefault("OVRFSR_TIE_HALF_MIN", "0.5")


def audit_counters(reset=False):
    """the audit build's device counters (see g_ovrfsr_tie_audit, fsr_kernels.hip); None with a product library"""
    import ctypes
    import openvr_fsr_amd as A
    fn = getattr(A.library(), "ovrfsr_debug_tie_audit", None)
    if fn is None:
        return None
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 6)()
    if fn(buf, 1 if reset else 0) != 0:
        return None
    f = lambda bits: float(np.uint32(bits & 0xffffffff).view(np.float32))  # noqa: E731
    return {"audited": buf[0], "listed": buf[1], "flips": buf[2], "small_half_diffs": buf[3], "max_dist_bytes": f(buf[4]), "max_dist_half_spacings": f(buf[5])}
P = 8   # patch edge in texels (set by search(): 8 for EASU, 12 for NVScaler; 30 / 20 patches per image edge -> 240 x 240 texels,
        # divisible by 3 and 10: exact output sizes at every scale of SCALES)
BAND = 2.0 ** -9


def fresh(rng):
    """a random patch of one of several families (bytes, [P,P,3])"""
    k = rng.integers(0, 6)
    if k == 0:
        return rng.integers(0, 256, (P, P, 3)).astype(np.uint8)
    if k == 1:   # two levels split by a random line, per channel levels
        a, b = rng.integers(0, 256, 3), rng.integers(0, 256, 3)
        th = rng.uniform(0, np.pi)
        yy, xx = np.mgrid[0:P, 0:P]
        side = (np.cos(th) * (xx - P / 2 + rng.uniform(-1, 1)) + np.sin(th) * (yy - P / 2 + rng.uniform(-1, 1))) > 0
        return np.where(side[..., None], a, b).astype(np.uint8)
    if k == 2:   # extremes only
        return (rng.integers(0, 2, (P, P, 3)) * 255).astype(np.uint8)
    if k == 3:   # ramp + noise
        yy, xx = np.mgrid[0:P, 0:P]
        g = rng.uniform(-30, 30, 2)
        base = rng.integers(0, 256, 3)
        v = base[None, None, :] + (g[0] * xx + g[1] * yy)[..., None] + rng.integers(-4, 5, (P, P, 3))
        return np.clip(v, 0, 255).astype(np.uint8)
    if k == 4:   # near-constant with one or two outliers (ring texels much larger than the inner four)
        v = np.full((P, P, 3), rng.integers(0, 40), np.int64) + rng.integers(0, 3, (P, P, 3))
        for _ in range(int(rng.integers(1, 6))):
            v[rng.integers(0, P), rng.integers(0, P), :] = rng.integers(200, 256)
        return np.clip(v, 0, 255).astype(np.uint8)
    lv = rng.integers(0, 256, (2, 3))   # checkerboard-ish, two levels
    yy, xx = np.mgrid[0:P, 0:P]
    return lv[(xx // rng.integers(1, 3) + yy // rng.integers(1, 3)) % 2].astype(np.uint8)


def mutate(rng, p):
    q = p.copy()
    k = rng.integers(0, 8)
    n = int(rng.choice([1, 1, 2, 3, 6]))
    for _ in range(n):
        y, x = rng.integers(0, P, 2)
        if k == 0:
            q[y, x, rng.integers(0, 3)] = rng.integers(0, 256)
        elif k == 1:
            c = rng.integers(0, 3)
            q[y, x, c] = np.clip(int(q[y, x, c]) + int(rng.integers(-8, 9)), 0, 255)
        elif k == 2:
            q[y, x, :] = rng.integers(0, 256, 3)
        elif k == 3:
            q[y, x, :] = rng.choice([0, 255])
        elif k == 4:
            q[y, x, :] = q[(y + rng.integers(-1, 2)) % P, (x + rng.integers(-1, 2)) % P, :]
        elif k == 5:
            q[y, x, :] = np.clip(q[y, x, :].astype(np.int64) + rng.integers(-2, 3), 0, 255)
        elif k == 6:
            q[y, :, rng.integers(0, 3)] = rng.integers(0, 256)
        else:
            q[:, x, rng.integers(0, 3)] = rng.integers(0, 256)
    return q


SCALES = (("scale 3/4 (C2)", 4, 3), ("scale 0.77 (C4 shape)", 13, 10), ("scale 1/2", 2, 1), ("scale 2/3", 3, 2),
          ("scale 1 (sharpen only)", 1, 1))   # index 4: PIPE=1 -> RCAS alone, NIS=1 -> NVSharpen (not part of the default sweep)


def search(scale_index, gens, seed, nis=False, verbose=True):
    """(1 + lambda) evolution at SCALES[scale_index]; returns (worst distance in bytes, champion patch)."""
    from tests.util import run_gpu
    half = nis == "half"
    if half:            # EASU of RGBA16F texels, float output; distance in units of the HALF guard's band at the output value (>= 0.5 only)
        kw, (p, g_, lo, hi) = dict(stage_mask=1), (8, 30, 2, 4)
    elif nis == "pipe":   # EASU -> RCAS, float output (the intermediate keeps its format rounding): RCAS's own re-association distance
        kw, (p, g_, lo, hi) = dict(sharpness=0.9), (10, 24, 3, 6)
    elif nis:
        kw, (p, g_, lo, hi) = dict(use_nis=1, sharpness=0.9), (12, 20, 4, 7)
    else:
        kw, (p, g_, lo, hi) = dict(stage_mask=1), (8, 30, 2, 4)
    global P
    P = p   # fresh() / mutate() read the patch edge
    name, num, den = SCALES[scale_index]
    rng = np.random.default_rng(seed)
    iw = ih = p * g_
    ow = oh = iw * num // den
    # output pixels whose footprint stays inside their own patch: base texel (floor of the source position) at patch-local lo..hi
    pos = (np.arange(ow) + 0.5) * (iw / ow) - 0.5
    f = np.floor(pos).astype(np.int64)
    ok1 = ((f % p) >= lo) & ((f % p) <= hi)
    pid1 = f // p
    ok = ok1[:, None] & ok1[None, :]
    pid = pid1[:, None] * g_ + pid1[None, :]
    champ, best = fresh(rng), 0.0
    t0 = time.time()
    for g in range(gens):
        patches = [champ]
        for i in range(g_ * g_ - 1):
            r = rng.uniform()
            patches.append(fresh(rng) if r < 0.15 or best == 0.0 else mutate(rng, champ if r < 0.9 else patches[int(rng.integers(0, len(patches)))]))
        img = np.empty((ih, iw, 4), np.uint8)
        img[..., 3] = 255
        arr = np.stack(patches).reshape(g_, g_, p, p, 3)
        img[..., :3] = arr.transpose(0, 2, 1, 3, 4).reshape(ih, iw, 3)
        src = (img.astype(np.float32) * np.float32(HSCALE / 255.0)).astype(np.float16) if half else img
        fs = run_gpu(src, ow, oh, np.float32, precision=STRICT, **kw)
        fp = run_gpu(src, ow, oh, np.float32, precision=FP32, **kw)
        if AUDIT and (half or not nis):   # the guarded store path of the same candidates (counted on the device by an audit build)
            run_gpu(src, ow, oh, np.float16 if half else np.uint8, precision=FP32, **kw)
        if half:
            a, b = fp[..., :3].astype(np.float64), fs[..., :3].astype(np.float64)
            band = 2.0 ** (np.floor(np.log2(np.maximum(b, 2.0 ** -14))) - 10 - 6)   # 2^-6 of the half spacing of the reference value
            d = (np.where(b >= 0.5, np.abs(a - b) / band, 0.0)).max(axis=2) * BAND  # in "bands", scaled so that the printout's ratio is right
        else:
            d = np.abs(fp[..., :3].astype(np.float64) - fs[..., :3].astype(np.float64)).max(axis=2) * 255.0
        d = np.where(ok, d, 0.0)
        per = np.zeros(g_ * g_)
        np.maximum.at(per, pid.ravel(), d.ravel())
        j = int(per.argmax())
        if per[j] > best:
            best, champ = float(per[j]), patches[j]
        if verbose and (g in (0, 9, 49, 99, 199) or (g + 1) % 500 == 0 or g == gens - 1):
            print("  %-22s generation %4d: worst distance found %.3e byte (%.2f of the band), %.0f s" % (name, g + 1, best, best / BAND, time.time() - t0), flush=True)
    if verbose and half:
        print("  %-22s (HALF: distances are in units of the half guard's band at the output value, x 2^-9 for the printout; outputs >= 0.5 only)" % name)
        return best, champ
    if verbose:
        print("  %-22s champion patch (R plane): %s" % (name, champ[..., 0].tolist()))
        # how the champion fares in the UNORM8 output (the guard's job): both builds, bytes that differ
        img = np.empty((p * 4, p * 4, 4), np.uint8)
        img[..., 3] = 255
        img[..., :3] = np.tile(champ, (4, 4, 1))
        o = p * 4 * num // den
        us = run_gpu(img, o, o, np.uint8, precision=STRICT, **kw)
        up = run_gpu(img, o, o, np.uint8, precision=FP32, **kw)
        print("  %-22s champion tiled 4x4, UNORM8 output: %d bytes differ between product and strict builds (max %d LSB)"
              % (name, int((us != up).sum()), int(np.abs(us.astype(np.int16) - up.astype(np.int16)).max())))
    return best, champ


def main():
    gens = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pick = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(4)
    print("lib:", os.environ.get("OVRFSR_LIB", "(default)"), "EASU+RCAS" if NIS == "pipe" else "EASU of RGBA16F texels (x%g)" % HSCALE if NIS == "half" else "NVScaler" if NIS else "EASU", " %d generations, seed %d, band 2^-9 = %.3e byte" % (gens, seed, BAND))
    if AUDIT:
        assert audit_counters(reset=True) is not None, "AUDIT=1 needs an audit build (OVRFSR_LIB=ab/audit.so)"
    for i in pick:
        search(i, gens, seed, nis=NIS)
    if AUDIT:
        c = audit_counters()
        print("AUDIT: %d pixels of the search's candidates audited on their guarded store path, %d listed, FLIPS %d, small-channel half differences %d, "
              "max distance %.3e byte / %.3e half spacings" % (c["audited"], c["listed"], c["flips"], c["small_half_diffs"], c["max_dist_bytes"], c["max_dist_half_spacings"]))
        sys.exit(1 if c["flips"] else 0)


if __name__ == "__main__":
    main()
