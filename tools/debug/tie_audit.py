#!/usr/bin/env python3
"""tools/debug/tie_audit.py -- drive an AUDIT build of the library (tools/build_variant.sh audit "-DOVRFSR_TIE_AUDIT") over a content campaign.

The product build claims that its quantised EASU stores (the UNORM8 / half intermediate, an EASU-only UNORM8 output) are the strict build's:
pixels whose re-associated result lies within the near-tie band of a rounding boundary are re-resolved in the reference's operator order, all
others are stored as they are.  The audit build re-resolves EVERY pixel in reference order inside the same kernels and counts, on the device,
the pixels that were NOT listed by the guard and whose stored value differs from the strict build's ("flips"): the claim is flips == 0.
It also records the largest |product - strict| it met (bytes of the UNORM8 domain; half spacings for half stores): the band is 2^-9 byte /
2^-6 spacing.

    OVRFSR_LIB=$PWD/ab/audit.so python tools/debug/tie_audit.py [scale=1.0] [seed_offset=0]      (GPU; `scale` multiplies the images per
    configuration, `seed_offset` moves every generator seed: a further campaign on content no earlier one saw)

Content: natural images (round 6: tests/golden/natural_*.npz, mirror-tiled), the bench's structured and uniform-random generators, 0 / 255-heavy images, and a mosaic of 8x8-texel patches of the families the
adversarial search (tools/debug/easu_err_search.py) mutates -- two-level edges at random angles, ramps with noise, near-constant patches with
outliers, extremes --; for the half pipelines the same content as half(b / 255 * s), s in {1, 6, 40} (unit range and HDR).
Shapes: BASELINE C2 (x4/3), C4 (x1.3), C5 (x4/3, masked and unmasked, half), and odd ratios (x1.7, x1.11, x2)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import openvr_fsr_amd as A  # noqa: E402

DEV = torch.device("cuda")


def counters(reset=False):
    lib = A.library()
    fn = lib.ovrfsr_debug_tie_audit      # AttributeError: not an audit build
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 6)()
    assert fn(buf, 1 if reset else 0) == 0
    v = list(buf)
    f = lambda bits: float(np.uint32(bits & 0xffffffff).view(np.float32))  # noqa: E731
    return {"audited": v[0], "listed": v[1], "flips": v[2], "small_half_diffs": v[3], "max_dist_bytes": f(v[4]), "max_dist_half_spacings": f(v[5])}


def extremes_batch(n, w, h, seed):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    vals = torch.tensor([0, 0, 255, 255, 1, 254, 128], dtype=torch.uint8, device=DEV)
    out = vals[torch.randint(0, 7, (n, h, w, 4), generator=g, device=DEV)]
    out[..., 3] = 255
    return out


def mosaic_batch(n, w, h, seed, P=8):
    """every 8x8-texel patch from one of the adversarial search's families, chosen at random"""
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    ph, pw = (h + P - 1) // P, (w + P - 1) // P
    r = lambda *shape: torch.rand(shape, generator=g, device=DEV)  # noqa: E731
    ri = lambda lo, hi, *shape: torch.randint(lo, hi, shape, generator=g, device=DEV)  # noqa: E731
    yy = (torch.arange(ph * P, device=DEV) % P).float()[None, :, None, None] - P / 2
    xx = (torch.arange(pw * P, device=DEV) % P).float()[None, None, :, None] - P / 2
    up = lambda t: t.repeat_interleave(P, 1).repeat_interleave(P, 2)  # noqa: E731  -- per-patch value -> per-texel
    fam = up(ri(0, 5, n, ph, pw, 1))
    # two levels split by a random line
    th = up(r(n, ph, pw, 1) * 3.14159)
    side = (torch.cos(th) * (xx + up(r(n, ph, pw, 1) * 2 - 1)) + torch.sin(th) * (yy + up(r(n, ph, pw, 1) * 2 - 1))) > 0
    two = torch.where(side, up(ri(0, 256, n, ph, pw, 3)).float(), up(ri(0, 256, n, ph, pw, 3)).float())
    # ramp + noise
    ramp = up(ri(0, 256, n, ph, pw, 3)).float() + up(r(n, ph, pw, 1) * 60 - 30) * xx + up(r(n, ph, pw, 1) * 60 - 30) * yy + ri(-4, 5, n, ph * P, pw * P, 3).float()
    # near-constant with outliers
    flat = up(ri(0, 40, n, ph, pw, 1)).float() + ri(0, 3, n, ph * P, pw * P, 3).float()
    flat = torch.where(r(n, ph * P, pw * P, 1) < 0.05, ri(200, 256, n, ph * P, pw * P, 1).float().expand(-1, -1, -1, 3), flat)
    ext = (ri(0, 2, n, ph * P, pw * P, 3) * 255).float()
    rnd = ri(0, 256, n, ph * P, pw * P, 3).float()
    img = torch.where(fam == 0, two, torch.where(fam == 1, ramp, torch.where(fam == 2, flat, torch.where(fam == 3, ext, rnd))))
    out = torch.empty((n, h, w, 4), dtype=torch.uint8, device=DEV)
    out[..., :3] = img[:, :h, :w].clamp_(0, 255).to(torch.uint8)
    out[..., 3] = 255
    return out


CONTENT = {
    # round 6: NATURAL content -- the fixtures of tests/golden/natural_*.npz (rendered game art, a rendered UI with text, a photograph), mirror-tiled
    "natural": lambda n, w, h, s: bench.natural_batch(n, w, h, torch.uint8, DEV, s),
    "structured": lambda n, w, h, s: bench.synth_batch(n, w, h, torch.uint8, DEV, s),
    "random": lambda n, w, h, s: bench.random_batch(n, w, h, torch.uint8, DEV, s),
    "extremes": extremes_batch,
    "mosaic": mosaic_batch,
}


def run(tag, inW, inH, outW, outH, content, n, seed, half_scale=None, **cfg):
    texs = CONTENT[content](n, inW, inH, seed)
    if half_scale is not None:
        texs = (texs.float() * (half_scale / 255.0)).to(torch.float16)
        texs[..., 3] = 1.0
    outs = torch.empty((n, outH, outW, 4), dtype=texs.dtype, device=DEV)
    pp = A.PostProcessor(fsr_enabled=1, out_width=outW, out_height=outH, sharpness=0.9, quantize_intermediate=1, **cfg)
    counters(reset=True)
    pp.apply_batch(texs, outs, first_eye=A.EYE_LEFT, alternate_eyes=True)
    torch.cuda.synchronize()
    c = counters()
    pp.close()
    print("%-58s %-10s n=%3d  audited %12d  listed %9d (%.2f %%)  FLIPS %d  small-channel half diffs %d  max dist %.3e byte  %.3e half spacings"
          % (tag, content + ("" if half_scale is None else " x%g" % half_scale), n, c["audited"], c["listed"], 100.0 * c["listed"] / max(1, c["audited"]),
             c["flips"], c["small_half_diffs"], c["max_dist_bytes"], c["max_dist_half_spacings"]), flush=True)
    return c


def main():
    k = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    N = lambda n: max(2, int(round(n * k)) & ~1)  # noqa: E731
    total = {"audited": 0, "listed": 0, "flips": 0, "small_half_diffs": 0, "max_dist_bytes": 0.0, "max_dist_half_spacings": 0.0}

    def acc(c):
        for key in ("audited", "listed", "flips", "small_half_diffs"):
            total[key] += c[key]
        for key in ("max_dist_bytes", "max_dist_half_spacings"):
            total[key] = max(total[key], c[key])

    t0 = time.time()
    seed = 0x5EED0000 + (int(sys.argv[2], 0) if len(sys.argv) > 2 else 0)
    C2 = (1683, 1869, 2244, 2492)
    for content in ("structured", "random", "extremes", "mosaic", "natural"):
        for rep in range(2):
            seed += 1000
            acc(run("C2 two-pass RGBA8 (easu_fast_kernel -> UNORM8 intermediate)", *C2, content, N(24), seed, radius=2.0))
    for content in ("structured", "mosaic", "random", "natural"):
        seed += 1000
        acc(run("C2 fused=1 (fused_kernel, UNORM8 plane)", *C2, content, N(8), seed, radius=2.0, fused=1))
        seed += 1000
        acc(run("C2r masked sorted (radius 0.5)", *C2, content, N(16), seed, radius=0.5))
        seed += 1000
        acc(run("C2 EASU only -> UNORM8 output", *C2, content, N(8), seed, radius=2.0, stage_mask=1))
    for content in ("structured", "mosaic", "random", "natural"):
        seed += 1000
        acc(run("C4 x1.3 two-pass RGBA8", 2244, 2492, 2916, 3240, content, N(8), seed, radius=2.0))
    for (iw, ih, ow, oh, name) in ((1000, 900, 1695, 1525, "x1.7"), (1000, 900, 1111, 1000, "x1.11"), (1000, 900, 2000, 1800, "x2"), (997, 811, 1329, 1081, "x4/3 odd")):
        for content in ("mosaic", "random"):
            seed += 1000
            acc(run("odd ratio %s two-pass RGBA8" % name, iw, ih, ow, oh, content, N(16), seed, radius=2.0))
    C5 = (2370, 2370, 3160, 3160)
    for hs in (1.0, 6.0, 40.0):
        for content in ("structured", "mosaic", "random", "natural"):
            seed += 1000
            acc(run("C5 masked half pipeline (fused_kernel, half plane)", *C5, content, N(6), seed, half_scale=hs, radius=0.5))
            seed += 1000
            acc(run("C5 shape unmasked two-pass half (easu_fast_kernel -> half)", *C5, content, N(2), seed, half_scale=hs, radius=2.0, fused=0))
    print("TOTAL audited %d pixels, listed %d (%.2f %%), FLIPS %d, small-channel half differences %d (outside the contract: below xmin), "
          "max |product - strict| %.3e byte = %.2f of the 2^-9 band, %.3e half spacings = %.2f of the 2^-6 band   [%.0f s]"
          % (total["audited"], total["listed"], 100.0 * total["listed"] / max(1, total["audited"]), total["flips"], total["small_half_diffs"],
             total["max_dist_bytes"], total["max_dist_bytes"] * 512.0, total["max_dist_half_spacings"], total["max_dist_half_spacings"] * 64.0, time.time() - t0))
    sys.exit(1 if total["flips"] else 0)


if __name__ == "__main__":
    main()
