#!/usr/bin/env python3
"""tests/debug/oracle_pin_campaign.py [seconds=300] [first_seed=100000] -- the PIN of the oracle, widened (CPU only).

Everything in this repository is measured against oracle/ (the C restatement); the oracle itself is pinned against oracle/_ref -- the reference's
own HLSL bodies compiled through oracle/hlsl_shim.hpp -- by tests/test_oracle*.py on a handful of seeds.  This campaign runs the same comparison,
bit for bit, for as long as it is given: random sizes (8..260) and scales (0.5..1), mask radii and centres, both eyes, debug tint on / off, sharpness
0..1; content families: the three synthetic generators, crops of the natural fixtures, HDR floats (unit content x 6 / x 40), and WILD floats
(negative values, denormals, zeros and 1e30s, no NaN / Inf); a seventh family holds NaN / Inf texels (HLSL leaves min / max of a NaN to the
implementation: the shim and the restatement agree on IEEE fmin / fmax semantics, which is what is compared; two NaNs count as equal).  That family
found the one place where the restatement took a shortcut that only non-finite texels can see -- NIS luma tiles loaded as texels instead of through
the reference's SampleLevel at texel centres (weights 1, 0, 0, 0: 0 x NaN) -- restated since; all seven families agree.
Stages: EASU, RCAS behind the UNORM8-quantised EASU result and behind the float one, NVScaler, NVSharpen."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import natural, synth  # noqa: E402


def same_bits(a, b):
    """bit for bit; two NaNs count as equal whatever their sign and payload (no operation of either implementation defines them)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    eq = a.view(np.uint32) == b.view(np.uint32)
    return bool((eq | (np.isnan(a) & np.isnan(b))).all())


def content(family, w, h, seed, rng):
    if family < 3:
        return O.unorm8_to_float([synth.structured_u8, synth.random_u8, synth.extremes_u8][family](w, h, seed))
    if family == 3:
        return O.unorm8_to_float(natural.tiled_u8(w, h, seed))
    if family == 4:
        return O.unorm8_to_float(synth.random_u8(w, h, seed)) * np.float32(rng.choice([6.0, 40.0]))
    img = rng.standard_normal((h, w, 4)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 3))
    sel = rng.random((h, w, 4))
    img[sel < 0.05] = 0.0
    img[(sel >= 0.05) & (sel < 0.08)] = np.float32(1e-41)     # denormal
    img[(sel >= 0.08) & (sel < 0.10)] = np.float32(1e30)
    img[(sel >= 0.10) & (sel < 0.12)] = np.float32(-1e30)
    if family == 6:
        img[(sel >= 0.12) & (sel < 0.13)] = np.nan
        img[(sel >= 0.13) & (sel < 0.14)] = np.inf
        img[(sel >= 0.14) & (sel < 0.15)] = -np.inf
    return img


NAMES = ["structured", "random", "extremes", "natural", "hdr", "wild floats", "wild floats + NaN / Inf"]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    assert O.have_ref(), "oracle/_ref is not built (needs /root/reference; __graft_entry__.build())"
    cs, cu = O.ref_nis_coefs()
    t0 = time.time()
    stats = {n: [0, 0, 0] for n in NAMES}     # instances, pixels, mismatches
    first_bad = []
    while time.time() - t0 < seconds:
        rng = np.random.default_rng(seed)
        family = seed % 7
        iw, ih = int(rng.integers(8, 260)), int(rng.integers(8, 260))
        s = float(rng.choice([0.5, 0.59, 0.67, 0.75, 0.77, rng.uniform(0.5, 1.0)]))
        ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
        img = content(family, iw, ih, seed, rng)
        proj = tuple(rng.uniform(0.2, 0.8, 4))
        eye, dbg = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        centre, rad = O.mask_constants(ow, oh, float(rng.choice([2.0, rng.uniform(0.15, 1.3)])), proj, True, eye)
        con = O.easu_con(iw, ih, ow, oh)
        bad = []
        a = O.easu(img, ow, oh, con, centre, rad)
        if not same_bits(a, O.ref_easu(img, ow, oh, con, centre, rad)): bad.append("easu")
        rcon = O.rcas_con(float(rng.uniform(0, 1)), dbg)
        q = O.unorm8_to_float(O.float_to_unorm8(np.nan_to_num(a, nan=0.0, posinf=1.0, neginf=0.0)))
        if not same_bits(O.rcas(q, rcon, centre, rad), O.ref_rcas(q, rcon, centre, rad)): bad.append("rcas(unorm8)")
        if not same_bits(O.rcas(a, rcon, centre, rad), O.ref_rcas(a, rcon, centre, rad)): bad.append("rcas(float)")
        px = 3 * ow * oh
        if ow <= 2 * iw and oh <= 2 * ih:
            sharp = float(rng.uniform(0, 1))
            ok, cfg = O.ref_nis_scaler_config(sharp, iw, ih, ow, oh)
            if ok:
                blk = O.nis_block(cfg, centre, rad, dbg)
                if not same_bits(O.nis_upscale(img, ow, oh, blk, cs, cu), O.ref_nis_upscale(img, ow, oh, blk, cs, cu)): bad.append("nis_upscale")
                px += ow * oh
            ok, cfg = O.ref_nis_scaler_config(sharp, iw, ih, iw, ih)
            c2, r2 = O.mask_constants(iw, ih, float(rng.uniform(0.3, 1.5)), proj, True, eye)
            blk = O.nis_block(cfg, c2, r2, dbg)
            if not same_bits(O.nis_sharpen(img, blk), O.ref_nis_sharpen(img, blk, cs, cu)): bad.append("nis_sharpen")
            px += iw * ih
        st = stats[NAMES[family]]
        st[0] += 1; st[1] += px; st[2] += len(bad)
        if bad and len(first_bad) < 10:
            first_bad.append((seed, NAMES[family], iw, ih, ow, oh, bad))
        seed += 1
    print("family                        instances        pixels   stages that differ from oracle/_ref")
    for n in NAMES:
        print("%-28s %10d %13d   %d" % (n, *stats[n]))
    tot = [sum(stats[n][i] for n in NAMES) for i in range(3)]
    print("TOTAL %d instances, %d output pixels compared bit for bit, %d mismatching stages   [%.0f s, seeds %s..%d]" % (tot[0], tot[1], tot[2], time.time() - t0, sys.argv[2] if len(sys.argv) > 2 else "100000", seed - 1))
    for f in first_bad:
        print("MISMATCH", f)
    sys.exit(1 if tot[2] else 0)


if __name__ == "__main__":
    main()
