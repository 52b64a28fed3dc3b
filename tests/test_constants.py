"""Known-answer tests for the constant setup.  No GPU needed.

Three implementations are checked against the committed outputs of the reference's own code
(tests/golden/ref_consts.json, produced by tests/golden/make_golden.py from oracle/_ref):
  * the product library's host code (ovrfsr_easu_con / ovrfsr_rcas_con / ovrfsr_nis_* via the C ABI),
  * the oracle's C restatement,
  * and, where oracle/_ref exists, the reference again (guards against a stale fixture).
"""
import json
import os

import numpy as np
import pytest

import openvr_fsr_amd as A
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "ref_consts.json")))


def u32(hexlist):
    return np.array([int(h, 16) for h in hexlist], np.uint32)


def test_survey_kats_easu():
    # SURVEY.md 8c, captured from the compiled reference headers
    con = A.easu_con(1683, 1869, 2244, 2492)
    assert list(con[:4]) == [0x3f400000, 0x3f400000, 0xbe000000, 0xbe000000]
    assert list(con[4:8]) == [0x3a1bc28c, 0x3a0c424b, 0x3a1bc28c, 0xba0c424b]
    assert list(con[8:12]) == [0xba1bc28c, 0x3a8c424b, 0x3a1bc28c, 0x3a8c424b]
    assert list(con[12:]) == [0, 0x3b0c424b, 0, 0]
    con = A.easu_con(2244, 2492, 2916, 3240)
    assert list(con[:4]) == [0x3f45010e, 0x3f44e616, 0xbdebfbc8, 0xbdec67a8]


def test_survey_kats_rcas():
    stops = float(np.float32(2.0) - np.float32(2.0) * np.float32(0.9))
    assert abs(stops - 0.200000048) < 1e-9
    con = A.rcas_con(stops)
    assert list(con) == [0x3f5edc66, 0x3af63af6, 0, 0]


@pytest.mark.parametrize("case", G["easu_con"], ids=lambda c: "%dx%d" % tuple(c["in"]))
def test_easu_con_golden(case):
    want = u32(case["con"])
    iw, ih = case["in"]
    ow, oh = case["out"]
    assert np.array_equal(A.easu_con(iw, ih, ow, oh), want)
    assert np.array_equal(O.easu_con(iw, ih, ow, oh), want)


@pytest.mark.parametrize("case", G["rcas_con"], ids=lambda c: str(c["sharpness"]))
def test_rcas_con_golden(case):
    want = u32(case["con"])
    stops = float(u32([case["stops_bits"]]).view(np.float32)[0])
    assert np.array_equal(A.rcas_con(stops), want)
    got = O.rcas_con(case["sharpness"])
    assert np.array_equal(got, want)  # includes the clamp of sharpness to [0,1]


def test_f32_to_f16_trunc_golden():
    for bits, want in G["f32_to_f16"]:
        f = float(u32([bits]).view(np.float32)[0])
        assert O.lib().ovo_f32_to_f16_trunc(f) == int(want, 16), bits


def test_mask_constants_kats():
    # radius 0.5 -> r = uint(0.25*outH)
    c, r = A.mask_constants(2244, 2492, (0.5, 0.5, 0.5, 0.5), 0.5, True, 0)
    assert list(r) == [623, 388129, 2244, 2492] and list(c) == [1122, 1246, 1122, 1246]
    c, r = A.mask_constants(3160, 3160, (0.5, 0.5, 0.5, 0.5), 0.5, True, 1)
    assert list(r) == [790, 624100, 3160, 3160]
    # shared side-by-side texture: integer outW/2 first (PostProcessor.cpp:298,300)
    c, r = A.mask_constants(4489, 2492, (0.45, 0.5, 0.55, 0.52), 0.5, False, 0)
    half = 4489 // 2
    assert list(c) == [int(np.float32(half) * np.float32(0.45)), int(np.float32(2492) * np.float32(0.5)),
                       int(np.float32(half) * (np.float32(1) + np.float32(0.55))), int(np.float32(2492) * np.float32(0.52))]
    for args in [(2244, 2492, (0.37, 0.61, 0.58, 0.44), 0.73, True, 1), (1000, 900, (0.5, 0.5, 0.5, 0.5), 2.0, False, 0)]:
        a = A.mask_constants(*args)
        o = O.mask_constants(args[0], args[1], args[3], args[2], args[4], args[5])
        assert np.array_equal(a[0], o[0]) and np.array_equal(a[1], o[1])


@pytest.mark.parametrize("case", G["nis_scaler"], ids=lambda c: "%s-%dx%d" % (c["sharpness"], c["in"][0], c["out"][0]))
def test_nis_scaler_config_golden(case):
    ok, buf = A.nis_scaler_config(case["sharpness"], case["in"][0], case["in"][1], case["out"][0], case["out"][1])
    assert int(ok) == case["ok"]
    assert np.array_equal(buf, u32(case["cfg"]))  # all 256 bytes, including the half-filled failure case


@pytest.mark.parametrize("case", G["nis_sharpen"], ids=lambda c: str(c["sharpness"]))
def test_nis_sharpen_config_golden(case):
    ok, buf = A.nis_sharpen_config(case["sharpness"], case["in"][0], case["in"][1])
    assert int(ok) == case["ok"] and np.array_equal(buf, u32(case["cfg"]))


def test_nis_config_survey_kat():
    ok, buf = A.nis_scaler_config(0.9, 1683, 1869, 2244, 2492)
    f = buf.view(np.float32)
    assert ok and abs(f[0] - 1.10058594) < 1e-7 and f[1] == 0.0625 and f[2] == 2 and f[3] == 0.125
    assert abs(f[9] - 1.31999993) < 1e-7 and f[12] == 0.75 and f[13] == 0.75
    ok, _ = A.nis_scaler_config(0.9, 400, 400, 1000, 1000)  # scale 0.4 -> false
    assert not ok


def test_nis_coef_tables_golden():
    s, u = A.nis_coefs()
    assert np.array_equal(s.view(np.uint32).ravel(), u32(G["nis_coef_scale"]))
    assert np.array_equal(u.view(np.uint32).ravel(), u32(G["nis_coef_usm"]))
    assert (s[:, 6:] == 0).all() and (u[:, 6:] == 0).all()
    np.testing.assert_allclose(s[:, :6].sum(1), 1.0, atol=2e-3)   # polyphase rows are DC-normalised
    np.testing.assert_allclose(u[:, :6].sum(1), 0.0, atol=2e-3)   # USM rows are DC-free


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
def test_fixture_is_fresh_against_reference():
    R = O.ref()
    for case in G["easu_con"]:
        con = np.zeros(16, np.uint32)
        iw, ih = case["in"]
        ow, oh = case["out"]
        R.ref_easu_con(con.ctypes.data_as(O.u32p), iw, ih, iw, ih, ow, oh)
        assert np.array_equal(con, u32(case["con"]))
    rng = np.random.default_rng(0)
    for _ in range(2000):
        f = float((rng.uniform(1, 2) * 2.0 ** rng.integers(-40, 30)) * rng.choice([-1, 1]))
        f = float(np.float32(f))
        assert O.lib().ovo_f32_to_f16_trunc(f) == R.ref_f32_to_f16(f)


def test_unorm8_decode_two_op_form_is_correctly_rounded():
    """fsr_device.inc unorm8_to_unit: b/255 as fma(b, k_hi, RN(b*k_lo)) with 1/255 = k_hi + k_lo (0x3b808081, 0xaf7efeff) is the
    correctly rounded quotient for every byte -- checked here against exact rational arithmetic (the claim the kernel comment
    makes; every RGBA8 path of both builds decodes through it)."""
    from fractions import Fraction
    k_hi = np.array([0x3b808081], np.uint32).view(np.float32)[0]
    k_lo = np.array([0xaf7efeff], np.uint32).view(np.float32)[0]

    def rn(fr):  # round an exact Fraction to the nearest float32 (ties to even): float64 holds fr to 2^-53, far finer than
        return np.float32(float(fr))  # any float32 rounding boundary of these values (checked below by the margin assert)

    for b in range(256):
        lo = np.float32(np.float32(b) * k_lo)                              # RN(b * k_lo)
        exact = Fraction(int(b)) * Fraction(float(k_hi)) + Fraction(float(lo))   # fma: one rounding of the exact sum
        got = rn(exact)
        want = rn(Fraction(b, 255))
        assert got == want, (b, got, want)
        # margin: the exact fma argument and b/255 are both far (>= 2^-40 relative) from a float32 rounding boundary
        if b:
            ulp = float(np.spacing(want))
            for v in (exact, Fraction(b, 255)):
                frac = (v - Fraction(float(want))) / Fraction(ulp)
                assert abs(abs(frac) - Fraction(1, 2)) > Fraction(1, 2 ** 20), (b, float(frac))


def test_near_tie_band_formulas():
    """The near-tie guard's band tests (fsr_device.inc: near_tie_byte, half_tie_code), restated in numpy arithmetic
    and checked against the plain definition: a byte-domain value v is flagged iff frac(v) lies within 2^-9 of 0.5; a value
    x >= xmin about to be stored as half is flagged iff it lies within 2^-6 of a half spacing of the midpoint of two neighbouring
    half values (2^-17 in [0.5, 1), growing with the binade).  (The device functions
    are these expressions; the GPU tests prove their effect -- n_diff == 0 -- this pins the band they implement.)"""
    K = 9
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.uniform(0, 255, 200000), np.arange(0, 255)[:, None].repeat(64, 1).ravel() + 0.5 + rng.uniform(-2.0 ** -7, 2.0 ** -7, 255 * 64)]).astype(np.float32)
    bias = np.float32(256.0 + 2.0 ** -K)
    mask = np.uint32(((1 << (K - 1)) - 1) << (16 - K))
    flagged = ((v + bias).astype(np.float32).view(np.uint32) & mask) == np.uint32(0x4000)
    dist = np.abs((v.astype(np.float64) % 1.0) - 0.5)
    ulp = 2.0 ** -15   # v + 256 is rounded to multiples of 2^-15: the band edge is that sharp
    assert flagged[dist < 2.0 ** -K - ulp].all()
    assert not flagged[dist > 2.0 ** -K + ulp].any()
    assert 0.9 * 2.0 ** (1 - K) < flagged[:200000].mean() < 1.1 * 2.0 ** (1 - K)   # 2^(1-K) of uniformly distributed values

    # half stores (round 4: half_tie_code, fsr_device.inc): the low 13 bits of the fp32 pattern are x's position inside its half
    # spacing; flagged iff within 2^-6 of a SPACING of the boundary (midpoint of two neighbouring half values) and x >= xmin.
    # The band is relative: 2^-17 in [0.5, 1), 2^-16 in [1, 2), 2^-12 in [16, 32) ...
    KH = 6
    W = np.uint32(1 << (13 - KH))
    M = np.uint32(0x1fff & ~(2 * int(W) - 1))
    x = np.concatenate([rng.uniform(0.02, 1.9, 200000), rng.uniform(2.0, 60.0, 50000), np.float32(0.5) - rng.uniform(0, 3e-4, 2000),
                        np.float32(1.0) - rng.uniform(0, 6e-4, 2000), -rng.uniform(0.3, 2.0, 1000)]).astype(np.float32)
    bits = x.view(np.uint32)
    h16 = x.astype(np.float16)
    back = h16.astype(np.float64)
    up = np.nextafter(h16, np.float16(np.inf)).astype(np.float64)
    dn = np.nextafter(h16, np.float16(-np.inf)).astype(np.float64)
    x64 = x.astype(np.float64)
    mid = np.where(x64 >= back, (back + up) / 2, (back + dn) / 2)
    # spacing of x's OWN binade (a value just below a power of two rounds up into the next one; the code follows x)
    spacing = 2.0 ** (np.floor(np.log2(np.abs(x64))) - 10)
    d = np.abs(x64 - mid) / spacing
    for xmin in (np.float32(0.25), np.float32(0.5), np.float32(np.inf)):
        xb = np.float32(xmin).view(np.uint32)
        code = (((bits + W) & M) ^ np.uint32(0x1000)) | (((bits - xb) | bits) & np.uint32(0x80000000))
        flagged = code == 0
        sel = x >= xmin
        assert flagged[sel & (d < 2.0 ** -KH * 0.99)].all()
        assert not flagged[sel & (d > 2.0 ** -KH * 1.01 + 2.0 ** -12)].any()   # the fixed-point position has 13 bits
        assert not flagged[~sel].any()                                          # below xmin, negative, and everything when xmin = +inf
