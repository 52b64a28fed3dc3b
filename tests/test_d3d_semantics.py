"""The D3D11 fixed-function behaviours the oracle assumes, each pinned in isolation.

oracle/hlsl_shim.hpp (through which the reference's HLSL is compiled into oracle/_ref) and the C restatement
(oracle/fsr_oracle.c) both restate four things the D3D texture unit does for the reference: Gather4 component order and
footprint, clamp addressing, Load out-of-bounds = 0 and 8-bit sub-texel bilinear weights.  An error there would be
common to both, so each behaviour is tested here on its own -- against the statement of the D3D11 functional spec, against
the reference's own tap comments (src/fsr/ffx_fsr1.h:329-360) and, for the C restatement, against the shim.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

f32p = C.POINTER(C.c_float)


def _img(w, h, seed=None):
    """RGBA float image with a unique, position-coded red channel: R = 100*y + x (G, B offset by 1000 / 2000)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = 100 * y + x
    img[..., 1] = 1000 + 100 * y + x
    img[..., 2] = 2000 + 100 * y + x
    img[..., 3] = 1.0
    if seed is not None:
        img[..., :3] = np.random.default_rng(seed).random((h, w, 3), dtype=np.float32)
    return np.ascontiguousarray(img)


def _call(fn, img, *args):
    out = np.zeros(4, np.float32)
    h, w = img.shape[:2]
    fn(img.ctypes.data_as(f32p), w, h, *args, out.ctypes.data_as(f32p))
    return out


def gather(img, u, v, ch=0):
    return _call(O.shim_probe().probe_gather, img, np.float32(u), np.float32(v), ch)


def sample(img, u, v):
    return _call(O.shim_probe().probe_sample, img, np.float32(u), np.float32(v))


def load(img, x, y):
    return _call(O.shim_probe().probe_load, img, int(x), int(y))


def test_gather_component_order():
    """.x = bottom-left, .y = bottom-right, .z = top-right, .w = top-left of the 2x2 footprint of u*W-0.5 (D3D11 gather4:
    (-,+),(+,+),(+,-),(-,-)); the same for every channel selector."""
    w, h = 7, 5
    img = _img(w, h)
    for x0 in range(w - 1):
        for y0 in range(h - 1):
            # a coordinate strictly inside the quad (x0..x0+1, y0..y0+1): texel space = u*W - 0.5 in (x0, x0+1)
            for du, dv in ((0.5, 0.5), (0.01, 0.99), (0.99, 0.01)):
                u, v = (x0 + 0.5 + du) / w, (y0 + 0.5 + dv) / h
                for ch in range(3):
                    want = [img[y0 + 1, x0, ch], img[y0 + 1, x0 + 1, ch], img[y0, x0 + 1, ch], img[y0, x0, ch]]
                    assert gather(img, u, v, ch).tolist() == want, (x0, y0, du, dv, ch)


def test_easu_gather_positions_match_the_tap_comments():
    """FsrEasuCon's con1..con3 (the normalised gather offsets) with the gather order above must fetch exactly the taps the
    reference's comments name (src/fsr/ffx_fsr1.h:329-360): bczz = (b, c, -, -), ijfe = (i, j, f, e), klhg = (k, l, h, g),
    zzon = (-, -, o, n) around f = (fx, fy):        b c
                                                  e f g h
                                                  i j k l
                                                    n o                                                                    """
    iw, ih, ow, oh = 13, 11, 17, 15
    con = O.easu_con(iw, ih, ow, oh).view(np.float32)
    con1, con2, con3 = con[4:8], con[8:12], con[12:16]
    img = _img(iw, ih)
    R = lambda x, y: float(img[y, x, 0])
    for fx in range(1, iw - 2):
        for fy in range(1, ih - 2):
            p0 = (np.float32(fx) * con1[0] + con1[2], np.float32(fy) * con1[1] + con1[3])   # ffx_fsr1.h:344
            p1 = (p0[0] + con2[0], p0[1] + con2[1])
            p2 = (p0[0] + con2[2], p0[1] + con2[3])
            p3 = (p0[0] + con3[0], p0[1] + con3[1])
            bczz, ijfe, klhg, zzon = gather(img, *p0), gather(img, *p1), gather(img, *p2), gather(img, *p3)
            assert (bczz[0], bczz[1]) == (R(fx, fy - 1), R(fx + 1, fy - 1))                                        # b c
            assert ijfe.tolist() == [R(fx - 1, fy + 1), R(fx, fy + 1), R(fx, fy), R(fx - 1, fy)]                   # i j f e
            assert klhg.tolist() == [R(fx + 1, fy + 1), R(fx + 2, fy + 1), R(fx + 2, fy), R(fx + 1, fy)]           # k l h g
            assert (zzon[2], zzon[3]) == (R(fx + 1, fy + 2), R(fx, fy + 2))                                        # o n


def test_clamp_addressing():
    """D3D11_TEXTURE_ADDRESS_CLAMP: every texel index of a footprint is clamped to the image individually."""
    w, h = 6, 4
    img = _img(w, h)
    # left of the image: both columns of the quad are column 0; above: both rows are row 0
    g = gather(img, -0.3, 0.5 * (1.5 + 0.5) / h * 2 / 2)            # tx < 0, ty = inside rows 0..1
    assert g[0] == g[1] and g[2] == g[3]
    assert gather(img, -5.0, -5.0).tolist() == [img[0, 0, 0]] * 4
    assert gather(img, 9.0, 9.0).tolist() == [img[h - 1, w - 1, 0]] * 4
    # quad straddling the right edge: x0 = w-1, x0+1 clamps back to w-1
    u, v = (w - 1 + 0.5 + 0.25) / w, (1 + 0.5 + 0.25) / h
    assert gather(img, u, v).tolist() == [img[2, w - 1, 0], img[2, w - 1, 0], img[1, w - 1, 0], img[1, w - 1, 0]]
    # a bilinear sample far outside returns the corner texel exactly
    assert sample(img, -3.0, 7.0)[:3].tolist() == img[h - 1, 0, :3].tolist()
    assert sample(img, 1.0, 0.0)[:3].tolist() == img[0, w - 1, :3].tolist()      # u = 1: texel space W-0.5 -> quad (W-1, W) -> clamp


def test_load_out_of_bounds_is_zero():
    """ld / Texture2D::Load / operator[]: an address outside the resource returns 0 in every component (not the edge texel)."""
    w, h = 5, 3
    img = _img(w, h)
    for (x, y) in [(-1, 0), (0, -1), (w, 0), (0, h), (-7, -7), (w + 3, h + 3), (2, h), (w, 1)]:
        assert load(img, x, y).tolist() == [0.0, 0.0, 0.0, 0.0], (x, y)
    for (x, y) in [(0, 0), (w - 1, h - 1), (2, 1)]:
        assert load(img, x, y).tolist() == img[y, x].tolist()


def test_bilinear_has_8_subtexel_bits():
    """D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8: between two texel centres the weight takes exactly the 257 values k/256,
    a sample at a texel centre returns the texel exactly, and the snap is round-to-nearest (half up) of the texel-space
    coordinate u*W - 0.5."""
    img = np.zeros((1, 2, 4), np.float32)
    img[0, 1, :3] = 1.0                      # R: 0 | 1  -> the sample IS the weight of the right texel
    img[..., 3] = 1.0
    assert sample(img, 0.25, 0.5)[0] == 0.0 and sample(img, 0.75, 0.5)[0] == 1.0      # centres: t = 0 and t = 1
    us = np.linspace(0.25, 0.75, 8193, dtype=np.float64)
    vals = np.array([sample(img, u, 0.5)[0] for u in us], np.float64)
    steps = np.unique(vals)
    assert len(steps) == 257 and np.array_equal(steps, np.arange(257) / 256.0)
    assert np.all(np.diff(vals) >= 0)
    # round to nearest of t*256: t = u*2 - 0.5
    for k in (0, 1, 7, 128, 200, 255):
        t_lo, t_hi = (k - 0.49) / 256.0, (k + 0.49) / 256.0
        for t in (t_lo, t_hi):
            assert sample(img, (t + 0.5) / 2.0, 0.5)[0] == max(0, k) / 256.0, (k, t)
    # 2-D: weights are products of the two snapped fractions, blended in the order ((c00 w00 + c10 w10) + c01 w01) + c11 w11
    rng = np.random.default_rng(3)
    q = np.ascontiguousarray(rng.random((2, 2, 4), dtype=np.float32))
    for (fx8, fy8) in [(0, 0), (256, 256), (37, 201), (128, 128), (255, 1)]:
        fx, fy = np.float32(fx8 / 256.0), np.float32(fy8 / 256.0)
        u, v = (fx8 / 256.0 + 0.5) / 2.0, (fy8 / 256.0 + 0.5) / 2.0
        w00, w10 = (np.float32(1) - fx) * (np.float32(1) - fy), fx * (np.float32(1) - fy)
        w01, w11 = (np.float32(1) - fx) * fy, fx * fy
        want = ((q[0, 0] * w00 + q[0, 1] * w10) + q[1, 0] * w01) + q[1, 1] * w11
        assert sample(q, u, v).tolist() == want.astype(np.float32).tolist(), (fx8, fy8)


def test_unorm_store():
    """RWTexture2D<unorm float4>: values are saturated on store (NaN -> 0); in-range values are kept (the UNORM8 rounding
    floor(x*255+0.5) is the format conversion that follows, src/fsr/ffx_fsr1.h:1075-1080)."""
    P = O.shim_probe()
    src = np.array([-0.5, 0.25, 1.5, np.nan], np.float32)
    out = np.zeros(4, np.float32)
    P.probe_unorm_store(src.ctypes.data_as(f32p), out.ctypes.data_as(f32p))
    assert out.tolist() == [0.0, 0.25, 1.0, 0.0]
    x = np.array([0.0, 0.5 / 255, 0.49999 / 255, 1.0, 254.5 / 255, 0.3], np.float32)
    assert O.float_to_unorm8(x).tolist() == [int(np.floor(np.float32(v) * np.float32(255) + np.float32(0.5))) for v in x]


# ---- the C restatement (what the GPU parity tests compare against) agrees with the shim, behaviour by behaviour -------
def test_oracle_bilinear_fallback_equals_shim():
    """fsr_easu.hlsl:33-36: outside the radius EASU is SampleLevel(linearClamp, pos/outSize).  With radius 0 every pixel takes
    that branch, so the C oracle's sampler (clamp + 8-bit snap + blend order) is compared with the shim's, bit for bit."""
    iw, ih, ow, oh = 19, 13, 27, 18
    img = _img(iw, ih, seed=5)
    centre, rad = O.mask_constants(ow, oh, 0.0)
    got = O.easu(img, ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
    for oy in range(oh):
        for ox in range(ow):
            u, v = np.float32(ox) / np.float32(ow), np.float32(oy) / np.float32(oh)
            want = sample(img, u, v)
            assert got[oy, ox, :3].tolist() == want[:3].tolist(), (ox, oy)
    assert np.all(got[..., 3] == 1.0)


def test_oracle_rcas_border_equals_explicit_zero_ring():
    """FsrRcasF loads its 4 neighbours with Load (src/fsr/ffx_fsr1.h:698-707): out-of-image taps are 0.  Property: RCAS of an
    image equals the interior of RCAS of the same image with an explicit ring of zero texels, everywhere except the ring."""
    w, h = 23, 17
    img = _img(w, h, seed=9)
    con = O.rcas_con(0.8)
    # (a large radius: at the default 2.0 the corner groups of a wide image already fall outside r = outH)
    centre, rad = O.mask_constants(w, h, 100.0)
    out = O.rcas(img, con, centre, rad)
    pad = np.zeros((h + 2, w + 2, 4), np.float32)
    pad[1:-1, 1:-1] = img
    cp, rp = O.mask_constants(w + 2, h + 2, 100.0)
    outp = O.rcas(pad, con, cp, rp)
    a, b = np.ascontiguousarray(out[..., :3]), np.ascontiguousarray(outp[1:-1, 1:-1, :3])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_easu_clamps_taps_like_gather():
    """EASU at the image border: the C oracle clamps each of its 12 tap coordinates individually, which must equal running
    the same filter on an image extended by replicated edge texels (clamp addressing made explicit)."""
    iw, ih, ow, oh = 12, 9, 16, 12
    img = _img(iw, ih, seed=11)
    c0, r0 = O.mask_constants(ow, oh, 100.0)
    out = O.easu(img, ow, oh, None, c0, r0)
    padn = 3
    ext = np.pad(img, ((padn, padn), (padn, padn), (0, 0)), mode="edge")
    # same mapping: output pixel p samples input position (p + 0.5) * in/out - 0.5; in the extended image that position is
    # shifted by padn texels, which a pure translation of the output grid by padn*out/in reproduces only for integer shifts:
    # choose the extension so that it is: padn * ow/iw = 4 output pixels
    eow, eoh = ow + 2 * 4, oh + 2 * 4
    assert (iw + 2 * padn) * ow == eow * iw and (ih + 2 * padn) * oh == eoh * ih
    c1, r1 = O.mask_constants(eow, eoh, 100.0)
    oute = O.easu(ext, eow, eoh, None, c1, r1)
    a, b = np.ascontiguousarray(out[..., :3]), np.ascontiguousarray(oute[4:-4, 4:-4, :3])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
