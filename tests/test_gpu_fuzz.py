"""Seeded fuzz of the HIP path against the oracle (strict build: bit-exact): random sizes, scale ratios, masks,
projection centres, formats, and -- through torch views -- row pitches wider than the image and batch strides with
gaps.  Catches footprint/tile-extent/addressing corner cases that fixed shapes miss."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
STRICT, FP32 = 2, 0


def _cfg(rng):
    iw, ih = int(rng.integers(5, 150)), int(rng.integers(5, 150))
    s = float(rng.choice([0.5, 0.59, 0.67, 0.75, 0.77, 0.9, 0.97, rng.uniform(0.5, 1.0)]))
    ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
    radius = float(rng.choice([2.0, 2.0, rng.uniform(0.15, 1.3)]))
    proj = tuple(float(x) for x in rng.uniform(0.25, 0.75, 4))
    return iw, ih, ow, oh, radius, proj, int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))


@pytest.mark.parametrize("seed", range(24))
def test_fsr_fuzz_strict(gpu, seed):
    import torch
    import openvr_fsr_amd as A
    rng = np.random.default_rng(1000 + seed)
    iw, ih, ow, oh, radius, proj, eye, debug, sharp = _cfg(rng)
    gen = [synth.structured_u8, synth.random_u8, synth.extremes_u8][seed % 3]
    img8 = gen(iw, ih, seed)
    want = O.fsr_pipeline_u8(img8, ow, oh, sharpness=sharp, radius=radius, proj=proj, eye=eye, debug=debug)
    # input lives inside a wider buffer (row pitch > width*4), output likewise
    pad_in, pad_out = int(rng.integers(0, 9)), int(rng.integers(0, 9))
    big_in = torch.zeros((ih, iw + pad_in, 4), dtype=torch.uint8, device="cuda")
    big_in[:, :iw] = torch.from_numpy(img8).cuda()
    big_out = torch.full((oh, ow + pad_out, 4), 99, dtype=torch.uint8, device="cuda")
    for fused in (0, 1):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=sharp, radius=radius, proj_centre=proj,
                             debug_mode=debug, precision=STRICT, fused=fused)
        big_out.fill_(99)
        out = pp.apply(eye, big_in[:, :iw], out=big_out[:, :ow])
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got, want), (fused, iw, ih, ow, oh, radius)
        if pad_out:
            assert (big_out[:, ow:].cpu().numpy() == 99).all(), "wrote outside the output image"
        pp.close()


@pytest.mark.parametrize("seed", range(12))
def test_nis_fuzz_strict(gpu, seed):
    import torch
    import openvr_fsr_amd as A
    rng = np.random.default_rng(2000 + seed)
    iw, ih, ow, oh, radius, proj, eye, debug, sharp = _cfg(rng)
    ow, oh = min(ow, 2 * iw), min(oh, 2 * ih)     # NIS: 1x..2x
    gen = [synth.structured_u8, synth.random_u8, synth.extremes_u8][seed % 3]
    img8 = gen(iw, ih, seed)
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(sharp, iw, ih, ow, oh)
    assert ok
    centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
    want = O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, debug), cs, cu)
    pad_in = int(rng.integers(0, 9))
    big_in = torch.zeros((ih, iw + pad_in, 4), dtype=torch.uint8, device="cuda")
    big_in[:, :iw] = torch.from_numpy(img8).cuda()
    pp = A.PostProcessor(fsr_enabled=1, use_nis=1, out_width=ow, out_height=oh, sharpness=sharp, radius=radius,
                         proj_centre=proj, debug_mode=debug, precision=STRICT)
    out = pp.apply(eye, big_in[:, :iw], out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (iw, ih, ow, oh, radius)
    pp.close()


def test_batch_with_stride_gaps_and_formats(gpu):
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh = 70, 52, 93, 69
    n = 5
    proj = (0.45, 0.5, 0.55, 0.5)
    imgs8 = np.stack([synth.structured_u8(iw, ih, 40 + i) for i in range(n)])
    # images spaced 3 rows apart inside one allocation (stride > pitch*height), output too
    buf_in = torch.zeros((n, ih + 3, iw + 2, 4), dtype=torch.uint8, device="cuda")
    buf_in[:, :ih, :iw] = torch.from_numpy(imgs8).cuda()
    buf_out = torch.full((n, oh + 2, ow + 5, 4), 7, dtype=torch.uint8, device="cuda")
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9, radius=0.8, proj_centre=proj, precision=STRICT)
    pp.apply_batch(buf_in[:, :ih, :iw], buf_out[:, :oh, :ow], first_eye=1, alternate_eyes=True)
    torch.cuda.synchronize()
    got = buf_out.cpu().numpy()
    for i in range(n):
        want = O.fsr_pipeline_u8(imgs8[i], ow, oh, sharpness=0.9, radius=0.8, proj=proj, eye=1 ^ (i & 1))
        assert np.array_equal(got[i, :oh, :ow], want), i
    assert (got[:, oh:] == 7).all() and (got[:, :, ow:] == 7).all()
    pp.close()
    # float formats in and out: RGBA32F -> RGBA16F, strict, against the oracle on the same floats
    imgf = O.unorm8_to_float(imgs8[0])
    centre, rad = O.mask_constants(ow, oh, 2.0)
    e = O.easu(imgf, ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
    want = O.rcas(e, O.rcas_con(0.5), centre, rad).astype(np.float16)
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.5, radius=2.0, precision=STRICT, quantize_intermediate=0)
    out = pp.apply(0, torch.from_numpy(imgf).cuda(), out_dtype=torch.float16)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16))
    pp.close()


def _outside_px(ow, oh, centre, r2, gw, gh):
    """bool [oh, ow]: pixel lies in a gw x gh mask group that fails the radius test (uint32 wrap-around arithmetic of
    fsr_easu.hlsl:41-45 / NIS_Upscale.hlsl:98-101)."""
    gx = (np.arange(ow, dtype=np.uint32) // gw) * np.uint32(gw) + np.uint32(gw // 2)
    gy = (np.arange(oh, dtype=np.uint32) // gh) * np.uint32(gh) + np.uint32(gh // 2)
    c = np.asarray(centre, np.uint32)
    with np.errstate(over="ignore"):
        d1 = (c[0] - gx)[None, :] ** 2 + ((c[1] - gy) ** 2)[:, None]
        d2 = (c[2] - gx)[None, :] ** 2 + ((c[3] - gy) ** 2)[:, None]
    return ~((d1 <= np.uint32(r2)) | (d2 <= np.uint32(r2)))


@pytest.mark.parametrize("seed", range(16))
def test_masked_product_fuzz(gpu, seed):
    """Product build, radius-masked, every pipeline form (mask-sorted two-pass = auto, plain two-pass, fused): pixels
    of mask groups outside the radius are bit-identical to the oracle (unfused bilinear), pixels inside stay within the
    RCAS-amplified LSB bound; padded row pitches are respected.  Sizes span one to ~100 tiles, scales 0.5..0.99 and a few
    minifications (which take the generic outside kernel)."""
    import torch
    import openvr_fsr_amd as A
    rng = np.random.default_rng(3000 + seed)
    iw, ih = int(rng.integers(20, 330)), int(rng.integers(20, 330))
    s = float(rng.choice([0.5, 0.501, 0.67, 0.75, 0.77, 0.9, 0.99, rng.uniform(0.5, 1.0), 1.15]))
    ow, oh = max(8, int(iw / s)), max(8, int(ih / s))
    if s < 1:
        ow, oh = max(ow, iw + 1), max(oh, ih + 1)
    radius = float(rng.uniform(0.1, 0.9))
    proj = tuple(float(x) for x in rng.uniform(0.3, 0.7, 4))
    eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
    img8 = [synth.structured_u8, synth.random_u8][seed % 2](iw, ih, seed)
    want = O.fsr_pipeline_u8(img8, ow, oh, sharpness=sharp, radius=radius, proj=proj, eye=eye, debug=debug)
    centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
    outside = _outside_px(ow, oh, centre, rad[1], 16, 16)
    pad_in, pad_out = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    big_in = torch.zeros((ih, iw + pad_in, 4), dtype=torch.uint8, device="cuda")
    big_in[:, :iw] = torch.from_numpy(img8).cuda()
    big_out = torch.empty((oh, ow + pad_out, 4), dtype=torch.uint8, device="cuda")
    for fused in (-1, 0, 1):
        try:
            pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=sharp, radius=radius, proj_centre=proj,
                                 debug_mode=debug, precision=FP32, fused=fused)
            big_out.fill_(99)
            got = pp.apply(eye, big_in[:, :iw], out=big_out[:, :ow]).cpu().numpy()
        except A.OvrFsrError:
            assert fused == 1 and s > 1   # the fused kernel's tile footprint does not fit LDS when minifying
            continue
        finally:
            pp.close()
        assert np.array_equal(got[outside], want[outside]), (fused, iw, ih, ow, oh, radius, int((got[outside] != want[outside]).sum()))
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (fused, iw, ih, ow, oh, int(d.max()))
        if pad_out:
            assert (big_out[:, ow:].cpu().numpy() == 99).all(), "wrote outside the output image"


@pytest.mark.parametrize("seed", range(8))
def test_nis_masked_product_fuzz(gpu, seed):
    """NVScaler with a radius, product build: DirectCopy groups (mask-sorted, outside_rgba8_kernel for RGBA8 destinations,
    nis_outside_kernel otherwise) are bit-identical to the oracle."""
    import openvr_fsr_amd as A
    from tests.util import run_gpu
    rng = np.random.default_rng(4000 + seed)
    iw, ih = int(rng.integers(40, 300)), int(rng.integers(40, 300))
    s = float(rng.choice([0.5, 0.67, 0.75, 0.9, 0.99, rng.uniform(0.5, 1.0)]))
    ow, oh = min(2 * iw, max(iw + 1, int(iw / s))), min(2 * ih, max(ih + 1, int(ih / s)))
    radius = float(rng.uniform(0.1, 0.9))
    proj = tuple(float(x) for x in rng.uniform(0.3, 0.7, 4))
    eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
    img8 = [synth.structured_u8, synth.random_u8][seed % 2](iw, ih, seed)
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(sharp, iw, ih, ow, oh)
    assert ok
    centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
    want = O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, debug), cs, cu)
    outside = _outside_px(ow, oh, centre, rad[1], 32, 24)
    kw = dict(eye=eye, precision=FP32, use_nis=1, sharpness=sharp, radius=radius, proj_centre=proj, debug_mode=debug)
    got8 = run_gpu(img8, ow, oh, np.uint8, **kw)
    assert np.array_equal(got8[outside], O.float_to_unorm8(want)[outside]), (iw, ih, ow, oh, radius)
    gotf = run_gpu(img8, ow, oh, np.float32, **kw)
    assert np.array_equal(gotf[outside].view(np.uint32), want[outside].view(np.uint32)), (iw, ih, ow, oh, radius)
    assert np.abs(gotf - want).max() <= 1e-3   # north_star's max-abs (nis_getY is unfused in every build: no edge decision can flip)


@pytest.mark.parametrize("iw,ih,ow,oh", [(12288, 6, 16384, 8), (6, 12288, 8, 16384), (16383, 3, 16384, 5), (1, 1, 2, 2), (3, 2, 4, 3)])
def test_extreme_shapes(gpu, iw, ih, ow, oh):
    """The largest dimension the ABI accepts (16384) as thin strips, and the smallest images: strict build bit-exact,
    product build within 1 LSB, masked and unmasked (footprints, tile lists and tap tables at their extremes)."""
    from tests.util import run_gpu
    img8 = synth.random_u8(iw, ih, 5)
    for radius in (2.0, 0.4):
        want = O.fsr_pipeline_u8(img8, ow, oh, sharpness=0.6, radius=radius)
        got = run_gpu(img8, ow, oh, np.uint8, precision=STRICT, sharpness=0.6, radius=radius)
        assert np.array_equal(got, want), (radius,)
        gotp = run_gpu(img8, ow, oh, np.uint8, precision=FP32, sharpness=0.6, radius=radius)
        assert np.abs(gotp.astype(np.int16) - want.astype(np.int16)).max() <= 1, (radius,)


@pytest.mark.parametrize("cfg", [dict(radius=2.0, sharpness=0.8), dict(radius=0.6, sharpness=0.8), dict(radius=0.6, use_nis=1, sharpness=0.5), dict(radius=0.7, fused=1)],
                         ids=["two-pass", "masked sorted", "NVScaler masked", "fused masked"])
def test_largest_batch_in_one_launch(gpu, cfg):
    """65 535 images in one call (the grid's z extent; one more is refused): images 0, 1, 32 767, 65 533 and 65 534 equal what the same
    ctx writes for them one at a time -- the image index, its eye parity and `base + i * stride` at their extremes."""
    import torch
    import openvr_fsr_amd as A
    n, iw, ih, ow, oh = 65535, 36, 27, 48, 36
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    src = torch.randint(0, 256, (n, ih, iw, 4), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda")
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, proj_centre=(0.4, 0.5, 0.6, 0.5), **cfg)
    try:
        pp.apply_batch(src, out, first_eye=0, alternate_eyes=True)
        torch.cuda.synchronize()
        one = torch.zeros((1, oh, ow, 4), dtype=torch.uint8, device="cuda")
        for i in (0, 1, 32767, 65533, 65534):
            pp.apply_batch(src[i:i + 1], one, first_eye=i & 1)
            torch.cuda.synchronize()
            assert torch.equal(out[i], one[0]), "image %d of the batch differs from the same image on its own" % i
        big = torch.zeros((n + 1, ih, iw, 4), dtype=torch.uint8, device="cuda")
        with pytest.raises(A.OvrFsrError) as e:
            pp.apply_batch(big, torch.zeros((n + 1, oh, ow, 4), dtype=torch.uint8, device="cuda"))
        assert e.value.status == 1 and "65535" in str(e.value)
    finally:
        pp.close()


MAX_AREA = [
    ("EASU only RGBA8 -> RGBA32F, strict (4 GiB output: byte offsets up to 2^32 - 16)", dict(stage_mask=1, precision=STRICT), "u8", "f32"),
    ("EASU + RCAS RGBA16F, product (2 GiB output and intermediate)", dict(sharpness=0.7), "f16", "f16"),
    ("fused RGBA16F, product", dict(sharpness=0.7, fused=1), "f16", "f16"),
    ("NVScaler RGBA8 -> RGBA8 (1 GiB output)", dict(use_nis=1, sharpness=0.5), "u8", "u8"),
]


@pytest.mark.parametrize("name,cfg,inf,outf", MAX_AREA, ids=[c[0].split(" (")[0] for c in MAX_AREA])
def test_largest_image_addresses_every_band_like_a_small_one(gpu, name, cfg, inf, outf):
    """The largest image the ABI accepts at FULL area: 12288 x 12288 -> 16384 x 16384 (the strips of test_extreme_shapes reach the largest
    coordinate, not the largest byte offset).  Too large for the oracle; the property instead: at a scale of exactly 3/4 the band of output rows
    [4k, 4k + 400) depends on input rows [3k, 3k + 300) only (plus the filter's reach), so away from the band's own borders it must equal, bit for
    bit, what the library writes for that band of the input submitted as an image of its own -- an image small enough that every path it takes is
    pinned to the oracle elsewhere.  A 32-bit offset that wraps, a tile index that overflows or a row that lands in another tile shows up as a
    difference; bands at the top, in the middle, across the 2 GiB line and at the very end of the output."""
    import torch
    import openvr_fsr_amd as A
    IW = IH = 12288
    OW = OH = 16384
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    tin = {"u8": torch.uint8, "f16": torch.float16}[inf]
    tout = {"u8": torch.uint8, "f16": torch.float16, "f32": torch.float32}[outf]
    src = torch.empty((IH, IW, 4), dtype=tin, device="cuda")
    for r0 in range(0, IH, 1024):                      # smooth + noisy content, generated in slabs (no 2.4 GB temporaries)
        yy = torch.arange(r0, r0 + 1024, device="cuda", dtype=torch.float32)[:, None, None]
        xx = torch.arange(IW, device="cuda", dtype=torch.float32)[None, :, None]
        ch = torch.arange(4, device="cuda", dtype=torch.float32)[None, None, :]
        v = 127.5 + 90.0 * torch.sin(xx * 0.013 + ch) * torch.cos(yy * 0.011 - ch) + 30.0 * (torch.rand((1024, IW, 4), generator=g, device="cuda") - 0.5)
        v = v.clamp_(0, 255)
        src[r0:r0 + 1024] = v.to(torch.uint8) if inf == "u8" else (v / 255.0).to(torch.float16)
        del v
    out = torch.zeros((OH, OW, 4), dtype=tout, device="cuda")
    pp = A.PostProcessor(fsr_enabled=1, out_width=OW, out_height=OH, radius=2.0, **cfg)
    # (the radius is in units of outH / 2: 2.0 covers a square image, a 16384 x 400 band needs 50 to stay unmasked out to its corners)
    small = A.PostProcessor(fsr_enabled=1, out_width=OW, out_height=400, radius=50.0, **cfg)
    try:
        pp.apply_batch(src[None], out[None], first_eye=0)
        torch.cuda.synchronize()
        band = torch.zeros((400, OW, 4), dtype=tout, device="cuda")
        # Rows of the band compared: not the band's first and last TILE row (32 rows; 16 at the short last one).  Those see the band's own
        # clamped borders (NVScaler's 6-tap filter + edge map, RCAS behind EASU) -- and in the product build the tiles that touch an image
        # border run the bounds-checking instantiation of a kernel, whose contracted multiply-adds round a few pixels per million differently
        # from the interior instantiation (one half-ulp, inside the contract; found by this test: 29 of 6 M pixels of the half pipeline).
        edge = 32
        for k in (0, 1365, 2047, 2048, 2730, 3996):    # output rows 4k ..: top, 1/3, the 2 GiB line of a 4 GiB image, 2/3, the last band
            small.apply_batch(src[3 * k:3 * k + 300][None].contiguous(), band[None], first_eye=0)
            torch.cuda.synchronize()
            # (the image's own top / bottom borders ARE the band's; the last band starts half a tile row into the image's tiling: its rows
            # 368..383 are border-tile rows of the image and interior rows of the band)
            spans = [(0, 384)] if k == 0 else [(edge, 368), (384, 400)] if 4 * k + 400 == OH else [(edge, 384)]
            for lo, hi in spans:
                a, b = out[4 * k + lo:4 * k + hi], band[lo:hi]
                if not torch.equal(a.view(torch.uint8), b.view(torch.uint8)):
                    bad = (a != b).any(dim=2).nonzero()
                    raise AssertionError("%s: band at output row %d differs at %d pixels, first (row, column) %s" % (name, 4 * k, bad.shape[0], (bad[0] + torch.tensor([lo, 0], device=bad.device)).tolist()))
    finally:
        pp.close(); small.close()


@pytest.mark.parametrize("iw,ih,ow,oh", [(12288, 6, 16384, 8), (6, 12288, 8, 16384), (3, 2, 4, 3)])
def test_extreme_shapes_nis(gpu, iw, ih, ow, oh):
    import openvr_fsr_amd as A
    from tests.util import run_gpu
    img8 = synth.random_u8(iw, ih, 6)
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(0.5, iw, ih, ow, oh)
    assert ok
    for radius in (2.0, 0.4):
        centre, rad = O.mask_constants(ow, oh, radius)
        want = O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, 0), cs, cu)
        got = run_gpu(img8, ow, oh, np.float32, precision=STRICT, use_nis=1, sharpness=0.5, radius=radius)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (radius,)
        gotp = run_gpu(img8, ow, oh, np.float32, precision=FP32, use_nis=1, sharpness=0.5, radius=radius)
        assert np.abs(gotp - want).max() <= 1e-3, (radius,)


@pytest.mark.parametrize("seed", range(8))
def test_masked_product_fuzz_half(gpu, seed):
    """RGBA16F in / RGBA16F out with a radius (BASELINE C5's form), product build, auto and explicit pipeline forms:
    half intermediate, fused kernel on the tiles touching the radius, per-pixel outside kernel.  Strict build bit-exact,
    product build within the half-float tolerance of test_c4_c5_shapes_properties."""
    from tests.util import run_gpu
    rng = np.random.default_rng(5000 + seed)
    iw, ih = int(rng.integers(40, 300)), int(rng.integers(40, 300))
    s = float(rng.choice([0.5, 0.67, 0.75, 0.9, rng.uniform(0.5, 1.0)]))
    ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
    radius = float(rng.uniform(0.15, 0.9))
    proj = tuple(float(x) for x in rng.uniform(0.3, 0.7, 4))
    eye, debug, sharp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(0, 1))
    img8 = [synth.structured_u8, synth.random_u8][seed % 2](iw, ih, seed)
    imgh = (img8.astype(np.float32) / 255.0).astype(np.float16)
    centre, rad = O.mask_constants(ow, oh, radius, proj, True, eye)
    e = O.easu(imgh.astype(np.float32), ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
    e16 = e.astype(np.float16).astype(np.float32)                       # half-float intermediate texture
    want = O.rcas(e16, O.rcas_con(sharp, debug), centre, rad).astype(np.float16)
    kw = dict(eye=eye, sharpness=sharp, radius=radius, proj_centre=proj, debug_mode=debug)
    got = run_gpu(imgh, ow, oh, np.float16, precision=STRICT, **kw)
    assert np.array_equal(got, want), (iw, ih, ow, oh, radius)
    for fused in (-1, 0, 1):
        got = run_gpu(imgh, ow, oh, np.float16, precision=FP32, fused=fused, **kw).astype(np.float32)
        err = np.abs(got - want.astype(np.float32))
        assert (err <= 1e-3).mean() >= 0.999 and err.max() <= 2e-2, (fused, float((err <= 1e-3).mean()), float(err.max()))


@pytest.mark.parametrize("seed", range(6))
def test_ctx_lifecycle_stress(gpu, seed):
    """One ctx through a random sequence of set_config / reset / apply / apply_batch with changing input sizes, batch sizes,
    eye order, masks, pipeline forms and precisions: the lazy (re)build of per-configuration resources (constants, tile
    lists, tap tables, intermediate buffers, coefficient banks) must always match what was asked.  Every result is checked
    against the oracle (strict: bit-exact; product: <= 1 LSB and >= 99.5 % exact bytes for FSR, 99 % within 1 LSB for NIS)."""
    import torch
    import openvr_fsr_amd as A
    rng = np.random.default_rng(6000 + seed)
    ow, oh = 200, 160
    sizes = [(150, 120), (160, 128), (100, 80), (199, 159)]

    def random_cfg():
        return dict(fsr_enabled=1, out_width=ow, out_height=oh, use_nis=int(rng.integers(0, 2)), precision=int(rng.choice([STRICT, FP32])),
                    radius=float(rng.choice([2.0, 0.5, 0.3])), fused=int(rng.choice([-1, 0, 1])), debug_mode=int(rng.integers(0, 2)),
                    sharpness=float(rng.choice([0.2, 0.7, 0.9])), proj_centre=tuple(float(v) for v in rng.uniform(0.4, 0.6, 4)))

    def check(kw, img8, eye, got):
        if kw["use_nis"]:
            cs, cu = A.nis_coefs()
            ok, cfg = A.nis_scaler_config(kw["sharpness"], img8.shape[1], img8.shape[0], ow, oh)
            assert ok
            centre, rad = O.mask_constants(ow, oh, kw["radius"], kw["proj_centre"], True, eye)
            want = O.float_to_unorm8(O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, kw["debug_mode"]), cs, cu))
        else:
            want = O.fsr_pipeline_u8(img8, ow, oh, sharpness=kw["sharpness"], radius=kw["radius"], proj=kw["proj_centre"], eye=eye,
                                     debug=kw["debug_mode"])
        if kw["precision"] == STRICT:
            assert np.array_equal(got, want), kw
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16))
            if kw["use_nis"]:
                assert (d <= 1).mean() >= 0.99, (kw, float((d <= 1).mean()))
            else:
                assert d.max() <= 1 and (d == 0).mean() >= 0.995, (kw, int(d.max()), float((d == 0).mean()))

    kw = random_cfg()
    pp = A.PostProcessor(**kw)
    for step in range(10):
        op = rng.choice(["cfg", "reset", "apply", "apply", "batch", "batch"])
        if op == "cfg":
            kw = random_cfg()
            pp.set_config(A.Config.default(**kw))
        elif op == "reset":
            pp.reset()
        elif op == "apply":
            iw, ih = sizes[int(rng.integers(0, len(sizes)))]
            eye = int(rng.integers(0, 2))
            img8 = synth.structured_u8(iw, ih, 100 * seed + step)
            got = pp.apply(eye, torch.from_numpy(img8).cuda(), out_dtype=torch.uint8).cpu().numpy()
            check(kw, img8, eye, got)
        else:
            iw, ih = sizes[int(rng.integers(0, len(sizes)))]
            n, first, alt = int(rng.integers(1, 6)), int(rng.integers(0, 2)), bool(rng.integers(0, 2))
            imgs = np.stack([synth.random_u8(iw, ih, 1000 * seed + 10 * step + i) for i in range(n)])
            outs = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device="cuda")
            pp.apply_batch(torch.from_numpy(imgs).cuda(), outs, first_eye=first, alternate_eyes=alt)
            got = outs.cpu().numpy()
            for i in range(n):
                check(kw, imgs[i], first ^ (i & 1) if alt else first, got[i])
    pp.close()
