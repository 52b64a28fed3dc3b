"""The CPU oracle (oracle/fsr_oracle.c) against the reference.  No GPU needed.

  * bit-exact against the committed golden vectors (outputs of the reference's own FsrEasuF/FsrRcasF
    bodies and shader entry points, tests/golden/fsr_vectors.npz);
  * bit-exact against oracle/_ref on fresh random inputs where the reference is available;
  * the analytical invariants of SURVEY.md 8c.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

HERE = os.path.dirname(os.path.abspath(__file__))
V = np.load(os.path.join(HERE, "golden", "fsr_vectors.npz"))
META = json.loads(bytes(V["meta"]).decode())


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("m", META, ids=lambda m: m["name"])
def test_oracle_matches_golden(m):
    name = m["name"]
    img8 = V[name + "_in"]
    consts = V[name + "_consts"]
    con, rcon, centre, rad = consts[:16], consts[16:20], consts[20:24], consts[24:28]
    ow, oh = m["out"]
    # the constants themselves are restated too
    assert np.array_equal(O.easu_con(m["in"][0], m["in"][1], ow, oh), con)
    assert np.array_equal(O.rcas_con(m["sharpness"], m["debug"]), rcon)
    c2, r2 = O.mask_constants(ow, oh, m["radius"], m["proj"], True, m["eye"])
    assert np.array_equal(c2, centre) and np.array_equal(r2, rad)
    easu = O.easu(O.unorm8_to_float(img8), ow, oh, con, centre, rad)
    assert same_bits(easu, V[name + "_easu"])
    mid = O.unorm8_to_float(O.float_to_unorm8(easu))
    rcas = O.rcas(mid, rcon, centre, rad)
    assert same_bits(rcas, V[name + "_rcas"])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_reference_random(seed):
    rng = np.random.default_rng(seed)
    iw, ih = int(rng.integers(8, 90)), int(rng.integers(8, 90))
    s = rng.uniform(0.5, 0.95)
    ow, oh = max(iw + 1, int(iw / s)), max(ih + 1, int(ih / s))
    gen = [synth.structured_u8, synth.random_u8, synth.extremes_u8][seed % 3]
    img = O.unorm8_to_float(gen(iw, ih, seed))
    con = O.easu_con(iw, ih, ow, oh)
    proj = tuple(rng.uniform(0.2, 0.8, 4))
    centre, rad = O.mask_constants(ow, oh, float(rng.uniform(0.2, 1.2)), proj, True, seed & 1)
    a = O.easu(img, ow, oh, con, centre, rad)
    assert same_bits(a, O.ref_easu(img, ow, oh, con, centre, rad))
    q = O.unorm8_to_float(O.float_to_unorm8(a))
    rcon = O.rcas_con(float(rng.uniform(0, 1)), seed & 1)
    assert same_bits(O.rcas(q, rcon, centre, rad), O.ref_rcas(q, rcon, centre, rad))


def test_unorm8_roundtrip():
    b = np.arange(256, dtype=np.uint8)
    f = O.unorm8_to_float(b)
    assert f[0] == 0.0 and f[255] == 1.0
    assert np.array_equal(f, (b.astype(np.float32) / np.float32(255.0)))
    assert np.array_equal(O.float_to_unorm8(f), b)
    assert list(O.float_to_unorm8(np.array([-1.0, 0.5 / 255 - 1e-7, 0.5 / 255 + 1e-7, 2.0, np.nan], np.float32))) == [0, 0, 1, 255, 0]


def test_easu_constant_image_is_exact():
    # dering clamp (ffx_fsr1.h:437): min4 == max4 == c -> output == c exactly
    img = np.empty((20, 24, 4), np.float32)
    img[...] = np.array([0.25, 0.5, 0.75, 1.0], np.float32)
    out = O.easu(img, 32, 27)
    assert (out[..., 0] == 0.25).all() and (out[..., 1] == 0.5).all() and (out[..., 2] == 0.75).all() and (out[..., 3] == 1).all()


def test_easu_output_within_2x2_neighbourhood():
    iw, ih, ow, oh = 40, 30, 53, 40
    img = O.unorm8_to_float(synth.random_u8(iw, ih, 9))
    out = O.easu(img, ow, oh)
    con = O.easu_con(iw, ih, ow, oh).view(np.float32)
    for y in range(oh):
        for x in range(ow):
            fx = int(np.floor(np.float32(x) * con[0] + con[2]))
            fy = int(np.floor(np.float32(y) * con[1] + con[3]))
            xs = np.clip([fx, fx + 1], 0, iw - 1)
            ys = np.clip([fy, fy + 1], 0, ih - 1)
            blk = img[np.ix_(ys, xs)][..., :3].reshape(-1, 3)
            assert (out[y, x, :3] >= blk.min(0)).all() and (out[y, x, :3] <= blk.max(0)).all()


def test_mask_outside_is_bilinear_and_blocky():
    iw, ih, ow, oh = 60, 60, 80, 80
    img = O.unorm8_to_float(synth.structured_u8(iw, ih, 2))
    con = O.easu_con(iw, ih, ow, oh)
    centre, rad = O.mask_constants(ow, oh, 0.5, (0.5, 0.5, 0.5, 0.5), True, 0)
    full_c, full_r = O.mask_constants(ow, oh, 2.0)
    masked = O.easu(img, ow, oh, con, centre, rad)
    full = O.easu(img, ow, oh, con, full_c, full_r)
    same = np.all(masked == full, axis=-1)
    # the decision is per 16x16 group: every group is either entirely identical to full EASU or (almost surely) not
    inside_groups = 0
    for gy in range(5):
        for gx in range(5):
            blk = same[gy * 16:(gy + 1) * 16, gx * 16:(gx + 1) * 16]
            cx, cy = gx * 16 + 8, gy * 16 + 8
            inside = (int(centre[0]) - cx) ** 2 + (int(centre[1]) - cy) ** 2 <= int(rad[1])
            inside_groups += inside
            if inside:
                assert blk.all()
            else:
                assert not blk.all()
    assert 0 < inside_groups < 25
    # outside pixel = bilinear sample at pos/outSize (no half-pixel centre): check one by hand
    x, y = 3, 5
    tx, ty = np.float32(x) / np.float32(ow) * np.float32(iw) - np.float32(0.5), np.float32(y) / np.float32(oh) * np.float32(ih) - np.float32(0.5)
    # D3D11 texel addressing: 8 fractional bits (D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT), round to nearest
    sx, sy = np.floor(tx * 256 + 0.5), np.floor(ty * 256 + 0.5)
    x0, y0 = int(np.floor(sx / 256)), int(np.floor(sy / 256))
    fx, fy = np.float32((sx - x0 * 256) / 256), np.float32((sy - y0 * 256) / 256)
    cl = lambda v, hi: min(max(v, 0), hi)
    p = lambda xx, yy: img[cl(yy, ih - 1), cl(xx, iw - 1), :3]
    want = p(x0, y0) * (1 - fx) * (1 - fy) + p(x0 + 1, y0) * fx * (1 - fy) + p(x0, y0 + 1) * (1 - fx) * fy + p(x0 + 1, y0 + 1) * fx * fy
    np.testing.assert_allclose(masked[y, x, :3], want, atol=1e-6)


def test_rcas_properties():
    w, h = 50, 37
    img = O.unorm8_to_float(synth.structured_u8(w, h, 5))
    out0 = O.rcas(img, O.rcas_con(0.0))
    out1 = O.rcas(img, O.rcas_con(1.0))
    assert (out0[..., 3] == 1).all() and np.isfinite(out1).all()
    # lobe in [-0.1875*sharp, 0]: a constant image is a fixed point, and sharpening never inverts contrast sign locally
    const = np.empty((16, 16, 4), np.float32)
    const[...] = np.array([0.2, 0.4, 0.6, 1.0], np.float32)
    np.testing.assert_allclose(O.rcas(const, O.rcas_con(1.0))[2:-2, 2:-2, :3], const[2:-2, 2:-2, :3], atol=2e-3)
    # sharper setting moves further from the input than the weaker one
    assert np.abs(out1 - img)[..., :3].mean() > np.abs(out0 - img)[..., :3].mean()
    # border taps read 0 (Texture2D.Load OOB), they are not clamped: corner differs from interior behaviour but stays finite
    assert np.isfinite(out1[0, 0]).all()


def test_rcas_debug_tint_outside_radius():
    w, h = 64, 64
    img = O.unorm8_to_float(synth.random_u8(w, h, 1))
    centre, rad = O.mask_constants(w, h, 0.3)
    out = O.rcas(img, O.rcas_con(0.5, debug=1), centre, rad)
    # group (0,0) is outside: copy x (1, .7, .7, 1), alpha passes through
    np.testing.assert_array_equal(out[:16, :16, 0], img[:16, :16, 0])
    np.testing.assert_array_equal(out[:16, :16, 1], (np.float32(1.0) - np.float32(1.0) * np.float32(0.3)) * img[:16, :16, 1])
    np.testing.assert_array_equal(out[:16, :16, 3], img[:16, :16, 3])


def test_pipeline_u8_stages_and_alpha():
    img8 = synth.structured_u8(48, 40, 3)
    out = O.fsr_pipeline_u8(img8, 64, 53)
    assert out.shape == (53, 64, 4) and (out[..., 3] == 255).all()
    e = O.fsr_pipeline_u8(img8, 64, 53, stages=1)
    want = O.float_to_unorm8(O.easu(O.unorm8_to_float(img8), 64, 53))
    assert np.array_equal(e, want)
    with pytest.raises(RuntimeError):
        O.fsr_pipeline_u8(img8, 64, 53, stages=2)  # sharpen-only needs equal sizes


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("ow,seed,mutated", [(160, 21, False), (240, 22, True), (156, 23, True)])
def test_oracle_matches_reference_on_adversarial_patches(ow, seed, mutated):
    """The content the round-4 adversarial search lives on (tools/debug/easu_err_search.py: cancelling gradient directions next to
    the zero guard, extremes beside flats, mutated constant rows / columns) through the REFERENCE's own FsrEasuF / FsrRcasF compiled
    for the CPU: the restatement is bit-identical there too -- the ill-conditioned direction blend is where an operator-order slip
    in the oracle would show first."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("easu_err_search", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "tools", "debug", "easu_err_search.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.P = 8
    rng = np.random.default_rng(seed)
    base = m.fresh(rng)
    g = 15
    patches = [m.fresh(rng) if (not mutated or i % 3 == 0) else m.mutate(rng, m.mutate(rng, base)) for i in range(g * g)]
    img8 = np.empty((8 * g, 8 * g, 4), np.uint8)
    img8[..., 3] = 255
    img8[..., :3] = np.stack(patches).reshape(g, g, 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(8 * g, 8 * g, 3)
    img = O.unorm8_to_float(img8)
    con = O.easu_con(8 * g, 8 * g, ow, ow)
    centre, rad = O.mask_constants(ow, ow, 2.0, (0.5, 0.5, 0.5, 0.5), True, 0)
    a = O.easu(img, ow, ow, con, centre, rad)
    assert same_bits(a, O.ref_easu(img, ow, ow, con, centre, rad))
    q = O.unorm8_to_float(O.float_to_unorm8(a))
    rcon = O.rcas_con(0.9, 0)
    assert same_bits(O.rcas(q, rcon, centre, rad), O.ref_rcas(q, rcon, centre, rad))


# ---- natural content (round 6): the three fixtures of tests/golden/natural_*.npz -------------------------------------------------------
@pytest.mark.parametrize("name", ["cube", "portal", "hopper"])
def test_oracle_matches_natural_golden(name):
    """the oracle's EASU / EASU->RCAS / NVScaler outputs on the natural-content fixtures at 256 -> 341, as SHA-256 digests cut where
    /root/reference was present (tests/golden/make_natural.py): a change of the oracle (or of the fixtures) is a failure here, GPU or not"""
    import json
    import os
    from tests import natural
    gold = json.load(open(os.path.join(natural.GOLD, "natural_golden.json")))
    img = natural.load(name)
    assert img.shape == (256, 256, 4) and img.dtype == np.uint8 and (img[..., 3] == 255).all() and img[..., :3].std() > 40
    assert natural.oracle_digests(img) == gold[name]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("name", ["cube", "portal", "hopper"])
@pytest.mark.parametrize("ow,radius", [(341, 2.0), (341, 0.5), (333, 0.7), (512, 2.0)])
def test_oracle_matches_reference_on_natural_content(name, ow, radius):
    """... and the restatement against the reference's own HLSL text compiled here, bit for bit, on the same natural images: EASU (real
    edges: the dering clamp and the 1/32768 zero-direction branch, ffx_fsr1.h:388-437), RCAS behind the UNORM8 intermediate, NVScaler's edge
    classification (NIS_Scaler.h:176-293)"""
    from tests import natural
    img8 = natural.load(name)
    img = O.unorm8_to_float(img8)
    con = O.easu_con(256, 256, ow, ow)
    centre, rad = O.mask_constants(ow, ow, radius, (0.45, 0.55, 0.6, 0.5), True, 0)
    a = O.easu(img, ow, ow, con, centre, rad)
    assert same_bits(a, O.ref_easu(img, ow, ow, con, centre, rad))
    q = O.unorm8_to_float(O.float_to_unorm8(a))
    rcon = O.rcas_con(0.9, 0)
    assert same_bits(O.rcas(q, rcon, centre, rad), O.ref_rcas(q, rcon, centre, rad))
    cs, cu = O.ref_nis_coefs()
    ok, cfg = O.ref_nis_scaler_config(0.9, 256, 256, ow, ow)
    assert ok
    blk = O.nis_block(cfg, centre, rad, 0)
    assert same_bits(O.nis_upscale(img, ow, ow, blk, cs, cu), O.ref_nis_upscale(img, ow, ow, blk, cs, cu))
    ok, cfg = O.ref_nis_scaler_config(0.9, 256, 256, 256, 256)
    c2, r2 = O.mask_constants(256, 256, radius, (0.45, 0.55, 0.6, 0.5), True, 0)
    blk = O.nis_block(cfg, c2, r2, 0)
    assert same_bits(O.nis_sharpen(img, blk), O.ref_nis_sharpen(img, blk, cs, cu))
