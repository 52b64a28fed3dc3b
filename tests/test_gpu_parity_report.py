"""Measured parity numbers at BASELINE.json's FULL sizes, every config x {strict, product} build, written down.

Each case runs the HIP path through the C ABI and the CPU oracle on the same seeded eye image and records
  max_abs   largest |difference| on float outputs (unit domain)
  max_lsb   largest byte difference on UNORM8 outputs
  n_diff    how many channel values differ at all, of n_total
into gpurun_out/parity_r06.json (merged back from the GPU box; the copy under profiles/ is the committed record).
Content: the synthetic generators of tests/synth.py (structured, uniform-random) and -- round 6 -- NATURAL content: the three fixtures of
tests/golden/natural_*.npz (rendered game art, a rendered UI with text, a photograph), mirror-tiled to the full sizes (tests/natural.py).
The asserts are the stated tolerances:
  strict build   bit-exact everywhere (n_diff == 0)
  product build  float outputs max-abs <= 1e-3 (north_star), measured ~3e-6;
                 UNORM8 outputs of the EASU pass: bit-identical (n_diff == 0) -- the near-tie guard re-resolves every pixel
                 whose re-associated result lies within 2^-9 byte of a rounding boundary in the reference's operator order;
                 UNORM8 outputs of EASU -> UNORM8 -> RCAS and of every other single pass: <= 1 LSB.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import natural, synth
from tests.util import run_gpu

pytestmark = pytest.mark.gpu

STRICT, FP32 = 2, 0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RECORDS = []


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    if not _RECORDS:
        return
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_r06.json"), "w") as f:
        json.dump({"note": "HIP path vs CPU oracle at full BASELINE sizes; written by tests/test_gpu_parity_report.py",
                   "records": _RECORDS}, f, indent=1)


def _rec(config, build, content, output, got, want):
    r = {"config": config, "build": build, "content": content, "output": output, "n_total": int(want.size)}
    if want.dtype == np.uint8:
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        r["max_lsb"] = int(d.max())
        r["n_diff"] = int((d != 0).sum())
        r["n_gt1"] = int((d > 1).sum())
    else:
        g32, w32 = got.astype(np.float32), want.astype(np.float32)
        d = np.abs(g32 - w32)
        r["n_nan"] = int(np.isnan(g32).sum() + np.isnan(w32).sum())
        r["max_abs"] = float(np.max(d))   # NaN propagates: a NaN output fails every "max_abs <=" assert
        r["n_diff"] = int((g32.view(np.uint32) != w32.view(np.uint32)).sum()) if got.dtype == np.float32 else int((got != want).sum())
        r["n_gt_1e-3"] = int((d > 1e-3).sum())
    _RECORDS.append(r)
    # Regression alarms, NOT the contract (the callers assert that): 4-10x the largest values this report has ever held for the
    # product build at full size (profiles/parity_r05.json: FSR float 2.3e-6, NIS float 8.9e-7, differing UNORM8 bytes 2.2e-5 of
    # an image) -- a change that spends more of the tolerance than that should be looked at before it is believed.
    # Natural content (round 6) spends more of it in ONE place: the un-quantised float pipeline behind RCAS reads 2.8e-5 on the UI fixture
    # (EASU alone 6.6e-7 as everywhere: RCAS's 1 / (4 mn - 4) cancels next to white panels, where a contracted 4*mn-4 and the reference's two
    # roundings differ most) -- still 36 times inside the 1e-3 contract; its alarm is set 3.5x above that measurement.
    if build == "product" and not config.startswith("C5"):   # (C5's float distance is its half rounding: 9.8e-4, asserted by the caller)
        if "max_abs" in r:
            assert r["max_abs"] <= (1e-4 if content.startswith("natural") else 1e-5), r
        else:
            assert r["n_diff"] <= 2e-4 * r["n_total"], r
    return r


GEN = {"structured": synth.structured_u8, "random": synth.random_u8, "natural": natural.tiled_u8}


def _nis_want(img8, ow, oh, sharp, radius):
    import openvr_fsr_amd as A
    ih, iw = img8.shape[:2]
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(sharp, iw, ih, ow, oh)
    assert ok
    centre, rad = O.mask_constants(ow, oh, radius)
    return O.nis_upscale(O.unorm8_to_float(img8), ow, oh, O.nis_block(cfg, centre, rad, 0), cs, cu)


@pytest.mark.parametrize("content", ["structured", "random", "natural"])
def test_c1_easu_only(gpu, content):
    """C1: single left eye 1683x1869 -> 2244x2492 RGBA8, EASU only."""
    iw, ih, ow, oh = 1683, 1869, 2244, 2492
    img8 = GEN[content](iw, ih, synth.seed_for(0, 0))
    want = O.easu(O.unorm8_to_float(img8), ow, oh)
    want8 = O.float_to_unorm8(want)
    r = _rec("C1", "strict", content, "float", run_gpu(img8, ow, oh, np.float32, precision=STRICT, stage_mask=1), want)
    assert r["n_diff"] == 0
    r = _rec("C1", "product", content, "float", run_gpu(img8, ow, oh, np.float32, precision=FP32, stage_mask=1), want)
    assert r["max_abs"] <= 2e-5, r
    r = _rec("C1", "product", content, "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=FP32, stage_mask=1), want8)
    assert r["n_diff"] == 0, r   # near-tie guard: the product build's UNORM8 EASU output is the oracle's


@pytest.mark.parametrize("cfg,iw,ih,ow,oh,radius", [("C2", 1683, 1869, 2244, 2492, 2.0), ("C2r", 1683, 1869, 2244, 2492, 0.5),
                                                     ("C4", 2244, 2492, 2916, 3240, 2.0)])
@pytest.mark.parametrize("content", ["structured", "random", "natural"])
def test_fsr_pipeline(gpu, cfg, iw, ih, ow, oh, radius, content):
    """C2 / C2r (shipped radius 0.5) / C4 (renderScale 1.3 shape): EASU -> UNORM8 -> RCAS -> UNORM8, and the same
    pipeline with float intermediate and output (the form north_star's 1e-3 is meaningful on)."""
    img8 = GEN[content](iw, ih, synth.seed_for(0, 1))
    want8 = O.fsr_pipeline_u8(img8, ow, oh, sharpness=0.9, radius=radius, eye=1)
    _, wantf = O.fsr_pipeline_u8(img8, ow, oh, sharpness=0.9, radius=radius, eye=1, quantize_intermediate=False, want_float=True)
    kw = dict(eye=1, sharpness=0.9, radius=radius)
    r = _rec(cfg, "strict", content, "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=STRICT, **kw), want8)
    assert r["n_diff"] == 0
    r = _rec(cfg, "strict", content, "float (float intermediate)",
             run_gpu(img8, ow, oh, np.float32, precision=STRICT, quantize_intermediate=0, fused=0, **kw), wantf)
    assert r["n_diff"] == 0
    r = _rec(cfg, "product", content, "float (float intermediate)",
             run_gpu(img8, ow, oh, np.float32, precision=FP32, quantize_intermediate=0, fused=0, **kw), wantf)
    assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
    r = _rec(cfg, "product", content, "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=FP32, **kw), want8)
    assert r["max_lsb"] <= 1, r
    # the 8-bit intermediate itself (what rounds 1-2 got wrong in ~12 bytes per million, amplified up to 4x by RCAS)
    if radius >= 2.0:
        mid_want = O.float_to_unorm8(O.easu(O.unorm8_to_float(img8), ow, oh))
        r = _rec(cfg, "product", content, "unorm8 intermediate (EASU pass)",
                 run_gpu(img8, ow, oh, np.uint8, precision=FP32, stage_mask=1, **kw), mid_want)
        assert r["n_diff"] == 0, r


@pytest.mark.parametrize("cfg,radius", [("C3", 2.0), ("C3r", 0.5)])
@pytest.mark.parametrize("content", ["structured", "random", "natural"])
def test_nis_scaler(gpu, cfg, radius, content):
    """C3 / C3r: NVScaler 1683x1869 -> 2244x2492."""
    iw, ih, ow, oh = 1683, 1869, 2244, 2492
    img8 = GEN[content](iw, ih, synth.seed_for(1, 0))
    want = _nis_want(img8, ow, oh, 0.9, radius)
    want8 = O.float_to_unorm8(want)
    kw = dict(use_nis=1, sharpness=0.9, radius=radius)
    r = _rec(cfg, "strict", content, "float", run_gpu(img8, ow, oh, np.float32, precision=STRICT, **kw), want)
    assert r["n_diff"] == 0
    r = _rec(cfg, "product", content, "float", run_gpu(img8, ow, oh, np.float32, precision=FP32, **kw), want)
    assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
    r = _rec(cfg, "product", content, "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=FP32, **kw), want8)
    assert r["max_lsb"] <= 1, r


def _c5_case(gen, seed, scale=1.0):
    iw, ih, ow, oh = 2370, 2370, 3160, 3160
    imgh = (gen(iw, ih, seed).astype(np.float32) * (scale / 255.0)).astype(np.float16)
    imgh[..., 3] = np.float16(1.0)
    centre, rad = O.mask_constants(ow, oh, 0.5)
    e = O.easu(imgh.astype(np.float32), ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
    e16 = e.astype(np.float16).astype(np.float32)
    return imgh, O.rcas(e16, O.rcas_con(0.9), centre, rad).astype(np.float16)


@pytest.mark.parametrize("content,seed", [("structured", 77), ("structured", 78), ("random", 79)])
def test_c5_masked_half(gpu, content, seed):
    """C5: radius-masked EASU+RCAS, 2370x2370 -> 3160x3160, RGBA16F in, half intermediate, RGBA16F out (two structured images and
    a uniform-random one)."""
    ow, oh = 3160, 3160
    imgh, want = _c5_case(GEN[content], seed)
    tag = "%s (seed %d)" % (content, seed)
    r = _rec("C5", "strict", tag, "half", run_gpu(imgh, ow, oh, np.float16, precision=STRICT, sharpness=0.9, radius=0.5), want)
    assert r["n_diff"] == 0
    r = _rec("C5", "product", tag, "half", run_gpu(imgh, ow, oh, np.float16, precision=FP32, sharpness=0.9, radius=0.5), want)
    # the half intermediate's near-ties are re-resolved in the reference's operator order wherever a flipped half-ulp could
    # exceed the tolerance behind RCAS's gain (near_tie_half3): north_star's bound holds on every value
    assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (1, 1.0), (2, 1.0), (0, 6.0), (2, 6.0)])
def test_c5_masked_half_natural(gpu, seed, scale):
    """C5 on natural content (round 6): the three fixtures as RGBA16F at unit range, and two of them as HDR -- half(b / 255 * 6): highlights
    six times the unit range next to dark texels, what the half guard's footprint-scaled band exists for.  Strict bit-exact; product within
    1e-3 on every value of the unit-range images, within max(1e-3, one half spacing of the value) on the HDR ones (header contract)."""
    ow, oh = 3160, 3160
    imgh, want = _c5_case(natural.tiled_u8, seed, scale)
    tag = "natural (%s%s)" % (natural.NAMES[seed % 3], "" if scale == 1.0 else " x%g HDR" % scale)
    r = _rec("C5", "strict", tag, "half", run_gpu(imgh, ow, oh, np.float16, precision=STRICT, sharpness=0.9, radius=0.5), want)
    assert r["n_diff"] == 0
    got = run_gpu(imgh, ow, oh, np.float16, precision=FP32, sharpness=0.9, radius=0.5)
    r = _rec("C5", "product", tag, "half", got, want)
    if scale == 1.0:
        assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
    else:
        w32, g32 = want.astype(np.float32), got.astype(np.float32)
        spacing = np.maximum(np.exp2(np.floor(np.log2(np.maximum(np.abs(w32), 2.0 ** -14)))) * 2.0 ** -10, 0.0)
        over = np.abs(g32 - w32) > np.maximum(1e-3, spacing) * 1.0001
        r["n_beyond_contract"] = int(over.sum())
        assert r["n_beyond_contract"] == 0, r


@pytest.mark.parametrize("cfg", ["C2", "C2r", "C5", "C2sbsr"])
def test_timed_call_full_size(gpu, cfg):
    """The call bench.py times -- ovrfsr_apply_batch over a batch of full-size images (L,R,L,R..., blockIdx.z > 0, XCD-ordered
    tile indices, per-eye mask lists) -- product build, EVERY image against the oracle.  (The other records of this file go
    through single-image ovrfsr_apply.)  C2sbsr: the shared side-by-side form, ovrfsr_apply_batch_shared."""
    import torch
    import openvr_fsr_amd as A
    n = 6
    shared = cfg == "C2sbsr"
    iw, ih, ow, oh, radius = {"C2": (1683, 1869, 2244, 2492, 2.0), "C2r": (1683, 1869, 2244, 2492, 0.5),
                              "C5": (2370, 2370, 3160, 3160, 0.5), "C2sbsr": (3366, 1869, 4488, 2492, 0.5)}[cfg]
    if shared:
        n = 3
    half = cfg == "C5"
    gens = [synth.structured_u8, natural.tiled_u8, synth.random_u8]
    imgs8 = [gens[i % 3](iw, ih, synth.seed_for(10 + i // 2, i & 1)) for i in range(n)]
    if half:
        src = np.stack([(im.astype(np.float32) / 255.0).astype(np.float16) for im in imgs8])
    else:
        src = np.stack(imgs8)
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9, radius=radius, precision=FP32)
    t = torch.from_numpy(src).cuda()
    outs = torch.empty((n, oh, ow, 4), dtype=t.dtype, device="cuda")
    pp.apply_batch(t, outs, first_eye=A.EYE_LEFT, alternate_eyes=True, shared=shared)
    torch.cuda.synchronize()
    got = outs.cpu().numpy()
    pp.close()
    centre_rad = None
    for i in range(n):
        eye = 0 if shared else i & 1
        if half:
            centre, rad = O.mask_constants(ow, oh, radius, (0.5,) * 4, True, eye)
            e = O.easu(src[i].astype(np.float32), ow, oh, O.easu_con(iw, ih, ow, oh), centre, rad)
            want = O.rcas(e.astype(np.float16).astype(np.float32), O.rcas_con(0.9), centre, rad).astype(np.float16)
        else:
            want = O.fsr_pipeline_u8(imgs8[i], ow, oh, sharpness=0.9, radius=radius, eye=eye, one_eye_per_texture=not shared)
        r = _rec(cfg + " batched (image %d of %d)" % (i, n), "product", ("structured", "natural", "random")[i % 3],
                 "half" if half else "unorm8", got[i], want)
        if half:
            assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
        else:
            assert r["max_lsb"] <= 1, r


@pytest.mark.parametrize("content", ["structured", "random", "natural"])
def test_sharpen_only_configs(gpu, content):
    """renderScale 1 at 2244x2492 (PostProcessor.cpp:586-594: no upscale stage): RCAS alone (C2s, the DPP kernel) and NVSharpen
    alone (C3s)."""
    import openvr_fsr_amd as A
    w, h = 2244, 2492
    img8 = GEN[content](w, h, synth.seed_for(2, 0))
    centre, rad = O.mask_constants(w, h)
    want = O.rcas(O.unorm8_to_float(img8), O.rcas_con(0.9), centre, rad)
    kw = dict(render_scale=1.0, sharpness=0.9)
    r = _rec("C2s", "strict", content, "float", run_gpu(img8, w, h, np.float32, precision=STRICT, **kw), want)
    assert r["n_diff"] == 0
    r = _rec("C2s", "product", content, "float", run_gpu(img8, w, h, np.float32, precision=FP32, **kw), want)
    assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
    r = _rec("C2s", "product", content, "unorm8", run_gpu(img8, w, h, np.uint8, precision=FP32, **kw), O.float_to_unorm8(want))
    assert r["max_lsb"] <= 1, r
    ok, cfg = A.nis_sharpen_config(0.9, w, h)
    wantn = O.nis_sharpen(O.unorm8_to_float(img8), O.nis_block(cfg, centre, rad, 0))
    kw = dict(use_nis=1, render_scale=1.0, sharpness=0.9)
    r = _rec("C3s", "strict", content, "float", run_gpu(img8, w, h, np.float32, precision=STRICT, **kw), wantn)
    assert r["n_diff"] == 0
    r = _rec("C3s", "product", content, "float", run_gpu(img8, w, h, np.float32, precision=FP32, **kw), wantn)
    assert r["max_abs"] <= 1e-3 and r["n_gt_1e-3"] == 0, r
    r = _rec("C3s", "product", content, "unorm8", run_gpu(img8, w, h, np.uint8, precision=FP32, **kw), O.float_to_unorm8(wantn))
    assert r["max_lsb"] <= 1, r


def test_c2_other_settings(gpu):
    """C2's shape with the knobs a user turns: sharpness 0 and 1, an off-centre projection with a mask, the right eye, the debug
    tint: strict bit-exact, product <= 1 LSB."""
    iw, ih, ow, oh = 1683, 1869, 2244, 2492
    img8 = synth.structured_u8(iw, ih, synth.seed_for(3, 1))
    for name, kw in (("sharpness 0", dict(sharpness=0.0, radius=2.0)), ("sharpness 1", dict(sharpness=1.0, radius=2.0)),
                     ("radius 0.7 off-centre debug", dict(sharpness=0.75, radius=0.7, proj=(0.42, 0.55, 0.61, 0.47), debug=1))):
        okw = dict(sharpness=kw["sharpness"], radius=kw["radius"], eye=1, proj=kw.get("proj", (0.5,) * 4), debug=kw.get("debug", 0))
        want8 = O.fsr_pipeline_u8(img8, ow, oh, **okw)
        gkw = dict(eye=1, sharpness=kw["sharpness"], radius=kw["radius"], proj_centre=kw.get("proj", (0.5,) * 4), debug_mode=kw.get("debug", 0))
        r = _rec("C2 (" + name + ")", "strict", "structured", "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=STRICT, **gkw), want8)
        assert r["n_diff"] == 0
        r = _rec("C2 (" + name + ")", "product", "structured", "unorm8", run_gpu(img8, ow, oh, np.uint8, precision=FP32, **gkw), want8)
        assert r["max_lsb"] <= 1, r
