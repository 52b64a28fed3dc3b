"""The C-ABI library loads and exports every symbol include/*.h declares; host-side entry points behave
like the reference's host code.  No GPU needed (no compute call is made)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import openvr_fsr_amd as A
from openvr_fsr_amd import _capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "openvr_fsr_amd.h")).read()
    return sorted(set(re.findall(r"OVRFSR_API\s+[\w\s\*]+?\b(ovrfsr_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(A.library_path())
    names = declared_symbols()
    assert len(names) >= 18, names
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_abi_version_and_struct_sizes():
    L = A.library()
    assert L.ovrfsr_abi_version() == 5
    cfg = A.Config.default()
    assert cfg.struct_size == C.sizeof(A.Config) == 80
    assert C.sizeof(A.Image) == 24 and C.sizeof(A.Bounds) == 16


def test_config_defaults_match_reference_config_h():
    cfg = A.Config.default()   # src/postprocess/Config.h:10-17
    assert cfg.fsr_enabled == 0 and cfg.use_nis == 0 and cfg.debug_mode == 0
    assert cfg.render_scale == 1.0 and cfg.sharpness == 0.75 and cfg.radius == 0.5
    assert list(cfg.proj_centre) == [0.5] * 4 and cfg.quantize_intermediate == 1


def test_output_size_truncation_rule():
    # PostProcessor.cpp:512-518: uint32 <- float truncation; <1 divides, >=1 multiplies
    assert A.output_size(A.Config.default(render_scale=0.75), 1683, 1869) == (2244, 2492)
    assert A.output_size(A.Config.default(render_scale=1.3), 2244, 2492) == (int(np.float32(2244) * np.float32(1.3)), int(np.float32(2492) * np.float32(1.3)))
    assert A.output_size(A.Config.default(render_scale=1.3), 2244, 2492) == (2917, 3239)   # not 2916x3240
    assert A.output_size(A.Config.default(render_scale=0.77), 1000, 1000) == (int(np.float32(1000) / np.float32(0.77)),) * 2
    assert A.output_size(A.Config.default(render_scale=0.5, out_width=123, out_height=45), 10, 10) == (123, 45)


@pytest.mark.parametrize("scale", [0.0, -1.0, float("nan"), float("inf"), 1e-6, 1e9])
def test_output_size_rejects_undefined_and_unbounded_scales(scale):
    """A zero / negative / non-finite renderScale has no defined uint conversion and a tiny (or huge) one asks for an
    unbounded image: INVALID_ARGUMENT, nothing written, nothing thrown across the C boundary."""
    cfg = A.Config.default(render_scale=scale)
    w, h = C.c_uint32(7), C.c_uint32(7)
    assert A.library().ovrfsr_output_size(C.byref(cfg), 1683, 1869, C.byref(w), C.byref(h)) == 1
    assert (w.value, h.value) == (7, 7)


def test_output_size_is_capped_at_the_image_limit():
    w, h = C.c_uint32(), C.c_uint32()
    cfg = A.Config.default(render_scale=0.5)
    assert A.library().ovrfsr_output_size(C.byref(cfg), 8192, 8192, C.byref(w), C.byref(h)) == 0 and w.value == 16384
    assert A.library().ovrfsr_output_size(C.byref(cfg), 8193, 100, C.byref(w), C.byref(h)) == 1
    cfg = A.Config.default(out_width=16385, out_height=10)
    assert A.library().ovrfsr_output_size(C.byref(cfg), 10, 10, C.byref(w), C.byref(h)) == 1


def test_bad_struct_size_is_rejected():
    cfg = A.Config.default()
    cfg.struct_size = 12
    w, h = C.c_uint32(), C.c_uint32()
    assert A.library().ovrfsr_output_size(C.byref(cfg), 10, 10, C.byref(w), C.byref(h)) == 1
    ctx = C.c_void_p()
    assert A.library().ovrfsr_create(0, C.byref(cfg), C.byref(ctx)) == 1 and not ctx


def test_create_rejects_configurations_no_kernel_exists_for():
    """precision 1 (a packed-half mode ABI 1 reserved and never built), unknown stage masks / fused values: rejected at
    create with INVALID_ARGUMENT before any device is touched, instead of a ctx that disables itself at the first apply."""
    ctx = C.c_void_p()
    for kw in (dict(precision=1), dict(precision=7), dict(stage_mask=3), dict(fused=2), dict(pair_submit=2), dict(pair_submit=-1)):
        cfg = A.Config.default(fsr_enabled=1, **kw)
        assert A.library().ovrfsr_create(0, C.byref(cfg), C.byref(ctx)) == 1 and not ctx
    assert not hasattr(A, "PRECISION_FP16")
    hdr = open(os.path.join(ROOT, "include", "openvr_fsr_amd.h")).read()
    assert "OVRFSR_PRECISION_FP16" not in hdr.split("typedef enum ovrfsr_precision")[1].split("}")[0]


@pytest.mark.parametrize("kw", [dict(radius=-1.0), dict(radius=float("nan")), dict(radius=float("inf")), dict(radius=-0.001),
                                dict(proj_centre=(float("nan"), 0.5, 0.5, 0.5)), dict(proj_centre=(0.5, -1.5, 0.5, 0.5)), dict(proj_centre=(0.5, 0.5, 1e9, 0.5)),
                                dict(proj_centre=(0.5, 0.5, 0.5, float("inf"))), dict(sharpness=float("nan")), dict(sharpness=float("-inf")),
                                dict(render_scale=float("nan"))])
def test_create_rejects_values_the_mask_conversion_is_undefined_for(kw):
    """Round 6: radius / proj_centre feed float -> uint32 conversions (PostProcessor.cpp:298-305) that are undefined
    behaviour for NaN, negative or huge values; the reference validates nothing (Config.h:36-45 clamps a negative sharpness only).  A ctx
    is never built around such a value, and set_config keeps the old configuration."""
    ctx = C.c_void_p()
    cfg = A.Config.default(fsr_enabled=1, **kw)
    assert A.library().ovrfsr_create(0, C.byref(cfg), C.byref(ctx)) == 1 and not ctx


@pytest.mark.parametrize("kw", [dict(radius=0.0), dict(radius=1e9), dict(radius=2.0), dict(proj_centre=(-0.5, -1.0, 2.0, 1.5)), dict(sharpness=-3.0),
                                dict(sharpness=7.0)])
def test_create_accepts_every_value_with_a_defined_meaning(kw):
    """... and nothing more is refused than that: radius 0 (everything outside), a huge finite radius (the conversion saturates), centres
    off the image (clamped to 0 by the conversion), any finite sharpness (clamped to [0,1] like the reference's AClampF1): create gets past the
    validation -- without a GPU it then reports NO_DEVICE, never INVALID_ARGUMENT."""
    import torch
    ctx = C.c_void_p()
    cfg = A.Config.default(fsr_enabled=1, **kw)
    rc = A.library().ovrfsr_create(0, C.byref(cfg), C.byref(ctx))
    assert rc == (0 if torch.cuda.is_available() else 4), rc
    if ctx:
        A.library().ovrfsr_destroy(ctx)


def test_mask_constants_is_total_and_keeps_the_known_answers():
    """ovrfsr_mask_constants on the probes of VERDICT r5 (radius -1 / NaN / 1e9, proj NaN / -0.5 / 1e9): saturating conversion -- NaN and
    negatives give 0, >= 2^32 gives 0xffffffff -- the same in the product and in the oracle; in-range values keep SURVEY.md 8c's answers."""
    from oracle import oracle as O
    half = (0.5, 0.5, 0.5, 0.5)
    # known answers: radius 0.5 -> 623 @2492, 790 @3160
    c, r = A.mask_constants(2244, 2492, half, 0.5, True, 0)
    assert list(c) == [1122, 1246, 1122, 1246] and list(r) == [623, 388129, 2244, 2492]
    c, r = A.mask_constants(3160, 3160, half, 0.5, True, 1)
    assert list(r) == [790, 624100, 3160, 3160]
    probes = [(-1.0, half), (float("nan"), half), (1e9, half), (float("inf"), half), (0.5, (float("nan"), 0.5, 0.5, 0.5)),
              (0.5, (-0.5, 0.5, -0.5, 0.5)), (0.5, (1e9, 0.5, 1e9, 0.5)), (0.5, (float("-inf"), float("inf"), 0.5, 0.5)), (-0.0, half), (1e30, half)]
    for radius, proj in probes:
        for one_eye in (True, False):
            for eye in (0, 1):
                c, r = A.mask_constants(2244, 2492, proj, radius, one_eye, eye)
                co, ro = O.mask_constants(2244, 2492, radius, proj, one_eye, eye)
                assert list(c) == list(co) and list(r) == list(ro), (radius, proj, one_eye, eye)
    c, r = A.mask_constants(2244, 2492, half, -1.0, True, 0)
    assert r[0] == 0 and r[1] == 0
    c, r = A.mask_constants(2244, 2492, half, float("nan"), True, 0)
    assert r[0] == 0
    c, r = A.mask_constants(2244, 2492, half, 1e9, True, 0)
    assert r[0] == 0xffffffff and r[1] == 1                      # uint32 wrap of r * r, as the reference's multiply
    c, r = A.mask_constants(2244, 2492, (float("nan"), -0.5, 1e9, 0.5), 0.5, True, 0)
    assert list(c) == [0, 0, 0, 0]
    c, r = A.mask_constants(2244, 2492, (float("nan"), -0.5, 1e9, 0.5), 0.5, True, 1)
    assert list(c) == [0xffffffff, 1246, 0xffffffff, 1246]


def test_config_file_never_yields_a_cfg_that_create_refuses():
    """ovrfsr_config_from_json: a negative radius is clamped to 0 (the reference's rule for sharpness, Config.h:40), a number that
    overflows float is a read error (defaults); what it returns always passes create's validation."""
    rc, cfg = A.config_from_json('{"fsr": {"enabled": true, "radius": -1, "sharpness": -2}}')
    assert rc == 0 and cfg.radius == 0.0 and cfg.sharpness == 0.0 and cfg.fsr_enabled == 1
    for text in ('{"fsr": {"enabled": true, "radius": 1e999}}', '{"fsr": {"sharpness": 1e999}}', '{"fsr": {"enabled": true, "renderScale": 1e400}}'):
        rc, cfg = A.config_from_json(text)
        assert rc == 1 and cfg.fsr_enabled == 0 and cfg.radius == 0.5 and cfg.sharpness == 0.75 and cfg.render_scale == 1.0, text
    rc, cfg = A.config_from_json('{"fsr": {"sharpness": -1e999}}')   # -inf < 0: clamped to 0 by the reference's own rule
    assert rc == 0 and cfg.sharpness == 0.0


def test_capture_writers_validate_the_image_before_touching_it(tmp_path):
    """ovrfsr_save_ppm / ovrfsr_save_dds apply the rules ovrfsr_apply puts on caller images (size, pitch >= width * texel, alignment) before
    anything is copied: a too-small pitch used to make the row loop read past the host copy.  No device is touched."""
    L = A.library()
    path = str(tmp_path / "x.bin").encode()
    fake = 0x7f0000001000   # never dereferenced: every case is refused first
    bad = [K.Image(fake, 64, 8, 64 * 4 - 4, K.FORMAT_RGBA8), K.Image(fake, 64, 8, 64 * 8 - 8, K.FORMAT_RGBA16F), K.Image(fake, 0, 8, 256, K.FORMAT_RGBA8),
           K.Image(fake, 64, 0, 256, K.FORMAT_RGBA8), K.Image(fake, 16385, 8, 16385 * 4, K.FORMAT_RGBA8), K.Image(fake, 8, 16385, 32, K.FORMAT_RGBA8),
           K.Image(fake + 2, 64, 8, 256, K.FORMAT_RGBA8), K.Image(fake, 64, 8, 258, K.FORMAT_RGBA8), K.Image(fake, 64, 8, 256, 9), K.Image(0, 64, 8, 256, K.FORMAT_RGBA8)]
    for img in bad:
        assert L.ovrfsr_save_dds(C.byref(img), path, None) == 1
        assert L.ovrfsr_save_ppm(C.byref(img), path, None) == 1
    assert not os.path.exists(path.decode())


def test_null_ctx_is_an_error_not_a_crash():
    L = A.library()
    assert L.ovrfsr_reset(None) == 1
    assert L.ovrfsr_apply(None, 0, None, None, None, None) == 1
    assert L.ovrfsr_last_error(None) == b"null ctx"
    L.ovrfsr_destroy(None)


def test_no_entry_point_dereferences_a_null_it_was_handed():
    """every pointer argument of every entry point that can be called without a device, as NULL: a status (or nothing), never a crash --
    in a subprocess, so that a crash is a test failure and not the end of the test run"""
    import subprocess
    import sys
    code = """
import ctypes as C, sys
sys.path.insert(0, %r)
import openvr_fsr_amd as A
L = A.library()
cfg = A.Config.default(fsr_enabled=1)
w, h = C.c_uint32(), C.c_uint32()
L.ovrfsr_config_default(None)
assert L.ovrfsr_output_size(None, 8, 8, C.byref(w), C.byref(h)) == 1
assert L.ovrfsr_output_size(C.byref(cfg), 8, 8, None, C.byref(h)) == 1 and L.ovrfsr_output_size(C.byref(cfg), 8, 8, C.byref(w), None) == 1
assert L.ovrfsr_create(0, None, None) == 1 and L.ovrfsr_create(0, C.byref(cfg), None) == 1
ctx = C.c_void_p()
assert L.ovrfsr_create(0, None, C.byref(ctx)) == 1
for f in ("ovrfsr_set_config", "ovrfsr_get_config"):
    assert getattr(L, f)(None, None) == 1 and getattr(L, f)(None, C.byref(cfg)) == 1
assert L.ovrfsr_apply_batch(None, 1, 0, 1, None, 0, None, 0, None) == 1 and L.ovrfsr_apply_batch_shared(None, 1, None, 0, None, 0, None) == 1
assert L.ovrfsr_last_gpu_time_ms(None, None) == 1 and L.ovrfsr_average_gpu_time_ms(None, None, None) == 1
assert L.ovrfsr_pair_pending(None) == 0
L.ovrfsr_easu_con.argtypes = [C.c_void_p] + [C.c_float] * 6; L.ovrfsr_easu_con.restype = None
L.ovrfsr_easu_con(None, 1, 1, 1, 1, 2, 2)
L.ovrfsr_rcas_con.argtypes = [C.c_void_p, C.c_float]; L.ovrfsr_rcas_con.restype = None
L.ovrfsr_rcas_con(None, 0.2)
L.ovrfsr_mask_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float, C.c_int, C.c_int]; L.ovrfsr_mask_constants.restype = None
four = (C.c_uint32 * 4)(); proj = (C.c_float * 4)(0.5, 0.5, 0.5, 0.5)
L.ovrfsr_mask_constants(None, four, 64, 64, proj, 0.5, 1, 0); L.ovrfsr_mask_constants(four, None, 64, 64, proj, 0.5, 1, 0); L.ovrfsr_mask_constants(four, four, 64, 64, None, 0.5, 1, 0)
L.ovrfsr_nis_scaler_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 4
L.ovrfsr_nis_sharpen_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 2
assert L.ovrfsr_nis_scaler_config(None, 0.5, 8, 8, 10, 10) == 0 and L.ovrfsr_nis_sharpen_config(None, 0.5, 8, 8) == 0
L.ovrfsr_config_from_json.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
assert L.ovrfsr_config_from_json(None, 5, C.byref(cfg)) == 1 and L.ovrfsr_config_from_json(b"{}", 2, None) == 1
L.ovrfsr_save_ppm.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]; L.ovrfsr_save_dds.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
assert L.ovrfsr_save_ppm(None, b"/tmp/x.ppm", None) == 1 and L.ovrfsr_save_dds(None, b"/tmp/x.dds", None) == 1
img = A.Image(); 
assert L.ovrfsr_save_ppm(C.byref(img), None, None) == 1 and L.ovrfsr_save_dds(C.byref(img), None, None) == 1
print("survived")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    cfg = A.Config.default(fsr_enabled=1)
    assert A.library().ovrfsr_create(0, C.byref(cfg), C.byref(ctx)) == 4   # OVRFSR_ERR_NO_DEVICE: no CPU fallback
    with pytest.raises(K.OvrFsrError):
        A.PostProcessor(fsr_enabled=1)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing in the product package or its shared library may import, link or call
    it, and outside tests/ only bench.py's checker legs -- oracle_expected() / parity_and_cpu(): the parity_check of the timed
    call's outputs and the cpu_baseline sample, never the timed region -- and __graft_entry__.smoke() may."""
    import ast
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_importers(path):
        """names of the functions (or '<module>') of a python file that import the oracle package"""
        tree = ast.parse(open(path).read())
        hits = []
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            body = fn.body if isinstance(fn, ast.Module) else fn.body
            for node in body if isinstance(fn, ast.Module) else ast.walk(fn):
                if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                    hits.append(getattr(fn, "name", "<module>"))
                if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                    hits.append(getattr(fn, "name", "<module>"))
        return sorted(set(hits))

    allowed = {"bench.py": ["oracle_expected", "parity_and_cpu"], "__graft_entry__.py": ["smoke"]}
    for dirpath, dirs, files in os.walk(root):
        dirs[:] = [d for d in dirs if d not in (".git", "tests", "oracle", "gpurun_out", "ab", "__pycache__", "build")]
        for f in files:
            if f.endswith(".py"):
                rel = os.path.relpath(os.path.join(dirpath, f), root)
                assert oracle_importers(os.path.join(dirpath, f)) == allowed.get(rel, []), rel
    lib = os.path.join(root, "openvr_fsr_amd", "libopenvr_fsr_amd.so")
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "oracle" not in needed
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "ovo_" not in syms and "ref_" not in syms.replace("_ref_count", "")
    for src in os.listdir(os.path.join(root, "openvr_fsr_amd", "csrc")):
        if src.endswith((".cpp", ".hpp", ".h", ".hip", ".inc")):
            text = open(os.path.join(root, "openvr_fsr_amd", "csrc", src)).read()
            assert '../../oracle' not in text and 'liboracle' not in text, src
