"""ctypes/numpy front-end for oracle/liboracle.so (the C restatement) and, when present,
oracle/_ref/libovrfsr_ref.so (the reference's own code compiled through oracle/hlsl_shim.hpp).

TEST INFRASTRUCTURE ONLY -- see oracle/fsr_oracle.c for the contract and citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("fsr_oracle.c", "nis_oracle.c") if os.path.exists(os.path.join(HERE, f))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        try:
            _LIB = C.CDLL(build())
        except OSError:
            _LIB = C.CDLL(build(force=True))
        L = _LIB
        L.ovo_easu_con.argtypes = [u32p] + [C.c_float] * 6
        L.ovo_rcas_con.argtypes = [u32p, C.c_float]
        L.ovo_f32_to_f16_trunc.argtypes = [C.c_float]
        L.ovo_f32_to_f16_trunc.restype = C.c_uint32
        L.ovo_rcas_stops_from_sharpness.argtypes = [C.c_float]
        L.ovo_rcas_stops_from_sharpness.restype = C.c_float
        L.ovo_mask_constants.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32, f32p, C.c_float, C.c_int, C.c_int]
        L.ovo_unorm8_to_float.argtypes = [u8p, C.c_size_t, f32p]
        L.ovo_float_to_unorm8.argtypes = [f32p, C.c_size_t, u8p]
        L.ovo_easu.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, u32p, u32p, u32p, C.c_int]
        L.ovo_rcas.argtypes = [f32p, C.c_int, C.c_int, f32p, u32p, u32p, u32p, C.c_int]
        L.ovo_fsr_pipeline_u8.argtypes = [u8p, C.c_int, C.c_int, u8p, f32p, C.c_int, C.c_int, u32p, u32p,
                                          u32p, u32p, C.c_int, C.c_int, C.c_int]
        L.ovo_fsr_pipeline_u8.restype = C.c_int
        L.ovo_max_threads.restype = C.c_int
    return _LIB


_PROBE = None


def shim_probe():
    """oracle/libshimprobe.so: the D3D11 fixed-function behaviours of hlsl_shim.hpp, one entry point each."""
    global _PROBE
    if _PROBE is None:
        so = os.path.join(HERE, "libshimprobe.so")
        srcs = [os.path.join(HERE, f) for f in ("shim_probe.cpp", "hlsl_shim.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
            subprocess.check_call(["make", "-C", HERE, "-B", "libshimprobe.so"], stdout=subprocess.DEVNULL)
        P = C.CDLL(so)
        P.probe_gather.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, f32p]
        P.probe_sample.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
        P.probe_load.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        P.probe_unorm_store.argtypes = [f32p, f32p]
        _PROBE = P
    return _PROBE


def ref_path():
    return os.path.join(HERE, "_ref", "libovrfsr_ref.so")


def have_ref():
    return os.path.exists(ref_path())


def ref():
    """The reference's own code (oracle/_ref).  Built by oracle/build_ref.py where /root/reference exists."""
    global _REF
    if _REF is None:
        R = C.CDLL(ref_path())
        R.ref_easu_con.argtypes = [u32p] + [C.c_float] * 6
        R.ref_rcas_con.argtypes = [u32p, C.c_float]
        R.ref_f32_to_f16.argtypes = [C.c_float]
        R.ref_f32_to_f16.restype = C.c_uint32
        R.ref_clamp_f1.argtypes = [C.c_float] * 3
        R.ref_clamp_f1.restype = C.c_float
        R.ref_easu_dispatch.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, u32p]
        R.ref_easu_pixel.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, u32p, f32p]
        R.ref_rcas_dispatch.argtypes = [f32p, C.c_int, C.c_int, f32p, u32p]
        R.ref_rmp8x8.argtypes = [C.c_uint32, u32p]
        R.ref_nis_config_size.restype = C.c_int
        R.ref_nis_scaler_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 4
        R.ref_nis_scaler_config.restype = C.c_int
        R.ref_nis_sharpen_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 2
        R.ref_nis_sharpen_config.restype = C.c_int
        R.ref_nis_coefs.argtypes = [f32p, f32p]
        _REF = R
    return _REF


# ---------------------------------------------------------------------------------------------
# constants
# ---------------------------------------------------------------------------------------------
def easu_con(inW, inH, outW, outH):
    """uint32[16] = con0..con3 (PostProcessor.cpp:297: viewport == input size)."""
    con = np.zeros(16, np.uint32)
    lib().ovo_easu_con(_ptr(con, u32p), inW, inH, inW, inH, outW, outH)
    return con


def rcas_con(sharpness, debug=0):
    """uint32[4]: FsrRcasCon(2-2*clamp(sharpness)) with const0[3]=debugMode (PostProcessor.cpp:420-430)."""
    con = np.zeros(4, np.uint32)
    stops = lib().ovo_rcas_stops_from_sharpness(sharpness)
    lib().ovo_rcas_con(_ptr(con, u32p), stops)
    con[3] = int(debug)
    return con


def mask_constants(outW, outH, radius=2.0, proj=(0.5, 0.5, 0.5, 0.5), one_eye_per_texture=True, eye=0):
    centre = np.zeros(4, np.uint32)
    rad = np.zeros(4, np.uint32)
    p = np.asarray(proj, np.float32)
    lib().ovo_mask_constants(_ptr(centre, u32p), _ptr(rad, u32p), outW, outH, _ptr(p, f32p), radius,
                             int(one_eye_per_texture), int(eye))
    return centre, rad


# ---------------------------------------------------------------------------------------------
# image helpers
# ---------------------------------------------------------------------------------------------
def unorm8_to_float(a8):
    a8 = np.ascontiguousarray(a8, np.uint8)
    out = np.empty(a8.shape, np.float32)
    lib().ovo_unorm8_to_float(_ptr(a8, u8p), a8.size, _ptr(out, f32p))
    return out


def float_to_unorm8(af):
    af = np.ascontiguousarray(af, np.float32)
    out = np.empty(af.shape, np.uint8)
    lib().ovo_float_to_unorm8(_ptr(af, f32p), af.size, _ptr(out, u8p))
    return out


def easu(img, outW, outH, con=None, centre=None, radius=None, nthreads=0):
    """img: float32 [inH, inW, 4] -> float32 [outH, outW, 4] (alpha = 1)."""
    img = np.ascontiguousarray(img, np.float32)
    inH, inW = img.shape[:2]
    if con is None:
        con = easu_con(inW, inH, outW, outH)
    if centre is None:
        centre, radius = mask_constants(outW, outH)
    out = np.empty((outH, outW, 4), np.float32)
    nt = nthreads or lib().ovo_max_threads()
    lib().ovo_easu(_ptr(img, f32p), inW, inH, _ptr(out, f32p), outW, outH, _ptr(con, u32p),
                   _ptr(np.ascontiguousarray(centre, np.uint32), u32p),
                   _ptr(np.ascontiguousarray(radius, np.uint32), u32p), nt)
    return out


def rcas(img, con, centre=None, radius=None, nthreads=0):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    if centre is None:
        centre, radius = mask_constants(W, H)
    out = np.empty((H, W, 4), np.float32)
    nt = nthreads or lib().ovo_max_threads()
    lib().ovo_rcas(_ptr(img, f32p), W, H, _ptr(out, f32p), _ptr(np.ascontiguousarray(con, np.uint32), u32p),
                   _ptr(np.ascontiguousarray(centre, np.uint32), u32p),
                   _ptr(np.ascontiguousarray(radius, np.uint32), u32p), nt)
    return out


def fsr_pipeline_u8(img8, outW, outH, sharpness=0.9, radius=2.0, proj=(0.5, 0.5, 0.5, 0.5), eye=0,
                    one_eye_per_texture=True, debug=0, stages=3, quantize_intermediate=True, nthreads=0,
                    want_float=False):
    """UNORM8 [inH,inW,4] -> UNORM8 [outH,outW,4] exactly as ApplyPostProcess chains the two passes."""
    img8 = np.ascontiguousarray(img8, np.uint8)
    inH, inW = img8.shape[:2]
    econ = easu_con(inW, inH, outW, outH)
    rcon = rcas_con(sharpness, debug)
    centre, rad = mask_constants(outW, outH, radius, proj, one_eye_per_texture, eye)
    out8 = np.empty((outH, outW, 4), np.uint8)
    outf = np.empty((outH, outW, 4), np.float32) if want_float else None
    nt = nthreads or lib().ovo_max_threads()
    rc = lib().ovo_fsr_pipeline_u8(_ptr(img8, u8p), inW, inH, _ptr(out8, u8p),
                                   _ptr(outf, f32p) if want_float else None, outW, outH,
                                   _ptr(econ, u32p), _ptr(rcon, u32p), _ptr(centre, u32p), _ptr(rad, u32p),
                                   stages, int(quantize_intermediate), nt)
    if rc != 0:
        raise RuntimeError("ovo_fsr_pipeline_u8 failed: %d" % rc)
    return (out8, outf) if want_float else out8


# ---------------------------------------------------------------------------------------------
# reference (oracle/_ref) wrappers
# ---------------------------------------------------------------------------------------------
def ref_easu(img, outW, outH, con, centre, radius):
    img = np.ascontiguousarray(img, np.float32)
    inH, inW = img.shape[:2]
    c = np.concatenate([con, centre, radius]).astype(np.uint32)
    out = np.zeros((outH, outW, 4), np.float32)
    ref().ref_easu_dispatch(_ptr(img, f32p), inW, inH, _ptr(out, f32p), outW, outH, _ptr(c, u32p))
    return out


def ref_rcas(img, con, centre, radius):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    c = np.concatenate([con, centre, radius]).astype(np.uint32)
    out = np.zeros((H, W, 4), np.float32)
    ref().ref_rcas_dispatch(_ptr(img, f32p), W, H, _ptr(out, f32p), _ptr(c, u32p))
    return out


# ---------------------------------------------------------------------------------------------
# NIS
# ---------------------------------------------------------------------------------------------
def _nis_lib():
    L = lib()
    if not getattr(L, "_nis_bound", False):
        L.ovo_nis_upscale.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_void_p, f32p, f32p, C.c_int]
        L.ovo_nis_upscale.restype = C.c_int
        L.ovo_nis_sharpen.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_void_p, C.c_int]
        L.ovo_nis_sharpen.restype = C.c_int
        L._nis_bound = True
    return L


def nis_block(cfg_words, centre, radius, debug=0):
    """NISConfig as PostProcessor.cpp:308-310 uploads it: reserved1 = debug (float), centre/radius at byte 112."""
    blk = np.array(cfg_words, np.uint32).copy()
    blk[27] = np.array([1.0 if debug else 0.0], np.float32).view(np.uint32)[0]
    blk[28:32] = centre
    blk[32:36] = radius
    return blk


def nis_upscale(img, outW, outH, blk, coef_scale, coef_usm, nthreads=0):
    img = np.ascontiguousarray(img, np.float32)
    inH, inW = img.shape[:2]
    out = np.empty((outH, outW, 4), np.float32)
    cs, cu = np.ascontiguousarray(coef_scale, np.float32), np.ascontiguousarray(coef_usm, np.float32)
    blk = np.ascontiguousarray(blk, np.uint32)
    nt = nthreads or lib().ovo_max_threads()
    rc = _nis_lib().ovo_nis_upscale(_ptr(img, f32p), inW, inH, _ptr(out, f32p), outW, outH, blk.ctypes.data,
                                    _ptr(cs, f32p), _ptr(cu, f32p), nt)
    assert rc == 0
    return out


def nis_sharpen(img, blk, nthreads=0):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    out = np.empty((H, W, 4), np.float32)
    blk = np.ascontiguousarray(blk, np.uint32)
    nt = nthreads or lib().ovo_max_threads()
    rc = _nis_lib().ovo_nis_sharpen(_ptr(img, f32p), W, H, _ptr(out, f32p), blk.ctypes.data, nt)
    assert rc == 0
    return out


def _ref_nis():
    R = ref()
    if not getattr(R, "_nis_bound", False):
        R.ref_nis_upscale_dispatch.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, u32p, f32p, f32p]
        R.ref_nis_sharpen_dispatch.argtypes = [f32p, C.c_int, C.c_int, f32p, u32p, f32p, f32p]
        R._nis_bound = True
    return R


def ref_nis_coefs():
    sc, us = np.zeros(512, np.float32), np.zeros(512, np.float32)
    ref().ref_nis_coefs(_ptr(sc, f32p), _ptr(us, f32p))
    return sc.reshape(64, 8), us.reshape(64, 8)


def ref_nis_scaler_config(sharpness, inW, inH, outW, outH):
    buf = np.zeros(64, np.uint32)
    ok = ref().ref_nis_scaler_config(buf.ctypes.data, sharpness, inW, inH, outW, outH)
    return bool(ok), buf


def ref_nis_upscale(img, outW, outH, blk, coef_scale, coef_usm):
    img = np.ascontiguousarray(img, np.float32)
    inH, inW = img.shape[:2]
    out = np.zeros((outH, outW, 4), np.float32)
    cs, cu = np.ascontiguousarray(coef_scale, np.float32), np.ascontiguousarray(coef_usm, np.float32)
    blk = np.ascontiguousarray(blk, np.uint32)
    _ref_nis().ref_nis_upscale_dispatch(_ptr(img, f32p), inW, inH, _ptr(out, f32p), outW, outH, _ptr(blk, u32p),
                                        _ptr(cs, f32p), _ptr(cu, f32p))
    return out


def ref_nis_sharpen(img, blk, coef_scale, coef_usm):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    out = np.zeros((H, W, 4), np.float32)
    cs, cu = np.ascontiguousarray(coef_scale, np.float32), np.ascontiguousarray(coef_usm, np.float32)
    blk = np.ascontiguousarray(blk, np.uint32)
    _ref_nis().ref_nis_sharpen_dispatch(_ptr(img, f32p), W, H, _ptr(out, f32p), _ptr(blk, u32p), _ptr(cs, f32p), _ptr(cu, f32p))
    return out
