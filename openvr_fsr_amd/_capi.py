"""ctypes declarations for include/openvr_fsr_amd.h (kept in the same order as the header)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FORMAT_RGBA8, FORMAT_RGBA16F, FORMAT_RGBA32F, FORMAT_RGB10A2, FORMAT_BGRA8 = 0, 1, 2, 3, 4
PRECISION_FP32, PRECISION_FP32_STRICT = 0, 2
EYE_LEFT, EYE_RIGHT = 0, 1

STATUS = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED", 3: "HIP", 4: "NO_DEVICE", 5: "DISABLED", 6: "OUT_OF_MEMORY"}


class OvrFsrError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("ovrfsr status %s (%d): %s" % (STATUS.get(status, "?"), status, msg))
        self.status = status


class Image(C.Structure):
    """ovrfsr_image"""
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pitch_bytes", C.c_uint32), ("format", C.c_uint32)]


class Bounds(C.Structure):
    """ovrfsr_bounds == vr::VRTextureBounds_t"""
    _fields_ = [("uMin", C.c_float), ("vMin", C.c_float), ("uMax", C.c_float), ("vMax", C.c_float)]


class Config(C.Structure):
    """ovrfsr_config: the numeric part of the reference's Config singleton (Config.h:10-17)."""
    _fields_ = [("struct_size", C.c_uint32), ("fsr_enabled", C.c_int32), ("use_nis", C.c_int32),
                ("debug_mode", C.c_int32), ("render_scale", C.c_float), ("sharpness", C.c_float),
                ("radius", C.c_float), ("proj_centre", C.c_float * 4), ("out_width", C.c_uint32),
                ("out_height", C.c_uint32), ("precision", C.c_int32), ("quantize_intermediate", C.c_int32),
                ("fused", C.c_int32), ("stage_mask", C.c_int32), ("pair_submit", C.c_int32), ("reserved", C.c_int32 * 2)]

    @classmethod
    def default(cls, **kw):
        cfg = cls()
        library().ovrfsr_config_default(C.byref(cfg))
        for k, v in kw.items():
            if k == "proj_centre":
                for i in range(4):
                    cfg.proj_centre[i] = v[i]
            else:
                if not hasattr(cfg, k):
                    raise AttributeError(k)
                setattr(cfg, k, v)
        return cfg


def library_path():
    # OVRFSR_LIB: A/B benchmarking of two builds of the same library in one gpurun call (tools/ab.sh)
    return os.environ.get("OVRFSR_LIB") or os.path.join(_HERE, "libopenvr_fsr_amd.so")


def have_library():
    return os.path.exists(library_path())


def library():
    """Load libopenvr_fsr_amd.so.  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise OvrFsrError(4, "%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    L = C.CDLL(path)
    u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.ovrfsr_abi_version.restype = C.c_uint32
    L.ovrfsr_config_default.argtypes = [C.POINTER(Config)]
    L.ovrfsr_config_default.restype = None
    L.ovrfsr_create.argtypes = [C.c_int, C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.ovrfsr_destroy.argtypes = [C.c_void_p]
    L.ovrfsr_destroy.restype = None
    L.ovrfsr_set_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.ovrfsr_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.ovrfsr_reset.argtypes = [C.c_void_p]
    L.ovrfsr_output_size.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32, u32p, u32p]
    L.ovrfsr_apply.argtypes = [C.c_void_p, C.c_int, C.POINTER(Image), C.POINTER(Bounds), C.POINTER(Image), C.c_void_p]
    L.ovrfsr_apply_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(Image), C.c_size_t,
                                     C.POINTER(Image), C.c_size_t, C.c_void_p]
    L.ovrfsr_apply_batch_shared.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Image), C.c_size_t, C.POINTER(Image), C.c_size_t, C.c_void_p]
    L.ovrfsr_last_error.argtypes = [C.c_void_p]
    L.ovrfsr_last_error.restype = C.c_char_p
    L.ovrfsr_last_gpu_time_ms.argtypes = [C.c_void_p, f32p]
    L.ovrfsr_average_gpu_time_ms.argtypes = [C.c_void_p, f32p, u32p]
    L.ovrfsr_easu_con.argtypes = [u32p] + [C.c_float] * 6
    L.ovrfsr_easu_con.restype = None
    L.ovrfsr_rcas_con.argtypes = [u32p, C.c_float]
    L.ovrfsr_rcas_con.restype = None
    L.ovrfsr_mask_constants.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32, f32p, C.c_float, C.c_int, C.c_int]
    L.ovrfsr_mask_constants.restype = None
    L.ovrfsr_nis_scaler_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 4
    L.ovrfsr_nis_sharpen_config.argtypes = [C.c_void_p, C.c_float] + [C.c_uint32] * 2
    L.ovrfsr_nis_coef_scale.restype = f32p
    L.ovrfsr_nis_coef_usm.restype = f32p
    L.ovrfsr_config_from_json.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Config)]
    L.ovrfsr_save_ppm.argtypes = [C.POINTER(Image), C.c_char_p, C.c_void_p]
    L.ovrfsr_save_dds.argtypes = [C.POINTER(Image), C.c_char_p, C.c_void_p]
    L.ovrfsr_pair_pending.argtypes = [C.c_void_p]
    if L.ovrfsr_abi_version() != 5:
        raise OvrFsrError(1, "ABI version mismatch")
    _LIB = L
    return L


# ---- constants-only helpers (no GPU) -----------------------------------------------------------
def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def easu_con(inW, inH, outW, outH):
    con = np.zeros(16, np.uint32)
    library().ovrfsr_easu_con(_u32p(con), inW, inH, inW, inH, outW, outH)
    return con


def rcas_con(stops):
    con = np.zeros(4, np.uint32)
    library().ovrfsr_rcas_con(_u32p(con), stops)
    return con


def mask_constants(outW, outH, proj, radius, only_one_eye, eye):
    centre, rad = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
    p = np.asarray(proj, np.float32)
    library().ovrfsr_mask_constants(_u32p(centre), _u32p(rad), outW, outH, p.ctypes.data_as(C.POINTER(C.c_float)),
                                    radius, int(only_one_eye), int(eye))
    return centre, rad


def nis_scaler_config(sharpness, inW, inH, outW, outH):
    buf = np.zeros(64, np.uint32)
    ok = library().ovrfsr_nis_scaler_config(buf.ctypes.data, sharpness, inW, inH, outW, outH)
    return bool(ok), buf


def nis_sharpen_config(sharpness, inW, inH):
    buf = np.zeros(64, np.uint32)
    ok = library().ovrfsr_nis_sharpen_config(buf.ctypes.data, sharpness, inW, inH)
    return bool(ok), buf


def nis_coefs():
    L = library()
    s = np.ctypeslib.as_array(L.ovrfsr_nis_coef_scale(), shape=(64, 8)).copy()
    u = np.ctypeslib.as_array(L.ovrfsr_nis_coef_usm(), shape=(64, 8)).copy()
    return s, u


def config_from_json(text):
    """Config::Load on the text of an openvr_mod.cfg; returns (status, Config)."""
    cfg = Config()
    b = text.encode() if isinstance(text, str) else text
    rc = library().ovrfsr_config_from_json(b, len(b), C.byref(cfg))
    return rc, cfg


def output_size(cfg, inW, inH):
    w, h = C.c_uint32(), C.c_uint32()
    rc = library().ovrfsr_output_size(C.byref(cfg), inW, inH, C.byref(w), C.byref(h))
    if rc != 0:
        raise OvrFsrError(rc)
    return w.value, h.value
