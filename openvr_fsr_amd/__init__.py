"""openvr_fsr_amd -- MI355X-native (gfx950) per-eye VR upscaler: FSR1 EASU + RCAS, NIS.

The product is the C-ABI shared library ``libopenvr_fsr_amd.so`` (include/openvr_fsr_amd.h).  This
package is the thin Python plumbing used by the tests and bench.py: it loads the library with
ctypes and hands it torch device pointers and streams.  There is no CPU fallback anywhere in this
package: if the HIP library is missing or no gfx950 device is present, calls raise.
"""
from ._capi import (  # noqa: F401
    Config, Image, Bounds, OvrFsrError, library, library_path, have_library,
    FORMAT_RGBA8, FORMAT_RGBA16F, FORMAT_RGBA32F, FORMAT_RGB10A2, FORMAT_BGRA8,
    PRECISION_FP32, PRECISION_FP32_STRICT,
    EYE_LEFT, EYE_RIGHT,
    easu_con, rcas_con, mask_constants, nis_scaler_config, nis_sharpen_config, nis_coefs, output_size, config_from_json,
)
from .postprocessor import PostProcessor  # noqa: F401
