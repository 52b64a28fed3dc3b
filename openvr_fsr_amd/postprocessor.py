"""Python mirror of the reference's vr::PostProcessor surface (src/postprocess/PostProcessor.h:12-13)
over the C ABI: ``apply(eye, texture, bounds)`` / ``reset()``.  torch is used only for device memory
and streams; every pixel is produced by the HIP kernels in libopenvr_fsr_amd.so."""
import ctypes as C

import torch

from . import _capi as K

_TORCH_FMT = {torch.uint8: (K.FORMAT_RGBA8, 4), torch.float16: (K.FORMAT_RGBA16F, 8), torch.float32: (K.FORMAT_RGBA32F, 16)}
_FMT_TORCH = {K.FORMAT_RGBA8: torch.uint8, K.FORMAT_RGBA16F: torch.float16, K.FORMAT_RGBA32F: torch.float32}


def image_of(t, fmt=None):
    """[H, W, 4] device tensor -> ovrfsr_image (no copy); ``fmt`` overrides the format implied by the dtype (e.g.
    K.FORMAT_BGRA8 for a uint8 tensor whose channel order is B,G,R,A).  A [H, W] int32 tensor is an R10G10B10A2_UNORM image (one packed
    dword per texel: R bits 0-9, G 10-19, B 20-29, A 30-31)."""
    if t.dim() == 2 and t.dtype == torch.int32 and t.is_cuda:
        if t.stride(1) != 1:
            raise ValueError("texels must be contiguous")
        return K.Image(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0) * 4, K.FORMAT_RGB10A2)
    if t.dim() != 3 or t.shape[2] != 4 or not t.is_cuda:
        raise ValueError("expected a [H, W, 4] tensor on the GPU")
    if t.stride(2) != 1 or t.stride(1) != 4:
        raise ValueError("texels must be contiguous RGBA")
    dfmt, tb = _TORCH_FMT[t.dtype]
    return K.Image(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0) * t.element_size(), dfmt if fmt is None else fmt)


class PostProcessor:
    def __init__(self, cfg=None, device=None, **cfg_kw):
        self._lib = K.library()
        self._ctx = C.c_void_p()
        if not torch.cuda.is_available():
            raise K.OvrFsrError(4, "no HIP device visible to torch")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.cfg = cfg if cfg is not None else K.Config.default(**cfg_kw)
        rc = self._lib.ovrfsr_create(self.device, C.byref(self.cfg), C.byref(self._ctx))
        if rc != 0:
            raise K.OvrFsrError(rc, "ovrfsr_create")

    def close(self):
        if self._ctx:
            self._lib.ovrfsr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise K.OvrFsrError(rc, (self._lib.ovrfsr_last_error(self._ctx) or b"").decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_config(self, cfg):
        self.cfg = cfg
        self._check(self._lib.ovrfsr_set_config(self._ctx, C.byref(cfg)))

    def reset(self):
        """PostProcessor::Reset()"""
        self._check(self._lib.ovrfsr_reset(self._ctx))

    def output_size(self, inW, inH):
        return K.output_size(self.cfg, inW, inH)

    def apply(self, eye, tex, bounds=None, out=None, out_dtype=None, in_format=None):
        """PostProcessor::Apply(eye, texture, bounds).  Returns the tensor the compositor should receive.
        With ``out=None`` and ``out_dtype=None`` the ctx-owned output is wrapped (valid until the next apply)."""
        img = image_of(tex, in_format)
        if out is None and out_dtype is not None:
            ow, oh = self.output_size(tex.shape[1], tex.shape[0])
            shape = (oh, ow) if out_dtype == torch.int32 else (oh, ow, 4)  # int32 = packed R10G10B10A2
            out = torch.empty(shape, dtype=out_dtype, device=tex.device)
        oimg = image_of(out) if out is not None else K.Image()
        b = C.byref(bounds) if bounds is not None else None
        self._check(self._lib.ovrfsr_apply(self._ctx, int(eye), C.byref(img), b, C.byref(oimg), self._stream()))
        if out is not None and oimg.data == out.data_ptr():
            return out
        if oimg.data == tex.data_ptr():
            return tex  # pass-through (fsr disabled)
        return _wrap(oimg, tex.device)

    def apply_batch(self, texs, outs, first_eye=K.EYE_LEFT, alternate_eyes=True, in_format=None, shared=False):
        """texs: [N, H, W, 4], outs: [N, outH, outW, 4] (image i = eye first_eye ^ (i & alternate)); shared=True: every image is a
        side-by-side texture holding both eyes (ovrfsr_apply_batch_shared)."""
        n = texs.shape[0]
        if outs.shape[0] != n:
            raise ValueError("outs holds %d images for a batch of %d" % (outs.shape[0], n))
        i0, o0 = image_of(texs[0], in_format), image_of(outs[0])
        if shared:
            self._check(self._lib.ovrfsr_apply_batch_shared(self._ctx, n, C.byref(i0), texs.stride(0) * texs.element_size(), C.byref(o0),
                                                            outs.stride(0) * outs.element_size(), self._stream()))
            return outs
        self._check(self._lib.ovrfsr_apply_batch(self._ctx, n, int(first_eye), int(bool(alternate_eyes)), C.byref(i0),
                                                 texs.stride(0) * texs.element_size(), C.byref(o0),
                                                 outs.stride(0) * outs.element_size(), self._stream()))
        return outs

    def pair_pending(self):
        """cfg.pair_submit: did the last apply only record its submission (ovrfsr_pair_pending)?"""
        return bool(self._lib.ovrfsr_pair_pending(self._ctx))

    def last_gpu_time_ms(self):
        ms = C.c_float()
        self._check(self._lib.ovrfsr_last_gpu_time_ms(self._ctx, C.byref(ms)))
        return ms.value


    def average_gpu_time_ms(self):
        """(last published mean of 500 readings in ms -- per frame when each eye has its own texture --, number published):
        the reference's "Average GPU processing time for upscale" log line (PostProcessor.cpp:605-626)."""
        ms, n = C.c_float(), C.c_uint32()
        self._check(self._lib.ovrfsr_average_gpu_time_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def _wrap(img, device):
    """View ctx-owned device memory as a torch tensor without copying (via __cuda_array_interface__)."""

    class _Holder:
        pass

    h = _Holder()
    if img.format == K.FORMAT_RGB10A2:   # one packed dword per texel
        h.__cuda_array_interface__ = {"shape": (img.height, img.width), "typestr": "<i4", "strides": (img.pitch_bytes, 4),
                                      "data": (img.data, False), "version": 2}
        return torch.as_tensor(h, device=device)
    dt = _FMT_TORCH[img.format]
    es = torch.empty((), dtype=dt).element_size()
    typestr = {torch.uint8: "|u1", torch.float16: "<f2", torch.float32: "<f4"}[dt]
    h.__cuda_array_interface__ = {"shape": (img.height, img.width, 4), "typestr": typestr,
                                  "strides": (img.pitch_bytes, 4 * es, es), "data": (img.data, False), "version": 2}
    return torch.as_tensor(h, device=device)
