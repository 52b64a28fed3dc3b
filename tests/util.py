"""Shared helpers for the GPU parity tests: run the product through its C ABI on numpy inputs."""
import numpy as np


def run_gpu(img, outW, outH, out_dtype, eye=0, **cfg_kw):
    """img: numpy [H,W,4] uint8/float16/float32 -> numpy [outH,outW,4] of out_dtype via ovrfsr_apply."""
    import torch
    import openvr_fsr_amd as A
    kw = dict(fsr_enabled=1, out_width=outW, out_height=outH, radius=2.0)
    kw.update(cfg_kw)
    pp = A.PostProcessor(**kw)
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16, np.float32: torch.float32}[np.dtype(out_dtype).type]
    out = pp.apply(eye, t, out_dtype=tdt)
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    pp.close()
    return res


def lsb_stats(a8, b8):
    d = np.abs(a8.astype(np.int16) - b8.astype(np.int16))
    return int(d.max()), float((d != 0).mean())
