"""Natural-content fixtures (tests/golden/natural_*.npz, cut by tests/golden/make_natural.py): loaders, mirror tiling to any size, and the
oracle digests pinned in tests/golden/natural_golden.json."""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ("cube", "portal", "hopper")


def load(name):
    return np.load(os.path.join(GOLD, "natural_%s.npz" % name))["rgba8"]


def tiled_u8(w, h, seed):
    """RGBA8 [h, w, 4]: fixture seed % 3, mirror-tiled (reflected at every 256-texel seam: no artificial edges) from an offset that depends on
    the seed -- the natural-content member of the generator family (same signature as tests/synth.py's)."""
    src = load(NAMES[seed % 3])
    n = src.shape[0]
    ox, oy = (seed * 37) % n, (seed * 101) % n

    def index(length, off):
        i = (np.arange(length) + off) % (2 * n)
        return np.where(i < n, i, 2 * n - 1 - i)

    return np.ascontiguousarray(src[index(h, oy)][:, index(w, ox)])


def oracle_digests(rgba8):
    """SHA-256 of the oracle's outputs for one fixture at 256 -> 341 (x4/3): EASU in UNORM8, EASU -> UNORM8 -> RCAS (sharpness 0.9), NVScaler"""
    import openvr_fsr_amd as A
    from oracle import oracle as O
    ih, iw = rgba8.shape[:2]
    ow, oh = iw * 4 // 3, ih * 4 // 3
    f = O.unorm8_to_float(rgba8)
    easu8 = O.float_to_unorm8(O.easu(f, ow, oh))
    pipe8 = O.fsr_pipeline_u8(rgba8, ow, oh, sharpness=0.9)
    cs, cu = A.nis_coefs()
    ok, cfg = A.nis_scaler_config(0.9, iw, ih, ow, oh)
    centre, rad = O.mask_constants(ow, oh)
    nis8 = O.float_to_unorm8(O.nis_upscale(f, ow, oh, O.nis_block(cfg, centre, rad, 0), cs, cu))
    return {"easu_u8": hashlib.sha256(easu8.tobytes()).hexdigest(), "pipeline_u8": hashlib.sha256(pipe8.tobytes()).hexdigest(),
            "nis_u8": hashlib.sha256(nis8.tobytes()).hexdigest()}
