# tests/debug/bounds_fuzz_dbg.py SEED -- one instance of test_masked_product_fuzz, product library vs checked build, where do they differ
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
import numpy as np
if os.environ.get("INNER") != "1":
    for lib in ("openvr_fsr_amd/libopenvr_fsr_amd.so", "ab/bounds.so"):
        r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, INNER="1", OVRFSR_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print("==", lib); print(r.stdout[-3000:]); print(r.stderr[-1500:])
    sys.exit(0)
import torch
import openvr_fsr_amd as A
from oracle import oracle as O
from tests import synth
from tests.test_gpu_fuzz import _outside_px
seed = int(sys.argv[1])
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
print("case", iw, ih, ow, oh, "scale", s, "radius", radius, "proj", proj, "eye", eye, "debug", debug, "sharp", sharp, "pads", pad_in, pad_out)
big_in = torch.zeros((ih, iw + pad_in, 4), dtype=torch.uint8, device="cuda")
big_in[:, :iw] = torch.from_numpy(img8).cuda()
big_out = torch.empty((oh, ow + pad_out, 4), dtype=torch.uint8, device="cuda")
for fused in (-1, 0, 1):
    for rep in range(1):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=sharp, radius=radius, proj_centre=proj, debug_mode=debug, precision=0, fused=fused)
        big_out.fill_(99)
        got = pp.apply(eye, big_in[:, :iw], out=big_out[:, :ow]).cpu().numpy()
        pp.close()
        d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
        ys, xs = np.nonzero(d > 1)
        lib = A.library()
        if hasattr(lib, "ovrfsr_debug_bounds"):
            n = lib.ovrfsr_debug_bounds_slots(); buf = (ctypes.c_ulonglong * n)()
            lib.ovrfsr_debug_bounds.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
            lib.ovrfsr_debug_bounds(buf, n, 1); nk = (n - 5) // 3; v = list(buf)
            print("   bounds: oob per kind", {i: x for i, x in enumerate(v[:nk]) if x}, "first", v[3 * nk:])
        print("fused", fused, "rep", rep, "max", int(d.max()), "n>1", len(ys), "outside-mismatch", int((got[outside] != want[outside]).sum()),
              "bbox", (int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())) if len(ys) else None, "tiles", sorted(set((int(x) // 32, int(y) // 32) for x, y in zip(xs, ys)))[:12])
lib = A.library()
if hasattr(lib, "ovrfsr_debug_bounds"):
    n = lib.ovrfsr_debug_bounds_slots(); buf = (ctypes.c_ulonglong * n)()
    lib.ovrfsr_debug_bounds.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
    lib.ovrfsr_debug_bounds(buf, n, 0); nk = (n - 5) // 3; v = list(buf)
    print("bounds: checked", sum(v[2 * nk:3 * nk]), "oob", sum(v[:nk]), "pad", sum(v[nk:2 * nk]), "first", v[3 * nk:])
