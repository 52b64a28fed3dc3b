# tools/debug/nis_content.py -- NVScaler time per eye for three content classes: flat (no edge anywhere: every wave
# takes the exact early-out), structured (bench default) and uniform-random (all four directional filters everywhere).
import sys, time; sys.path.insert(0, '.')
import torch
import openvr_fsr_amd as A
import bench
inW, inH, outW, outH = 1683, 1869, 2244, 2492
n = 32
dev = torch.device("cuda")
outs = torch.empty((n, outH, outW, 4), dtype=torch.uint8, device="cuda")
pp = A.PostProcessor(fsr_enabled=1, use_nis=1, out_width=outW, out_height=outH, sharpness=0.9, radius=2.0)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
flat = torch.full((n, inH, inW, 4), 120, dtype=torch.uint8, device=dev)
grad = bench.synth_batch(n, inW, inH, torch.uint8, dev, 1)
rnd = bench.random_batch(n, inW, inH, torch.uint8, dev, 1)
for name, t in (("flat", flat), ("structured", grad), ("random", rnd)):
    ms = timeit(lambda: pp.apply_batch(t, outs))
    print("%-10s %.4f ms/step  %.2f us/eye" % (name, ms, ms / n * 1e3))
