import sys, time; sys.path.insert(0, '.')
import torch
import openvr_fsr_amd as A
import bench
inW, inH, outW, outH = 1683, 1869, 2244, 2492
n = 32
texs = bench.synth_batch(n, inW, inH, torch.uint8, torch.device("cuda"), 1)
outs = torch.empty((n, outH, outW, 4), dtype=torch.uint8, device="cuda")
def mk(): return A.PostProcessor(fsr_enabled=1, out_width=outW, out_height=outH, sharpness=0.9, radius=2.0)
pp = mk(); ppa = mk(); ppb = mk()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def single():
    pp.apply_batch(texs, outs)
def dual(k=2):
    h = n // 2
    with torch.cuda.stream(sa): ppa.apply_batch(texs[:h], outs[:h])
    with torch.cuda.stream(sb): ppb.apply_batch(texs[h:], outs[h:])
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for r in range(3):
    print("single %.4f ms   dual-stream %.4f ms" % (timeit(single), timeit(dual)))
