# tools/debug/graph_capture.py -- can ovrfsr_apply_batch be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed?
import sys, time; sys.path.insert(0, '.')
import torch
import openvr_fsr_amd as A
import bench
dev = torch.device("cuda")
for name, radius, nis in (("C2", 2.0, 0), ("C2r", 0.5, 0), ("C3r", 0.5, 1)):
    inW, inH, outW, outH = 1683, 1869, 2244, 2492
    for pairs in (1, 16):
        n = 2 * pairs
        texs = bench.synth_batch(n, inW, inH, torch.uint8, dev, 1)
        ref = torch.empty((n, outH, outW, 4), dtype=torch.uint8, device=dev)
        out = torch.zeros_like(ref)
        pp = A.PostProcessor(fsr_enabled=1, use_nis=nis, out_width=outW, out_height=outH, sharpness=0.9, radius=radius)
        pp.apply_batch(texs, ref); pp.apply_batch(texs, out); torch.cuda.synchronize()   # warm: lazy resources exist now
        out.zero_()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                pp.apply_batch(texs, out)
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        def t(fn, it=200):
            for _ in range(10): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(it): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
        print("%-4s pairs=%2d graph replay == direct: %s   direct %.1f us/step   graph %.1f us/step" % (name, pairs, same, t(lambda: pp.apply_batch(texs, out)), t(g.replay)))
        pp.close()
