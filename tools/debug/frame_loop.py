# tools/debug/frame_loop.py RADIUS [FRAMES] -- the reference's call pattern and nothing else: one stereo pair per frame, one ovrfsr_apply per
# eye (L, R, L, R ...) at C2's shape.  Run under `rocprofv3 --kernel-trace` by tools/debug/frame_timeline.sh to get the per-launch timeline.
import sys; sys.path.insert(0, '.')
import torch
import openvr_fsr_amd as A
import bench
radius = float(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda")
inW, inH, outW, outH = 1683, 1869, 2244, 2492
texs = bench.synth_batch(2, inW, inH, torch.uint8, dev, 1)
outs = torch.empty((2, outH, outW, 4), dtype=torch.uint8, device=dev)
pp = A.PostProcessor(fsr_enabled=1, out_width=outW, out_height=outH, sharpness=0.9, radius=radius)
for _ in range(frames):
    pp.apply(A.EYE_LEFT, texs[0], out=outs[0])
    pp.apply(A.EYE_RIGHT, texs[1], out=outs[1])
torch.cuda.synchronize()
pp.close()
