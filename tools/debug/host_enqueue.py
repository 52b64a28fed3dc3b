#!/usr/bin/env python3
"""tools/debug/host_enqueue.py -- host-side cost of one ovrfsr_apply_batch call from Python (ctypes + the torch plumbing of
openvr_fsr_amd.PostProcessor.apply_batch) with 1, 2, 4, 8 host threads, each with its own ctx and stream.

bench.py's direct multi-GPU launcher (`python bench.py --gpus N`, one process) drives every device from its own Python
thread; the ctypes call releases the GIL but the Python around it (image_of, struct packing, status check) does not.  This
measures that ceiling: calls per second the host can issue, on images so small (one 32x32 tile) that the GPU side of a call
is a few microseconds.  All threads target device 0 (a gpurun box has one GPU): the host path is the same as with one
device per thread, and the device never becomes the bottleneck before the host does at these sizes.
The torchrun launcher (what the round driver uses for N > 1) has one process, hence one GIL, per GPU: there the
single-thread row applies."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import openvr_fsr_amd as A  # noqa: E402

CALLS = 3000


def worker(i, gate, out):
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        pp = A.PostProcessor(device=0, fsr_enabled=1, out_width=32, out_height=32, sharpness=0.9, radius=2.0)
        texs = torch.randint(0, 255, (2, 24, 24, 4), dtype=torch.uint8, device="cuda")
        outs = torch.empty((2, 32, 32, 4), dtype=torch.uint8, device="cuda")
        for _ in range(50):
            pp.apply_batch(texs, outs, first_eye=A.EYE_LEFT, alternate_eyes=True)
        stream.synchronize()
        gate.wait()
        t0 = time.perf_counter()
        for k in range(CALLS):
            pp.apply_batch(texs, outs, first_eye=A.EYE_LEFT, alternate_eyes=True)
            if (k & 255) == 255:
                stream.synchronize()   # keep the launch queue shallow: a full queue would make enqueue wait for the device
        t1 = time.perf_counter()
        stream.synchronize()
        out[i] = (t1 - t0) / CALLS
        pp.close()


print("threads  us_per_call(mean over threads)  calls_per_s(all threads)   [EASU + RCAS apply_batch of 2 images 24x24 -> 32x32, %d calls per thread]" % CALLS)
for n in (1, 2, 4, 8):
    gate = threading.Barrier(n)
    res = [0.0] * n
    ts = [threading.Thread(target=worker, args=(i, gate, res)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    mean = sum(res) / n
    print("%7d  %28.1f  %24.0f" % (n, mean * 1e6, n / mean))
