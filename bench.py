#!/usr/bin/env python3
"""bench.py -- stereo eye-pairs/sec of the EASU+RCAS hot path at 1683x1869 -> 2244x2492 (BASELINE.json
config C2), achieved fraction of the HBM roofline for the dominant kernel, and the CPU oracle timed
beside it.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; batches shard, no data-path collective)

A "step" = one pass of the hot path (PostProcessor::Apply for both eyes of every pair) over one
batch of `--pairs` synthetic stereo pairs that are already resident in HBM.  Weak scaling: every
rank owns its own batch.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CLOCK_RAMP_S = 0.3      # untimed sustained load before the warm-up steps (see main)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable by a float4 copy

WORKLOADS = {
    # name: (inW, inH, outW, outH, in/out dtype, radius, use_nis)   -- BASELINE.json configs
    "C2": (1683, 1869, 2244, 2492, torch.uint8, 2.0, 0),     # EASU+RCAS stereo pairs (headline)
    "C3": (1683, 1869, 2244, 2492, torch.uint8, 2.0, 1),     # NIS scaler (built-in USM sharpen)
    "C4": (2244, 2492, 2916, 3240, torch.uint8, 2.0, 0),     # renderScale 1.3 batch
    "C5": (2370, 2370, 3160, 3160, torch.float16, 0.5, 0),   # radius-masked, RGBA16F packed I/O
    "C2r": (1683, 1869, 2244, 2492, torch.uint8, 0.5, 0),    # C2's shape with the reference's shipped radius 0.5 (openvr_mod.cfg)
    "C3r": (1683, 1869, 2244, 2492, torch.uint8, 0.5, 1),    # C3 (NIS) with the shipped radius 0.5
    "C2s": (2244, 2492, 2244, 2492, torch.uint8, 2.0, 0),    # renderScale 1: RCAS only (PostProcessor.cpp:586-594)
    "C3s": (2244, 2492, 2244, 2492, torch.uint8, 2.0, 1),    # renderScale 1 with useNis: NVSharpen only
}


def random_batch(n_img, w, h, dtype, device, base_seed):
    """Uniform-random texels (second distribution of SURVEY.md 8d: no smooth regions, worst case for edge analysis)."""
    g = torch.Generator(device=device)
    g.manual_seed(base_seed)
    if dtype == torch.uint8:
        out = torch.randint(0, 256, (n_img, h, w, 4), generator=g, device=device, dtype=torch.uint8)
        out[..., 3] = 255
    else:
        out = torch.rand((n_img, h, w, 4), generator=g, device=device).to(dtype)
        out[..., 3] = 1.0
    return out


def synth_batch(n_img, w, h, dtype, device, base_seed):
    """Structured synthetic eye images generated ON the consuming GPU: sinusoid gradients,
    hard 45/135-degree edges, +-4/255 noise, a constant block; alpha = 1."""
    out = torch.empty((n_img, h, w, 4), dtype=dtype, device=device)
    y = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    period = max(16, min(w, h) // 6)
    for i in range(n_img):
        g = torch.Generator(device=device)
        g.manual_seed(base_seed + i)
        ph = torch.rand(3, generator=g, device=device) * 6.28
        img = torch.empty((h, w, 3), dtype=torch.float32, device=device)
        for c in range(3):
            img[..., c] = 0.5 + 0.35 * torch.sin(x * (0.011 + 0.004 * c) + ph[c]) * torch.cos(y * (0.008 + 0.003 * c) - ph[c])
        d45 = ((x + y) % period) < (period / 2)
        d135 = ((x - y) % (period * 1.5)) < (period * 0.5)
        img[..., 0] = torch.where(d45, img[..., 0] * 0.35, img[..., 0])
        img[..., 1] = torch.where(d135, 1.0 - img[..., 1] * 0.5, img[..., 1])
        img[..., 2] = torch.where(d45 & d135, torch.full_like(img[..., 2], 0.95), img[..., 2])
        img += (torch.randint(-4, 5, (h, w, 3), generator=g, device=device).float() / 255.0)
        img[h // 3:h // 3 + h // 8, w // 3:w // 3 + w // 8, :] = torch.tensor([0.25, 0.5, 0.75], device=device)
        img.clamp_(0.0, 1.0)
        if dtype == torch.uint8:
            out[i, ..., :3] = torch.floor(img * 255.0 + 0.5).to(torch.uint8)
            out[i, ..., 3] = 255
        else:
            out[i, ..., :3] = img.to(dtype)
            out[i, ..., 3] = 1.0
    return out


def shard_seed(pairs_per_gpu, rank):
    """Seed of the first eye image of `rank`'s shard: image g (global index) has seed 0x5EED0000 + g, i.e.
    0x5EED0000 + 2*pair + eye; rank r owns global pairs [r*P, (r+1)*P)."""
    return 0x5EED0000 + 2 * pairs_per_gpu * rank


def timed_region(step, steps, barrier):
    """EXACTLY `steps` calls of step() bracketed by barrier()+sync on both sides; returns local wall seconds."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(dt, world, device):
    if world == 1:
        return dt
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def measured_traffic(workload, kernel, n_img):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this workload (tools/profile.sh:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction on FETCH_SIZE), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_per_eye.json")
    try:
        per_eye = json.load(open(path))[workload]
        hit = [v for k, v in per_eye.items() if any(k.endswith("::" + part) for part in kernel.split("+"))]
        return int(sum(h["hbm_bytes_per_eye"] for h in hit) * n_img) if hit else None
    except (OSError, KeyError, ValueError):
        return None


def time_events(fn, iters, stream):
    """Average ms per call of fn() measured with HIP events recorded on `stream`."""
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()  # untimed: lazy resource build of a fresh context (allocations, table uploads, aux stream)
    s.record(stream)
    for _ in range(iters):
        fn()
    e.record(stream)
    e.synchronize()
    return s.elapsed_time(e) / iters


def usable_cores():
    """CPUs this process may actually use: affinity mask, capped by the container's cgroup CPU quota (cpu.max) --
    a 256-thread host with a 16-CPU quota has 16, and running 256 OpenMP threads there is slower than 16."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def copy_ceiling_gbps(dev):
    """What a plain device copy reaches on this chip right now (read + write bytes / time): the practical HBM ceiling
    next to the 8 TB/s vendor peak (SURVEY.md 8d asks for both)."""
    n = 1 << 29
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    return 2.0 * n * 10 / (time.perf_counter() - t0) / 1e9


def cpu_baseline(inW, inH, outW, outH, sharpness):
    """The oracle (C restatement, OpenMP) on stereo pairs of the same workload, one thread per usable core, for a
    bounded sample: pairs are processed until ~8 s of wall time have passed (at least one, at most eight)."""
    from oracle import oracle as O
    from tests import synth
    cores = min(usable_cores(), O.lib().ovo_max_threads())
    imgs = [synth.structured_u8(inW, inH, synth.seed_for(0, e)) for e in range(2)]
    O.fsr_pipeline_u8(imgs[0][:64, :64].copy(), 85, 85, sharpness=sharpness)  # warm the library
    pairs, t0 = 0, time.perf_counter()
    while pairs < 8 and (pairs == 0 or time.perf_counter() - t0 < 8.0):
        for im in imgs:
            O.fsr_pipeline_u8(im, outW, outH, sharpness=sharpness, nthreads=cores)
        pairs += 1
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()   # and one eye on a single thread (report single-thread too)
    O.fsr_pipeline_u8(imgs[0], outW, outH, sharpness=sharpness, nthreads=1)
    single = 0.5 / (time.perf_counter() - t1)
    return {"value": round(pairs / dt, 4), "unit": "eye-pairs/s", "cores": cores, "kind": "port", "single_thread_value": round(single, 4),
            "sample": "%d stereo pair(s) %dx%d->%dx%d RGBA8, EASU+RCAS (UNORM8 intermediate), oracle/liboracle.so with %d OpenMP "
                      "threads (usable cores of this container), %.2f s wall = %.1f core-seconds"
                      % (pairs, inW, inH, outW, outH, cores, dt, dt * cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--pairs", type=int, default=16, help="stereo pairs per GPU per step")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "strict"])
    ap.add_argument("--fused", type=int, default=-1)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--content", default="structured", choices=["structured", "random"],
                    help="synthetic eye content: structured (gradients+edges+noise, default) or uniform random")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)          # before the process group: its collectives must use THIS rank's GPU
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)   # RCCL: only the timing barrier / max-reduce use it

    import openvr_fsr_amd as A
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    sharpness = 0.9
    prec = {"fp32": A.PRECISION_FP32, "fp16": A.PRECISION_FP16, "strict": A.PRECISION_FP32_STRICT}[args.precision]
    n_img = 2 * args.pairs
    base_seed = shard_seed(args.pairs, rank)
    texs = (synth_batch if args.content == "structured" else random_batch)(n_img, inW, inH, dtype, dev, base_seed)
    outs = torch.empty((n_img, outH, outW, 4), dtype=dtype, device=dev)
    pp = A.PostProcessor(fsr_enabled=1, use_nis=use_nis, out_width=outW, out_height=outH, sharpness=sharpness, radius=radius,
                         precision=prec, fused=args.fused, quantize_intermediate=1, device=local_rank)

    def step():
        pp.apply_batch(texs, outs, first_eye=A.EYE_LEFT, alternate_eyes=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # clock ramp: a cold MI355X takes ~50 ms of sustained load to reach its steady clocks (20 steps after 3 warm-up
    # steps read 5 % low); run the step for a fixed wall time first so that short --warmup values measure the same
    # steady state as long ones.  Untimed, before the W warm-up steps; recorded in the JSON line.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < CLOCK_RAMP_S:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    dt = max_over_ranks(timed_region(step, args.steps, barrier), world, dev)

    pairs_total = args.pairs * world * args.steps
    value = pairs_total / dt
    bpp = texs.element_size() * 4
    algo_bytes_eye = bpp * (inW * inH + outW * outH)  # pipeline compulsory traffic per eye

    # ---- roofline of the dominant kernel, timed live with HIP events on the launch stream ---------
    stream = torch.cuda.current_stream(dev)
    roof = None
    if rank == 0:
        ms_step = time_events(step, max(5, args.steps // 2), stream)
        masked_fsr = (radius < 2.0) and not use_nis
        sharpen_only = (inW, inH) == (outW, outH)   # renderScale 1: RCAS / NVSharpen alone
        if masked_fsr or use_nis or sharpen_only:  # single-pass forms: the step is the kernel (+ its concurrent companion when masked)
            # masked EASU+RCAS runs as one mask-sorted pipeline (tiles touching the radius: EASU+RCAS or the fused kernel;
            # the rest written in final form by a concurrent kernel): the step itself is the dominant "kernel"
            ms_easu = ms_step
            easu_bytes = algo_bytes_eye * n_img
        else:
            # dominant kernel: EASU of the two-pass pipeline (NVScaler for NIS) -- launch it alone over the same batch
            cfge = A.Config.default(fsr_enabled=1, use_nis=use_nis, out_width=outW, out_height=outH, sharpness=sharpness,
                                    radius=radius, precision=prec, stage_mask=1)
            pe = A.PostProcessor(cfg=cfge, device=local_rank)
            ms_easu = time_events(lambda: pe.apply_batch(texs, outs, first_eye=A.EYE_LEFT, alternate_eyes=True),
                                  max(5, args.steps // 2), stream)
            pe.close()
            easu_bytes = bpp * (inW * inH + outW * outH) * n_img
        ach = easu_bytes / (ms_easu * 1e-3) / 1e9
        copy_gbps = copy_ceiling_gbps(dev)
        rgba8 = dtype == torch.uint8
        if sharpen_only:
            kname = "nis_sharpen_kernel" if use_nis else "rcas_direct_kernel"
        elif use_nis:   # NVScaler (+ the DirectCopy kernel of the groups outside the radius, concurrent)
            kname = "nis_scaler_kernel" + (("+outside_rgba8_kernel" if rgba8 else "+nis_outside_kernel") if radius < 2.0 else "")
        elif masked_fsr:  # tiles touching the radius: EASU+RCAS (RGBA8) or the fused kernel; the rest in final form, concurrent
            kname = "easu_fast_kernel+rcas_direct_kernel+outside_rgba8_kernel" if rgba8 else "fused_kernel+easu_outside_kernel"
        else:
            kname = "easu_fast_kernel"
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(args.workload, kname, n_img),
                "launch_ms": round(ms_easu, 4), "algorithmic_bytes_per_launch": easu_bytes,
                "copy_ceiling": round(copy_gbps, 1), "frac_of_copy": round(ach / copy_gbps, 4),
                "pipeline_ms_per_step_events": round(ms_step, 4),
                "pipeline_achieved_GBps": round(algo_bytes_eye * n_img / (ms_step * 1e-3) / 1e9, 1),
                "note": "the kernel is VALU-issue-bound on this chip (see DESIGN.md); frac is reported against the HBM roof the contract names"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not use_nis and dtype == torch.uint8:  # CPU baseline: N=1 only
        cpu = cpu_baseline(inW, inH, outW, outH, sharpness)

    if rank == 0:
        line = {
            "metric": "stereo eye-pairs/sec at 1683x1869->2244x2492 (EASU+RCAS)" if args.workload == "C2"
                      else "stereo eye-pairs/sec (%s)" % args.workload,
            "value": round(value, 2), "unit": "eye-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "fp16": "f16", "strict": "f32"}[args.precision],
            "data": "synthetic (%s)" % args.content,
            "config": {"workload": "%s: stereo pairs %dx%d->%dx%d %s, %s, sharpness 0.9, radius %.1f"
                                   % (args.workload, inW, inH, outW, outH, "RGBA8" if dtype == torch.uint8 else "RGBA16F",
                                      ("NIS NVSharpen" if (inW, inH) == (outW, outH) else "NIS NVScaler") if use_nis else
                                      ("RCAS" if (inW, inH) == (outW, outH) else "EASU+RCAS (UNORM8 intermediate)"), radius),
                       "pairs_per_gpu_per_step": args.pairs, "precision": args.precision, "clock_ramp_s": CLOCK_RAMP_S,
                       "parallelism": "batch sharded over %d GPU(s), no collective" % world},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    pp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
