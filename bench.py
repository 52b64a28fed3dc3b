#!/usr/bin/env python3
"""bench.py -- stereo eye-pairs/sec of the EASU+RCAS hot path at 1683x1869 -> 2244x2492 (BASELINE.json
config C2), achieved fraction of the HBM roofline for the dominant kernel (with the roof that actually binds
it, VALU issue, measured beside it), and the CPU oracle timed in the same run.

  python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (PostProcessor::Apply for both eyes of every pair) over one batch of
`--pairs` synthetic stereo pairs per GPU that are already resident in HBM.  Batches shard embarrassingly
(SURVEY.md 8e): every GPU owns its own sub-batch, its own ctx and its own stream; nothing is exchanged, no RCCL.

Two ways to run N GPUs, same shards, same JSON line:
  * direct      `python bench.py --gpus N`: ONE process, N ctxs (ovrfsr_create(dev_i)), one host thread and one
                stream per device, per-device HIP-event timing, host max-join.
  * torchrun    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: one rank per GPU; the ranks
                only meet at the timing barrier and the max-reduce of one scalar, over gloo (host sockets).
Documented scale configuration (BASELINE C4: 1024 pairs over 8 GPUs = 128 resident pairs per GPU):
  python bench.py --gpus 8 --workload C4 --pairs 128 --steps 10 --warmup 3
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

CLOCK_RAMP_S = 0.3      # untimed sustained load before the warm-up steps (see run_job)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~5500 reached by a plain device copy
# VALU issue peak: 1024 SIMD-32 units x 32 lanes x 2.4 GHz = lane-instructions per second if every wave64
# instruction issued in 2 cycles (MI355X_MICROARCH.md: v_fma_f32 = 2 cycles per wave64 on a SIMD-32)
VALU_PEAK_LANE_INSTR = 1024 * 32 * 2.4e9

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
    # one SHARED side-by-side texture per frame holding both eyes (what games that submit a single texture with half-width
    # bounds do: PostProcessor.cpp:146,155-158,298-301), through ovrfsr_apply_batch_shared: a "pair" is ONE 3366x1869 image
    "C2sbs": (3366, 1869, 4488, 2492, torch.uint8, 2.0, 0),
    "C2sbsr": (3366, 1869, 4488, 2492, torch.uint8, 0.5, 0),  # the same with the shipped radius 0.5: two mask centres per image
}
# launcher tests only (tests/mock_shard.py: the per-shard parity leg runs the real oracle on a tiny image); not a --workload choice
MOCK_WORKLOAD = "Tmock"
WORKLOADS[MOCK_WORKLOAD] = (48, 40, 64, 53, torch.uint8, 2.0, 0)
SHARED = {"C2sbs", "C2sbsr"}   # workloads whose images hold both eyes (images per pair = 1 instead of 2)
SHARPNESS = 0.9


def images_per_pair(workload):
    return 1 if workload in SHARED else 2


# ------------------------------------------------------------------------------------------------
# synthetic inputs, generated on the device that consumes them
# ------------------------------------------------------------------------------------------------
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


_NATURAL = {}


def natural_batch(n_img, w, h, dtype, device, base_seed):
    """NATURAL content (round 6): the three 256x256 fixtures of tests/golden/natural_*.npz -- rendered game art, a rendered UI with text, a
    photograph (provenance: tests/golden/make_natural.py) -- mirror-tiled to the eye size on the consuming GPU, image i from fixture
    (seed + i) % 3 at a seed-dependent offset (the same tiling as tests/natural.py)."""
    if device not in _NATURAL:
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
        _NATURAL[device] = [torch.from_numpy(np.load(os.path.join(gold, "natural_%s.npz" % k))["rgba8"]).to(device) for k in ("cube", "portal", "hopper")]
    out = torch.empty((n_img, h, w, 4), dtype=dtype, device=device)
    for i in range(n_img):
        seed = (base_seed + i) & 0xffff
        src = _NATURAL[device][seed % 3]
        n = src.shape[0]

        def index(length, off):
            j = (torch.arange(length, device=device) + off) % (2 * n)
            return torch.where(j < n, j, 2 * n - 1 - j)

        img = src[index(h, (seed * 101) % n)][:, index(w, (seed * 37) % n)]
        if dtype == torch.uint8:
            out[i] = img
        else:
            out[i] = (img.float() / 255.0).to(dtype)
            out[i, ..., 3] = 1.0
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


def shard_seed(pairs_per_gpu, shard):
    """Seed of the first eye image of shard `shard`: image g (global index) has seed 0x5EED0000 + g, i.e.
    0x5EED0000 + 2*pair + eye; shard s owns global pairs [s*P, (s+1)*P)."""
    return 0x5EED0000 + 2 * pairs_per_gpu * shard


# ------------------------------------------------------------------------------------------------
# shards: one per GPU.  A shard owns its inputs, outputs, ctx; step() is asynchronous on its device.
# ------------------------------------------------------------------------------------------------
class GpuShard:
    """The sub-batch of one GPU: `pairs` stereo pairs generated on that GPU, a ctx created on it
    (ovrfsr_create(device)), outputs resident there.  Nothing of a shard ever leaves its device."""

    def __init__(self, device_index, shard_index, args):
        import openvr_fsr_amd as A
        self.A = A
        self.device_index, self.shard_index = device_index, shard_index
        self.dev = torch.device("cuda", device_index)
        inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
        prec = {"fp32": A.PRECISION_FP32, "strict": A.PRECISION_FP32_STRICT}[args.precision]
        self.shared = args.workload in SHARED
        self.n_img = images_per_pair(args.workload) * args.pairs
        with torch.cuda.device(self.dev):
            gen = synth_batch if args.content == "structured" else random_batch
            self.texs = gen(self.n_img, inW, inH, dtype, self.dev, shard_seed(args.pairs, shard_index))
            self.outs = torch.empty((self.n_img, outH, outW, 4), dtype=dtype, device=self.dev)
            self.cfg_kw = dict(fsr_enabled=1, use_nis=use_nis, out_width=outW, out_height=outH, sharpness=SHARPNESS, radius=radius,
                               precision=prec, fused=args.fused, quantize_intermediate=1)
            self.pp = A.PostProcessor(device=device_index, **self.cfg_kw)
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def bind(self):
        torch.cuda.set_device(self.dev)     # per host thread

    def step(self):
        self.pp.apply_batch(self.texs, self.outs, first_eye=self.A.EYE_LEFT, alternate_eyes=True, shared=self.shared)

    def fetch(self, indices):
        """(input, output) host copies of the given images of the batch the timed call just processed."""
        self.sync()
        return [(self.texs[i].cpu().numpy(), self.outs[i].cpu().numpy()) for i in indices]

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def mark_start(self):
        self.ev0.record(torch.cuda.current_stream(self.dev))

    def mark_end(self):
        self.ev1.record(torch.cuda.current_stream(self.dev))

    def device_ms(self):
        self.ev1.synchronize()
        return self.ev0.elapsed_time(self.ev1)

    def close(self):
        self.pp.close()


# OVRFSR_BENCH_TIMED_GAP_MS (profiling aid, default off): milliseconds of host sleep right before and right after the timed steps, outside the
# timed region.  The device idles for that long, which brackets the timed dispatches in a rocprofv3 kernel trace (tools/profile.sh sets 3).
TIMED_GAP_S = float(os.environ.get("OVRFSR_BENCH_TIMED_GAP_MS", "0") or 0) * 1e-3


def build_shards(n_local, first_shard, make):
    """The launcher's partition: shard i of this process = global shard first_shard + i on local device i."""
    return [make(i, first_shard + i) for i in range(n_local)]


def run_local(shards, steps, warmup, ramp_s=0.0, cross_barrier=None):
    """Run `warmup` untimed + EXACTLY `steps` timed steps on every local shard concurrently (one host thread per
    device; one shard runs inline).  The timed region is bracketed by sync + barrier on both sides; returns
    (wall seconds of the slowest shard, [device milliseconds per shard from HIP events on the launch stream])."""
    n = len(shards)
    gate = threading.Barrier(n)
    t_start, t_end, dev_ms, errors = [0.0] * n, [0.0] * n, [0.0] * n, []

    def all_ranks():
        if cross_barrier is not None:
            cross_barrier()

    def work(i):
        try:
            s = shards[i]
            s.bind()
            # clock ramp: a cold MI355X takes ~50 ms of sustained load to reach its steady clocks; run the step for a
            # fixed wall time first so that short --warmup values read the same steady state as long ones.  Untimed.
            t = time.perf_counter()
            while time.perf_counter() - t < ramp_s:
                s.step()
                s.sync()
            for _ in range(warmup):
                s.step()
            s.sync()
            gate.wait()
            if i == 0:
                all_ranks()
            gate.wait()
            if TIMED_GAP_S:
                time.sleep(TIMED_GAP_S)   # profiling aid (tools/profile.sh): an idle gap in the kernel trace in front of the timed steps
            t_start[i] = time.perf_counter()
            s.mark_start()
            for _ in range(steps):
                s.step()
            s.mark_end()
            s.sync()
            if TIMED_GAP_S:
                time.sleep(TIMED_GAP_S)   # ... and behind them (outside the timed region on both sides)
            gate.wait()
            if i == 0:
                all_ranks()
            gate.wait()
            t_end[i] = time.perf_counter()
            dev_ms[i] = s.device_ms()
        except BaseException as e:  # noqa: BLE001 -- reported by the caller; never leave the other threads at the gate
            errors.append(e)
            gate.abort()

    if n == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return max(t_end) - min(t_start), dev_ms


def timed_region(step, steps, barrier):
    """EXACTLY `steps` calls of step() bracketed by barrier()+sync on both sides; returns local wall seconds.
    (The single-shard form of run_local, kept for the torchrun path's CPU test.)"""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(dt, world, device=None):
    """Host-side max-join of one scalar per rank (gloo: the data path has no collective, so the timing join does not
    need RCCL either)."""
    if world == 1:
        return dt
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------
# self-check of the timed call: the oracle on the very images the timed launch read and wrote
# ------------------------------------------------------------------------------------------------
def oracle_expected(workload, img, image_index, nthreads=0):
    """What the reference's pipeline produces for image `image_index` of a shard (eye = index & 1; a shared side-by-side
    texture holds both eyes), evaluated by the CPU oracle on the host copy of the GPU's own input.  Same dtype as the
    GPU output of that workload.  Test infrastructure: only called by the parity_check / cpu_baseline legs."""
    from oracle import oracle as O
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[workload]
    eye = 0 if workload in SHARED else image_index & 1
    one_eye = workload not in SHARED
    centre, rad = O.mask_constants(outW, outH, radius, (0.5, 0.5, 0.5, 0.5), one_eye, eye)
    same_size = (inW, inH) == (outW, outH)
    if use_nis:
        import openvr_fsr_amd as A
        f = O.unorm8_to_float(img)
        if same_size:
            ok, cfg = A.nis_sharpen_config(SHARPNESS, inW, inH)
            return O.float_to_unorm8(O.nis_sharpen(f, O.nis_block(cfg, centre, rad, 0), nthreads=nthreads))
        cs, cu = A.nis_coefs()
        ok, cfg = A.nis_scaler_config(SHARPNESS, inW, inH, outW, outH)
        return O.float_to_unorm8(O.nis_upscale(f, outW, outH, O.nis_block(cfg, centre, rad, 0), cs, cu, nthreads=nthreads))
    if dtype == torch.uint8:
        if same_size:
            return O.float_to_unorm8(O.rcas(O.unorm8_to_float(img), O.rcas_con(SHARPNESS), centre, rad, nthreads=nthreads))
        return O.fsr_pipeline_u8(img, outW, outH, sharpness=SHARPNESS, radius=radius, eye=eye, one_eye_per_texture=one_eye, nthreads=nthreads)
    # RGBA16F pipeline (C5): EASU in fp32 -> half intermediate -> RCAS in fp32 -> half store
    e = O.easu(img.astype(np.float32), outW, outH, O.easu_con(inW, inH, outW, outH), centre, rad, nthreads=nthreads)
    e16 = e.astype(np.float16).astype(np.float32)
    return O.rcas(e16, O.rcas_con(SHARPNESS), centre, rad, nthreads=nthreads).astype(np.float16)


def compare_images(got, want):
    """Difference record of one image: byte outputs in LSB, half/float outputs in max-abs (unit domain)."""
    r = {"n_total": int(want.size)}
    if want.dtype == np.uint8:
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        r.update(max_lsb=int(d.max()), n_diff=int((d != 0).sum()), n_gt1=int((d > 1).sum()))
    else:
        g, w = got.astype(np.float32), want.astype(np.float32)
        d = np.abs(g - w)
        r.update(max_abs=float(np.max(d)), n_diff=int((got != want).sum()), n_gt_1e3=int((d > 1e-3).sum()),
                 n_nan=int(np.isnan(g).sum() + np.isnan(w).sum()))
    return r


def check_indices(n_img):
    """Images {0, 1, n-2, n-1} of the batch: the first and the last stereo pair (blockIdx.z = 0 and the largest)."""
    return sorted({i for i in (0, 1, n_img - 2, n_img - 1) if 0 <= i < n_img})


def image_ok(args, r):
    """The tolerance of the contract on one comparison record: bit-exact for the strict build, <= 1 LSB on UNORM8 outputs,
    max-abs <= 1e-3 with no value above it (and no NaN) on half outputs."""
    if args.precision == "strict":
        return r["n_diff"] == 0
    return r["max_lsb"] <= 1 if "max_lsb" in r else (r["n_gt_1e3"] == 0 and r["n_nan"] == 0)


def shard_parity(args, shard):
    """Image 0 of ONE shard's timed batch (input and output downloaded after the timed loop) against the oracle: every device of
    a multi-GPU run proves its own outputs -- the per-device ctx, the per-(kernel, device) LDS attribute and the DeviceGuard paths
    have no other witness on device != 0.  Test infrastructure, like parity_and_cpu."""
    wl = getattr(shard, "parity_workload", args.workload)
    try:
        (src, got), = shard.fetch([0])
        r = compare_images(got, oracle_expected(wl, src, 0))
        r["ok"] = bool(image_ok(args, r))
    except Exception as e:  # noqa: BLE001 -- a shard that cannot be checked is a failed shard, reported by index
        r = {"ok": False, "error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    r.update(shard=shard.shard_index, device=shard.device_index, image=0)
    return r


def parity_and_cpu(args, fetched, indices, want_cpu, budget_s=8.0, max_pairs=8):
    """The oracle on the images the TIMED launch processed (host copies of shard 0's inputs, downloaded after the timed
    loop) against the outputs that launch wrote: `parity_check`.  The same oracle calls are the `cpu_baseline` sample --
    both legs process identical data -- continued on the first pair until ~budget_s of wall time have passed."""
    from oracle import oracle as O
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    cores = min(usable_cores(), O.lib().ovo_max_threads())
    ipp = images_per_pair(args.workload)
    recs, imgs_done, dt = [], 0, 0.0   # dt: wall time inside oracle calls only (the comparisons are not CPU-baseline work)

    def timed_oracle(src, i, threads):
        nonlocal dt, imgs_done
        t = time.perf_counter()
        want = oracle_expected(args.workload, src, i, threads)
        dt += time.perf_counter() - t
        imgs_done += 1
        return want

    for (src, got), i in zip(fetched, indices):
        r = compare_images(got, timed_oracle(src, i, cores))
        r["image"] = i
        recs.append(r)
    if want_cpu:
        while imgs_done < ipp * max_pairs and dt < budget_s:
            for k in range(ipp):
                timed_oracle(fetched[k][0], indices[k], cores)
    strict = args.precision == "strict"
    par = {"images": len(recs), "image_indices": indices, "n_total": sum(r["n_total"] for r in recs), "n_diff": sum(r["n_diff"] for r in recs),
           "call": "the timed ovrfsr_apply_batch%s of shard 0 (%d images per launch); inputs and outputs downloaded after the timed loop, "
                   "oracle/liboracle.so run on those inputs" % ("_shared" if args.workload in SHARED else "", images_per_pair(args.workload) * args.pairs)}
    if "max_lsb" in recs[0]:
        par.update(max_lsb=max(r["max_lsb"] for r in recs), n_gt1=sum(r["n_gt1"] for r in recs))
        par["tolerance"] = "bit-exact (strict build)" if strict else "<= 1 LSB on UNORM8 outputs"
        par["ok"] = (par["n_diff"] == 0) if strict else (par["max_lsb"] <= 1)
    else:
        par.update(max_lsb=None, max_abs=max(r["max_abs"] for r in recs), n_gt_1e3=sum(r["n_gt_1e3"] for r in recs), n_nan=sum(r["n_nan"] for r in recs))
        par["tolerance"] = "bit-exact (strict build)" if strict else "max-abs <= 1e-3 on half outputs, no value above it"
        par["ok"] = (par["n_diff"] == 0) if strict else (par["n_gt_1e3"] == 0 and par["n_nan"] == 0)
    cpu = None
    if want_cpu:
        t1 = time.perf_counter()   # and one image on a single thread (report single-thread too)
        oracle_expected(args.workload, fetched[0][0], indices[0], 1)
        single = (1.0 / ipp) / (time.perf_counter() - t1)
        pairs = imgs_done / float(ipp)
        cpu = {"value": round(pairs / dt, 4), "unit": "eye-pairs/s", "cores": cores, "kind": "port", "single_thread_value": round(single, 4),
               "sample": "%.1f stereo pair(s) of workload %s (%dx%d->%dx%d), the images the timed GPU launch processed (downloaded), oracle/liboracle.so with "
                         "%d OpenMP threads (usable cores of this container), %.2f s wall = %.1f core-seconds"
                         % (pairs, args.workload, inW, inH, outW, outH, cores, dt, dt * cores)}
    return par, cpu


# ------------------------------------------------------------------------------------------------
# roofline helpers
# ------------------------------------------------------------------------------------------------
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


def dominant_kernels(workload):
    """Kernel name(s) the roofline object is about (substring match on rocprofv3's Kernel_Name)."""
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[workload]
    rgba8 = dtype == torch.uint8
    if (inW, inH) == (outW, outH):
        return ["nis_sharpen_kernel"] if use_nis else ["rcas_dpp_kernel"]
    if use_nis:   # NVScaler (+ the DirectCopy kernel of the groups outside the radius, concurrent)
        return ["nis_scaler_kernel"] + ((["outside_staged_kernel"] if rgba8 else ["nis_outside_kernel"]) if radius < 2.0 else [])
    if radius < 2.0:  # tiles touching the radius: EASU+RCAS (RGBA8) or the fused kernel; the rest in final form, concurrent
        # (RCAS of the mask-sorted form: rcas_dpp_kernel on segments of the inside runs; rcas_direct_kernel under OVRFSR_RCAS_DPP=0)
        rcas = "rcas_direct_kernel" if os.environ.get("OVRFSR_RCAS_DPP", "1")[:1] == "0" else "rcas_dpp_kernel"
        return ["easu_fast_kernel", rcas, "outside_staged_kernel"] if rgba8 else ["fused_kernel", "easu_outside_kernel"]
    return ["easu_fast_kernel"]


# datasheet issue cycles per wave64 VALU instruction on a SIMD32 (MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles"; packed,
# conversion, min/max/med3, 3-operand integer and DPP forms at half rate; transcendentals at quarter rate)
DATASHEET_COST = {"fast": 2.0, "slow": 4.0, "pk": 4.0, "trans": 8.0}

PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"],
              # instruction counts per category and waves: what tools/isa_costs.py turns into VALU issue cycles (roofline.valu.issue)
              ["SQ_WAVES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VALU_TRANS_F32"]]
PMC_FULL_NAMES = {}   # short kernel name -> set of full (demangled) names seen in the counter passes
PMC_CHILD_PAIRS = 4


def pmc_counters(args, timeout_s=120):
    """HBM traffic and VALU counters of this workload's kernels, measured IN THIS RUN: bench.py re-runs itself (a few
    steps, `--pmc-child`) under `rocprofv3 --pmc <counters>`, one pass per counter group, counters only (no trace
    options -- the guide's recipe).  Returns {kernel name: {counter: mean per dispatch}} with images per dispatch, or
    None if rocprofv3 is not usable here."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    import csv
    agg = {}
    failed = False
    for counters in PMC_PASSES:
        if failed:
            break
        tmp = tempfile.mkdtemp(prefix="ovrfsr_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                                           "--pmc-child", "--workload", args.workload, "--content", args.content, "--precision", args.precision,
                                           "--fused", str(args.fused), "--pairs", str(PMC_CHILD_PAIRS)]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.endswith("counter_collection.csv")]
            for fn in files:
                for r in csv.DictReader(open(fn)):
                    k = r.get("Kernel_Name", "")
                    if "ovrfsr" not in k:
                        continue
                    m = re.search(r"ovrfsr_\w+::(\w+)", k)
                    short = m.group(1) if m else k
                    PMC_FULL_NAMES.setdefault(short, set()).add(k)
                    agg.setdefault(short, {}).setdefault(r.get("Counter_Name"), []).append(float(r.get("Counter_Value", 0)))
        except (subprocess.SubprocessError, OSError, ValueError):
            failed = len(agg) == 0   # the first pass is lost: do not spend more wall time on a profiler that is not usable here
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    if not agg:
        return None
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def pmc_child(args):
    """What runs under rocprofv3 --pmc: the same step on a small batch, a few times; prints nothing."""
    args.pairs = PMC_CHILD_PAIRS
    s = GpuShard(0, 0, args)
    s.bind()
    for _ in range(3):
        s.step()
    s.sync()
    s.close()


def profile_traffic(workload, kernels, n_img):
    """Fallback when counters cannot be read in this run: the committed rocprofv3 PMC summary of this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_per_eye.json")
    try:
        per_eye = json.load(open(path))[workload]
        hit = [v for k, v in per_eye.items() if any(k.endswith("::" + part) for part in kernels)]
        return int(sum(h["hbm_bytes_per_eye"] for h in hit) * n_img) if hit else None
    except (OSError, KeyError, ValueError):
        return None


def issue_roof(pmc, kernels, scale, out_px, ms_dom, sclk_mhz):
    """The roof these kernels sit on, from THIS run's counters and the shipped code objects: VALU issue cycles.
    openvr_fsr_amd/kernel_issue_costs.json (tools/isa_costs.py, emitted by the build from llvm-objdump of the library) holds
    every kernel's basic blocks with their instruction counts per issue class and the edges between them; how often each block
    executes is bounded by linear programming from the per-wave instruction counts measured here (SQ_INSTS_VALU / LDS /
    VMEM_RD / VMEM_WR [/ TRANS / SMEM / BRANCH / SALU] over SQ_WAVES) under flow conservation.  The result is an interval:
      cycles_per_64px  issue cycles per 64 output pixels (class costs in true shader cycles: profiles/r04_valu_issue_rates.txt)
      issue_frac       issue cycles x waves / (1024 SIMDs x measured sclk x kernel time): ~1 = the VALU issues back to back."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_costs
        doc = isa_costs.load()
    except (OSError, ValueError, ImportError) as e:
        return {"error": "no issue-cost table (%s): run __graft_entry__.build()" % e}
    per_kernel, lo_sum, hi_sum, ds_lo, ds_hi = {}, 0.0, 0.0, 0.0, 0.0
    for k in kernels:
        v = pmc.get(k)
        full = sorted(PMC_FULL_NAMES.get(k, ()))
        if not v or len(full) != 1 or not v.get("SQ_WAVES") or not v.get("SQ_INSTS_VALU"):
            per_kernel[k] = {"error": "counters or a unique instantiation missing", "instantiations": full}
            continue
        cfg = isa_costs.find_kernel(doc, full[0])
        if cfg is None:
            per_kernel[k] = {"error": "not in kernel_issue_costs.json", "name": full[0]}
            continue
        w = v["SQ_WAVES"]
        pw = {"valu": v["SQ_INSTS_VALU"] / w}
        for key, ctr in (("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR"), ("trans", "SQ_INSTS_VALU_TRANS_F32"),
                         ("smem", "SQ_INSTS_SMEM"), ("branch", "SQ_INSTS_BRANCH"), ("salu", "SQ_INSTS_SALU")):
            if ctr in v:
                pw[key] = v[ctr] / w
        b = isa_costs.issue_bounds(cfg, pw)
        if b is None:
            per_kernel[k] = {"error": "no block-execution profile fits the counters", "per_wave": {a: round(x, 3) for a, x in pw.items()}}
            continue
        waves = w * scale
        lo_sum += b["lo"] * waves
        hi_sum += b["hi"] * waves
        # the same execution profiles priced at DATASHEET issue costs (MI355X_MICROARCH.md, execution model: a wave64 VALU instruction
        # issues over 2 cycles on its SIMD32; packed / conversion / 3-operand / DPP forms over 4; transcendentals over 8): model-free
        d = isa_costs.issue_bounds(cfg, pw, cost=DATASHEET_COST)
        if d is not None:
            ds_lo += d["lo"] * waves
            ds_hi += d["hi"] * waves
        per_kernel[k] = {"kernel": full[0], "waves_per_launch": int(waves), "valu_instr_per_wave": round(pw["valu"], 1),
                         "per_wave_counters": {a: round(x, 4) for a, x in pw.items()},
                         "issue_cycles_per_wave": [round(b["lo"], 1), round(b["hi"], 1)], "mean_cycles_per_valu_instr": [round(b["mean_cost_lo"], 3), round(b["mean_cost_hi"], 3)],
                         "counters_used": b["constraints"], "tolerance": b["tolerance"], "blocks": len(cfg["blocks"])}
    if hi_sum == 0.0:
        return {"per_kernel": per_kernel}
    out = {"cycles_per_64px": [round(lo_sum / (out_px / 64.0), 1), round(hi_sum / (out_px / 64.0), 1)], "per_kernel": per_kernel,
           "class_costs_true_cycles": isa_costs.COST, "sclk_mhz": round(sclk_mhz, 1) if sclk_mhz else None,
           "method": "tools/isa_costs.py: CFG + per-class instruction counts from llvm-objdump of the shipped library; block executions bounded by LP from this run's "
                     "per-wave counters under flow conservation (interval = every profile the counters allow)",
           "model_error": "class costs are those of homogeneous instruction streams, added up; mixed streams measured in profiles/r04_valu_issue_rates.txt cost "
                          "0.69-0.81 (fast+slow alternating) to 1.18 (fast+packed) of the sum, an EASU-shaped mix 1.06: read 0.9-1.1 as 'issues back to back'"}
    if sclk_mhz:
        cap = 1024.0 * sclk_mhz * 1e6 * ms_dom * 1e-3   # SIMD-cycles available in the launch
        out["issue_frac"] = [round(lo_sum / cap, 3), round(hi_sum / cap, 3)]
        if ds_hi > 0.0:
            out["datasheet_issue_frac"] = [round(ds_lo / cap, 3), round(ds_hi / cap, 3)]
            out["datasheet_costs_cycles"] = DATASHEET_COST
    return out


def roofline(args, shard, timed_ms_step=None):
    """The dominant kernel against the HBM roof the contract names, and against the roof that binds it (VALU issue).
    timed_ms_step: HIP-event milliseconds per step of shard 0 INSIDE the timed region (GpuShard.mark_start / mark_end bracket exactly the
    `steps` timed calls): the pipeline figure of the line is that one, so it can never exceed the wall-clock ms_per_step it sits beside
    (round 5 re-timed the step in a later pass, which read 1.5 % more than the timed loop on a different clock state)."""
    A = shard.A
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    dev, n_img = shard.dev, shard.n_img
    bpp = shard.texs.element_size() * 4
    algo_bytes_eye = bpp * (inW * inH + outW * outH)  # pipeline compulsory traffic per eye
    stream = torch.cuda.current_stream(dev)
    iters = max(5, args.steps // 2)
    ms_step = timed_ms_step if timed_ms_step else time_events(shard.step, iters, stream)
    kernels = dominant_kernels(args.workload)
    single_pass = len(kernels) > 1 or use_nis or (inW, inH) == (outW, outH)
    sclk = SclkSampler(shard.device_index)   # shader clock while the dominant kernel runs (valu.issue_frac needs cycles, not nominal GHz)
    if single_pass:
        # masked / NIS / sharpen-only: the step IS the kernel (plus its concurrent companions when masked)
        with sclk:
            ms_dom = time_events(shard.step, max(iters, 10), stream)
        dom_bytes, out_px = algo_bytes_eye * n_img, outW * outH * n_img
    else:
        # dominant kernel of the two-pass pipeline: EASU -- launched alone over the same batch
        kw = dict(shard.cfg_kw, stage_mask=1)
        pe = A.PostProcessor(cfg=A.Config.default(**kw), device=shard.device_index)
        with sclk:
            ms_dom = time_events(lambda: pe.apply_batch(shard.texs, shard.outs, first_eye=A.EYE_LEFT, alternate_eyes=True, shared=shard.shared),
                                 max(iters, 10), stream)
        pe.close()
        dom_bytes, out_px = bpp * (inW * inH + outW * outH) * n_img, outW * outH * n_img
    # (an in-kernel probe -- one sleeping wave on a side stream comparing s_memtime with the 100 MHz counter -- was tried in round 4:
    # a second active hardware queue slows the timed kernel by 30-45 %, so the clock it reads is not the clock of the undisturbed run)
    sclk_mhz = sclk.mean_mhz()
    ach = dom_bytes / (ms_dom * 1e-3) / 1e9
    pipe = algo_bytes_eye * n_img / (ms_step * 1e-3) / 1e9
    copy_gbps = copy_ceiling_gbps(dev)
    roof = {"bound": "hbm", "kernel": "+".join(kernels), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": None,
            "launch_ms": round(ms_dom, 4), "algorithmic_bytes_per_launch": dom_bytes,
            "copy_ceiling": round(copy_gbps, 1), "frac_of_copy": round(ach / copy_gbps, 4),
            "pipeline_ms_per_step_events": round(ms_step, 4),
            "pipeline_events_source": "HIP events around the timed loop itself (shard 0)" if timed_ms_step else "a separate event-timed pass",
            "pipeline_achieved_GBps": round(pipe, 1),
            "pipeline_frac": round(pipe / HBM_PEAK_GBPS, 4), "binding_roof": "valu_issue", "valu": None,
            "sclk_mhz": round(sclk_mhz, 1) if sclk_mhz else None, "sclk_source": sclk.source, "sclk_samples": len(sclk.samples),
            # socket power while the dominant kernel ran, and the cap it runs under: for a VALU-issue-bound kernel the clock IS the roof, and
            # the clock is what the power cap leaves (a reader can tell a capped 2.1 GHz from a quiet 2.4 GHz box)
            "power_w": round(sclk.mean_power_w(), 1) if sclk.mean_power_w() else None, "power_cap_w": sclk.power_cap_w,
            "power_source": sclk.power_source}
    pmc = pmc_counters(args) if args.pmc == "auto" else None
    if pmc:
        child_img = images_per_pair(args.workload) * PMC_CHILD_PAIRS
        scale = n_img / float(child_img)   # counters were read on launches of child_img images
        hit = {k: v for k, v in pmc.items() if k in kernels}
        if hit and all("FETCH_SIZE" in v and "WRITE_SIZE" in v for v in hit.values()):
            # WRITE_SIZE + 2 x FETCH_SIZE, KiB per dispatch (gfx950: FETCH_SIZE reports half the bytes of a coalesced read)
            roof["traffic"] = int(sum(v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"] for v in hit.values()) * 1024.0 * scale)
            roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (launches of %d images, scaled to %d)" % (child_img, n_img)
            roof["traffic_per_kernel"] = {k: int((v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"]) * 1024.0 * scale) for k, v in hit.items()}
        if hit and all("SQ_INSTS_VALU" in v for v in hit.values()):
            instr = sum(v["SQ_INSTS_VALU"] for v in hit.values()) * scale          # wave-instructions per launch (step)
            lane_rate = instr * 64.0 / (ms_dom * 1e-3)
            busy = None
            if len(hit) == 1:
                v = next(iter(hit.values()))
                if v.get("GRBM_GUI_ACTIVE"):   # SQ_ACTIVE_INST_* tick in quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
                    busy = round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
            roof["valu"] = {"instr_per_64px": round(instr / (out_px / 64.0), 1), "lane_instr_per_s": round(lane_rate, 0),
                            "lane_instr_peak_if_every_op_took_2_cycles": VALU_PEAK_LANE_INSTR,
                            "instr_count_over_2cycle_peak": round(lane_rate / VALU_PEAK_LANE_INSTR, 4), "valu_active_ratio": busy,
                            "per_kernel_instr_per_64px": {k: round(v["SQ_INSTS_VALU"] * scale / (out_px / 64.0), 1) for k, v in hit.items()},
                            "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE pass of this run; "
                                      "valu_active_ratio = 4*SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): NOT a fraction -- the quad-cycle counter counts every "
                                      "instruction's full issue time, so ops slower than 4 cycles push it past 1 (1.0-1.3 on these kernels = the VALU never idles); "
                                      "instr_count_over_2cycle_peak is a COUNT ratio (1024 SIMD-32 x 32 lanes x 2.4 GHz if every op issued in 2 cycles; real ops "
                                      "take 2.5-8), not a utilisation: the utilisation figure is valu.issue.issue_frac"}
    if pmc and roof["valu"] is not None:
        # optional: needs openvr_fsr_amd/kernel_issue_costs.json (tools/isa_costs.py --emit, a build step) and scipy; without
        # them the object is simply absent and nothing else of the line changes
        try:
            issue = issue_roof(pmc, kernels, scale, out_px, ms_dom, sclk_mhz)
        except Exception as e:  # noqa: BLE001
            issue = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        if "error" not in issue:
            roof["valu"]["issue"] = issue
            if issue.get("datasheet_issue_frac") is not None:
                roof["valu"]["datasheet_issue_frac"] = issue["datasheet_issue_frac"]
        else:
            roof["valu"]["issue_unavailable"] = issue["error"]
    if roof["traffic"] is None:
        roof["traffic"] = profile_traffic(args.workload, kernels, n_img)
        if roof["traffic"] is not None:
            roof["traffic_source"] = "profiles/traffic_per_eye.json (committed rocprofv3 PMC summary; not re-measured in this run)"
    roof["note"] = ("frac is against the HBM roof the contract names; the kernels are VALU-issue-bound on this chip: valu.issue.issue_frac "
                    "(issue cycles from the shipped ISA and this run's counters over SIMD-cycles at the measured sclk) is the utilisation of the "
                    "roof that binds them, valu.instr_per_64px the figure that moves with kernel work")
    return roof


class SclkSampler:
    """Mean shader clock (MHz) of one device while a loop runs: a host thread polls amdsmi (or the driver's sysfs node)
    every few milliseconds.  None when neither is readable in this container."""

    def __init__(self, device_index, period_s=0.004):
        self.idx, self.period, self.samples, self._stop, self._thr = device_index, period_s, [], False, None
        self.power_samples, self.power_cap_w, self.power_source, self._read_power = [], None, None, None
        self._read = self._probe()
        self._probe_power()

    def _probe(self):
        try:
            import amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception:  # noqa: BLE001 -- already initialised
                pass
            h = amdsmi.amdsmi_get_processor_handles()[self.idx]

            def read():
                info = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                v = info.get("clk", info.get("cur_clk"))
                return float(v) if isinstance(v, (int, float)) and v > 0 else None
            if read() is not None:
                self.source = "amdsmi_get_clock_info(GFX)"
                return read
        except Exception:  # noqa: BLE001
            pass
        try:
            import glob
            nodes = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            node = nodes[min(self.idx, len(nodes) - 1)]

            def read_sysfs():
                for line in open(node).read().splitlines():
                    if line.rstrip().endswith("*"):
                        return float(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
                return None
            if read_sysfs() is not None:
                self.source = node
                return read_sysfs
        except Exception:  # noqa: BLE001
            pass
        self.source = None
        return None

    def _probe_power(self):
        """socket power (W) and its cap: amdsmi where it is importable, else the hwmon nodes of the device"""
        def watts(v):
            return None if not isinstance(v, (int, float)) or v <= 0 else (float(v) * 1e-6 if v > 20000 else float(v))   # microwatts or watts
        try:
            import amdsmi
            h = amdsmi.amdsmi_get_processor_handles()[self.idx]

            def read():
                info = amdsmi.amdsmi_get_power_info(h)
                return watts(info.get("current_socket_power")) or watts(info.get("average_socket_power")) or watts(info.get("socket_power"))
            if read() is not None:
                self._read_power, self.power_source = read, "amdsmi_get_power_info(socket power)"
                try:
                    cap = amdsmi.amdsmi_get_power_cap_info(h)
                    self.power_cap_w = watts(cap.get("power_cap")) or watts(amdsmi.amdsmi_get_power_info(h).get("power_limit"))
                except Exception:  # noqa: BLE001
                    self.power_cap_w = watts(amdsmi.amdsmi_get_power_info(h).get("power_limit"))
                return
        except Exception:  # noqa: BLE001
            pass
        try:
            import glob
            cards = sorted(glob.glob("/sys/class/drm/card*/device"))
            base = cards[min(self.idx, len(cards) - 1)]
            nodes = glob.glob(base + "/hwmon/hwmon*/power1_average") + glob.glob(base + "/hwmon/hwmon*/power1_input")
            node = nodes[0]

            def read_sysfs():
                return watts(int(open(node).read().strip()))
            if read_sysfs() is not None:
                self._read_power, self.power_source = read_sysfs, node
                try:
                    self.power_cap_w = watts(int(open(os.path.join(os.path.dirname(node), "power1_cap")).read().strip()))
                except (OSError, ValueError):
                    pass
        except Exception:  # noqa: BLE001
            pass

    def mean_power_w(self):
        return sum(self.power_samples) / len(self.power_samples) if self.power_samples else None

    def __enter__(self):
        if self._read or self._read_power:
            def loop():
                while not self._stop:
                    try:
                        v = self._read() if self._read else None
                        if v:
                            self.samples.append(v)
                        p = self._read_power() if self._read_power else None
                        if p:
                            self.power_samples.append(p)
                    except Exception:  # noqa: BLE001
                        pass
                    time.sleep(self.period)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr:
            self._thr.join()

    def mean_mhz(self):
        return sum(self.samples) / len(self.samples) if self.samples else None


def content_random_leg(args, shard, steps):
    """The same workload and launch on the second distribution of SURVEY.md 8d -- uniform-random texels: no smooth regions,
    every NVScaler wave on its 4-direction path, dense near-tie lists -- un-timed for the headline, reported beside it.
    Image 0 of that batch is checked against the oracle as well."""
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    keep = shard.texs
    stream = torch.cuda.current_stream(shard.dev)
    try:
        shard.texs = random_batch(shard.n_img, inW, inH, dtype, shard.dev, shard_seed(args.pairs, shard.shard_index) ^ 0x00A5A500)
        ms = time_events(shard.step, max(5, steps), stream)
        (src, got), = shard.fetch([0])
        r = compare_images(got, oracle_expected(args.workload, src, 0))
        ok = (r["n_diff"] == 0) if args.precision == "strict" else (r["max_lsb"] <= 1 if "max_lsb" in r else r["n_gt_1e3"] == 0 and r["n_nan"] == 0)
        return {"content": "uniform random texels", "value": round(args.pairs / (ms * 1e-3), 2), "unit": "eye-pairs/s", "ms_per_step": round(ms, 4),
                "parity_image0": dict(r, ok=bool(ok))}
    finally:
        shard.texs = keep


def content_natural_leg(args, shard, steps):
    """The same workload and launch on NATURAL content (natural_batch): real edges, text, photographic noise -- what the guard's listed-pixel
    rate, EASU's dering clamp and NVScaler's edge classification see in use -- un-timed for the headline, reported beside it.  Image 0 of
    that batch is checked against the oracle as well."""
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    keep = shard.texs
    stream = torch.cuda.current_stream(shard.dev)
    try:
        shard.texs = natural_batch(shard.n_img, inW, inH, dtype, shard.dev, shard_seed(args.pairs, shard.shard_index))
        ms = time_events(shard.step, max(5, steps), stream)
        (src, got), = shard.fetch([0])
        r = compare_images(got, oracle_expected(args.workload, src, 0))
        ok = (r["n_diff"] == 0) if args.precision == "strict" else (r["max_lsb"] <= 1 if "max_lsb" in r else r["n_gt_1e3"] == 0 and r["n_nan"] == 0)
        return {"content": "natural (tests/golden/natural_*.npz, mirror-tiled)", "value": round(args.pairs / (ms * 1e-3), 2), "unit": "eye-pairs/s",
                "ms_per_step": round(ms, 4), "parity_image0": dict(r, ok=bool(ok))}
    finally:
        shard.texs = keep


def frame_leg(args, shard, frames=520):
    """The reference's OWN performance figure: GPU time per frame for ONE submitted stereo pair, as its debug mode logs it
    ("Average GPU processing time for upscale", PostProcessor.cpp:605-626: timestamps around every Apply, ring of 6, mean of
    500 readings, x2 when each eye has its own texture) -- read through ovrfsr_average_gpu_time_ms from `frames` stereo
    submissions (L, R, L, R ... one ovrfsr_apply per eye, as the Submit detours call it).  Beside it: the same two applies
    captured in a HIP graph and replayed (what a host that pre-records its frame would see), and issued directly."""
    A = shard.A
    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    if args.workload in SHARED or shard.n_img < 2:
        return None
    L, R, oL, oR = shard.texs[0], shard.texs[1], shard.outs[0], shard.outs[1]
    dev = shard.dev

    def one(radius_v):
        rec = {"radius": radius_v}
        pp = A.PostProcessor(device=shard.device_index, **dict(shard.cfg_kw, radius=radius_v, debug_mode=1))
        try:
            for _ in range(frames):
                pp.apply(A.EYE_LEFT, L, out=oL)
                pp.apply(A.EYE_RIGHT, R, out=oR)
            torch.cuda.synchronize(dev)
            ms, reports = pp.average_gpu_time_ms()
            rec.update(gpu_ms_per_frame=round(ms, 5) if reports else None, reports=reports)
        finally:
            pp.close()
        pp = A.PostProcessor(device=shard.device_index, **dict(shard.cfg_kw, radius=radius_v))
        try:
            def frame():
                pp.apply(A.EYE_LEFT, L, out=oL)
                pp.apply(A.EYE_RIGHT, R, out=oR)

            def wall_ms(fn, it=200):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(it):
                    fn()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / it * 1e3
            frame(); frame()
            rec["direct_ms_per_frame"] = round(wall_ms(frame), 5)
            try:
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        frame()
                torch.cuda.synchronize(dev)
                rec["graph_ms_per_frame"] = round(wall_ms(g.replay), 5)
            except Exception as e:  # noqa: BLE001 -- graph capture is optional evidence, never fatal for the bench line
                rec["graph_ms_per_frame"] = None
                rec["graph_error"] = str(e)[:120]
        finally:
            pp.close()
        # cfg.pair_submit (ABI 4): the same two applies per frame, LEFT recorded and RIGHT launching both eyes as one batch of two
        pp = A.PostProcessor(device=shard.device_index, **dict(shard.cfg_kw, radius=radius_v, debug_mode=1, pair_submit=1))
        try:
            for _ in range(frames):
                pp.apply(A.EYE_LEFT, L, out=oL)
                pp.apply(A.EYE_RIGHT, R, out=oR)
            torch.cuda.synchronize(dev)
            ms, reports = pp.average_gpu_time_ms()
            rec["pair_submit_gpu_ms_per_frame"] = round(ms, 5) if reports else None
        finally:
            pp.close()
        return rec

    out = {"pairs_per_call": 1, "applies_per_frame": 2, "frames": frames,
           "method": "gpu_ms_per_frame = ovrfsr_average_gpu_time_ms (debug_mode=1: the reference's ring of 6 timestamp pairs, mean of 500, x2 per-eye "
                     "textures, PostProcessor.cpp:605-626); graph/direct = wall time per frame of the two applies replayed from a HIP graph / issued from Python; "
                     "pair_submit_gpu_ms_per_frame = the same figure from a cfg.pair_submit ctx (the LEFT apply is recorded, the RIGHT apply launches both eyes as one "
                     "batch of two: two launches per frame instead of four, same pixels)"}
    first = one(radius)
    out.update({k: v for k, v in first.items() if k != "radius"})
    out["radius"] = radius
    if radius != 0.5 and not (inW, inH) == (outW, outH):
        out["shipped_radius_0.5"] = one(0.5)   # openvr_mod.cfg's default: what a user of the reference runs
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--pairs", type=int, default=64,
                    help="stereo pairs per GPU per step (64 pairs of C2 = 7 GB resident; a step is then ~6 ms, so the round driver's 20 steps time > 0.1 s)")
    ap.add_argument("--workload", default="C2", choices=sorted(k for k in WORKLOADS if k != MOCK_WORKLOAD))
    ap.add_argument("--precision", default="fp32", choices=["fp32", "strict"])
    ap.add_argument("--fused", type=int, default=-1)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: read HBM-traffic and VALU counters in this run (rocprofv3 --pmc child passes, ~1 min)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing only: map shard i to device i %% device_count (exercise the N>1 launchers on a box with fewer GPUs)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed call's outputs (parity_check)")
    ap.add_argument("--no-extras", action="store_true", help="skip the un-timed extra legs (content_random, content_natural, frame)")
    ap.add_argument("--content", default="structured", choices=["structured", "random"],
                    help="synthetic eye content: structured (gradients+edges+noise, default) or uniform random")
    return ap.parse_args(argv)


def plan(args, env):
    """(world, rank, local device indices, first shard index): torchrun gives one device per rank; a direct
    invocation drives all --gpus devices from this process."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    if world > 1:
        return world, rank, [int(env.get("LOCAL_RANK", "0"))], rank
    return 1, 0, list(range(max(1, args.gpus))), 0


def shard_factory():
    """GpuShard, or -- for the launcher tests only -- the factory named by OVRFSR_BENCH_SHARD_FACTORY ("module:callable",
    called as f(device_index, shard_index, args)): lets tests/test_bench_launcher.py run EXACTLY the commands the round driver
    runs (`bench.py --gpus N`, and the same under torch.distributed.run) on a box without GPUs.  A mocked run carries
    "mock_shards": true in its line and no roofline / parity / cpu legs."""
    spec = os.environ.get("OVRFSR_BENCH_SHARD_FACTORY")
    if not spec:
        return None
    import importlib
    mod, fn = spec.split(":")
    return getattr(importlib.import_module(mod), fn)


def gather_over_ranks(values, world):
    """Per-device timings of every rank, in rank order (host-side gather of a few floats over gloo)."""
    if world == 1:
        return list(values)
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, list(values))
    return [v for part in out for v in part]


def main(argv=None):
    args = parse_args(argv)
    mock = shard_factory()
    assert mock or torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    if args.pmc_child:
        return pmc_child(args)
    world, rank, devices, first_shard = plan(args, os.environ)
    n_visible = max(devices) + 1 if mock else torch.cuda.device_count()
    if args.oversubscribe:
        devices = [d % n_visible for d in devices]
    assert max(devices) < n_visible, "--gpus %d but only %d device(s) visible" % (args.gpus, n_visible)
    cross = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # host-side barrier + max-join only; the data path has no collective.  gloo announces its connections on STDOUT
        # ("[Gloo] Rank 0 is connected to ..."): keep stdout for the one JSON line by pointing fd 1 at stderr while it connects
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        cross = dist.barrier
    if not mock:
        torch.cuda.set_device(devices[0])
    make = mock or GpuShard
    shards = build_shards(len(devices), first_shard, lambda i, s: make(devices[i], s, args))
    n_gpus = world * len(devices)

    dt_local, dev_ms = run_local(shards, args.steps, args.warmup, 0.0 if mock else CLOCK_RAMP_S, cross)
    dt = max_over_ranks(dt_local, world)
    dev_ms_local0 = dev_ms[0] if dev_ms else None   # shard 0 of THIS process: HIP events around exactly the timed steps
    dev_ms = gather_over_ranks(dev_ms, world)   # one entry per GPU of the job, whatever the launcher
    # every shard checks image 0 of its own timed batch against the oracle (collective: every rank contributes its records)
    per_shard = None if args.no_verify else gather_over_ranks([shard_parity(args, s) for s in shards], world)

    inW, inH, outW, outH, dtype, radius, use_nis = WORKLOADS[args.workload]
    value = args.pairs * n_gpus * args.steps / dt
    roof = cpu = par = None
    bad = False
    if rank == 0:
        shards[0].bind()
        extras = {"mock_shards": True} if mock else {}
        # the outputs of the TIMED launches, before anything else writes the output batch
        indices = check_indices(images_per_pair(args.workload) * args.pairs)
        fetched = None if (args.no_verify or mock) else shards[0].fetch(indices)
        roof = None if mock else roofline(args, shards[0], timed_ms_step=(dev_ms_local0 / args.steps) if dev_ms_local0 else None)
        if n_gpus == 1 and not args.no_extras and not mock:
            def leg(fn, *a):   # un-timed extra legs run before the line is printed: a failure in one must not lose the headline
                try:
                    return fn(*a)
                except Exception as e:  # noqa: BLE001
                    return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if args.content == "structured":
                extras["content_random"] = leg(content_random_leg, args, shards[0], args.steps // 2)
                extras["content_natural"] = leg(content_natural_leg, args, shards[0], args.steps // 2)
            extras["frame"] = leg(frame_leg, args, shards[0])
        if fetched is not None:
            par, cpu = parity_and_cpu(args, fetched, indices, want_cpu=(n_gpus == 1 and not args.no_cpu))   # CPU baseline: N=1 only
        if per_shard is not None:
            par = par or {"images": 0, "ok": True, "call": "per-shard records only (mocked shards)"}
            par["per_shard"] = per_shard
            par["failed_shards"] = [r["shard"] for r in per_shard if not r["ok"]]
            par["ok"] = bool(par["ok"] and not par["failed_shards"])
        if par is not None:
            bad = not par["ok"]
        shape = ("%dx%d->%dx%d" % (inW, inH, outW, outH)) + (" side-by-side textures holding both eyes" if args.workload in SHARED else "")
        line = {
            "metric": "stereo eye-pairs/sec at 1683x1869->2244x2492 (EASU+RCAS)" if args.workload == "C2"
                      else "stereo eye-pairs/sec (%s)" % args.workload,
            "value": round(value, 2), "unit": "eye-pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (%s)" % args.content,
            "config": {"workload": "%s: stereo pairs %s %s, %s, sharpness 0.9, radius %.1f"
                                   % (args.workload, shape, "RGBA8" if dtype == torch.uint8 else "RGBA16F",
                                      ("NIS NVSharpen" if (inW, inH) == (outW, outH) else "NIS NVScaler") if use_nis else
                                      ("RCAS" if (inW, inH) == (outW, outH) else "EASU+RCAS (UNORM8 intermediate)" if dtype == torch.uint8
                                       else "EASU+RCAS (half intermediate)"), radius),
                       "pairs_per_gpu_per_step": args.pairs, "precision": args.precision, "clock_ramp_s": CLOCK_RAMP_S,
                       "launcher": "torchrun, one rank per GPU, gloo timing barrier" if world > 1
                                   else "one process, %d device(s), one host thread + stream per device" % len(devices),
                       "per_device_ms_per_step": [round(m / args.steps, 4) for m in dev_ms],
                       **({"oversubscribed": "TEST RUN: %d shards on %d device(s), not a scaling measurement" % (n_gpus, n_visible)}
                          if args.oversubscribe else {}),
                       "parallelism": "batch sharded over %d GPU(s), no collective" % n_gpus},
            "roofline": roof, "cpu_baseline": cpu, "parity_check": par, **extras,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    for s in shards:
        s.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if bad:   # the line above says so too ("parity_check": {"ok": false}); a number for wrong pixels is not a result
        who = ", ".join("shard %d (device %d)" % (r["shard"], r["device"]) for r in (par.get("per_shard") or []) if not r["ok"])
        sys.exit("bench.py: the timed call's outputs differ from the oracle beyond the stated tolerance" + (": " + who if who else ""))


if __name__ == "__main__":
    main()
