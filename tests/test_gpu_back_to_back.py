"""Back-to-back launches of the multi-stream pipelines (GPU): a race detector for the ctx's stream graph.

A masked pipeline runs two or three kernels concurrently on ctx-owned auxiliary streams (touching tiles || outside tiles, then
RCAS on the joined result) and keeps its intermediate in ONE ctx-owned buffer; the unmasked two-kernel pipeline reuses that
buffer too.  Every call must therefore order itself behind the previous call's readers: a missing event edge shows up only when
calls are issued WITHOUT a host synchronisation in between, with different data -- exactly what bench.py's timed loop and a
game's render thread do.  Here: two different input batches A and B, 24 calls alternating A, B, A, B ... into 24 distinct output
buffers, one synchronisation OF THE CALLER'S STREAM at the end (the ABI's promise; a device-wide one would hide a missing join);
every A result must equal the first A result (taken with full synchronisation before the
loop), every B result the first B result, bit for bit.  Also through `ovrfsr_apply` (one image per call, alternating eyes).
Mutation check (rounds 4-5, GPU): a library built with -DOVRFSR_MUTATE_NO_JOIN (the aux stream's join edge dropped) fails these tests
(since round 5 only half-float sources fork; the RGBA8 cases run in order on the caller's stream and keep their place as
ordering tests of the shared intermediate)."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

FP32 = 0
CASES = [
    ("two-pass unmasked RGBA8", dict(radius=2.0, sharpness=0.9), np.uint8),
    ("sorted two-pass, radius 0.5 (3 streams)", dict(radius=0.5, sharpness=0.9, debug_mode=1), np.uint8),
    ("fused + outside, RGBA16F, radius 0.5 (2 streams)", dict(radius=0.5, sharpness=0.9), np.float16),
    ("fused kernel on request, radius 0.6", dict(radius=0.6, sharpness=0.7, fused=1), np.uint8),
    ("NVScaler + DirectCopy, radius 0.5", dict(radius=0.5, sharpness=0.9, use_nis=1), np.uint8),
    ("EASU only, radius 0.4", dict(radius=0.4, stage_mask=1), np.uint8),
    ("EASU only, radius 0.12: nearly every tile outside", dict(radius=0.12, stage_mask=1), np.uint8),
    ("NVScaler, radius 0.12: DirectCopy is the long pole", dict(radius=0.12, use_nis=1, sharpness=0.5), np.uint8),
    # round 5: RGBA8 pipelines run their outside-tile kernel in order on the caller's stream (faster at every batch size); only sources
    # whose outside tiles take the per-pixel kernel still fork onto the auxiliary stream -- half-float ones, as in the third case
    ("EASU only, RGBA16F, radius 0.3 (forked)", dict(radius=0.3, stage_mask=1), np.float16),
    ("NVScaler + DirectCopy, RGBA16F, radius 0.4 (forked)", dict(radius=0.4, use_nis=1, sharpness=0.6), np.float16),
]


def _batch(dt, seed, n, iw, ih):
    import torch
    imgs = []
    for i in range(n):
        g = synth.structured_u8 if (seed + i) % 2 else synth.random_u8
        a = g(iw, ih, 1000 * seed + i)
        if dt == np.float16:
            a = (a.astype(np.float32) / 255.0).astype(np.float16)
        imgs.append(a)
    return torch.from_numpy(np.stack(imgs)).cuda()


@pytest.mark.parametrize("name,cfg,dt", CASES, ids=[c[0] for c in CASES])
def test_back_to_back_batches_are_ordered(name, cfg, dt):
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, n, calls = 960, 810, 1280, 1080, 6, 16
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16}[dt]
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    try:
        ins = [_batch(dt, 1, n, iw, ih), _batch(dt, 2, n, iw, ih)]
        ref = []
        for t in ins:
            o = torch.zeros((n, oh, ow, 4), dtype=tdt, device="cuda")
            pp.apply_batch(t, o)
            torch.cuda.synchronize()
            ref.append(o.clone())
        assert not torch.equal(ref[0], ref[1])
        outs = [torch.zeros((n, oh, ow, 4), dtype=tdt, device="cuda") for _ in range(calls)]
        torch.cuda.synchronize()
        snaps = []
        for k in range(calls):            # no synchronisation between the calls
            pp.apply_batch(ins[k & 1], outs[k])
            # a consumer on the caller's stream right behind the call (the ABI's promise: the outputs are valid once the CALLER'S
            # stream gets here -- a kernel still running on an auxiliary stream would be caught mid-write by this copy)
            snaps.append(outs[k].clone())
        torch.cuda.current_stream().synchronize()
        for k in range(calls):
            assert torch.equal(snaps[k].view(torch.uint8), ref[k & 1].view(torch.uint8)), "%s: the consumer of call %d saw other data than its reference" % (name, k)
            assert torch.equal(outs[k].view(torch.uint8), ref[k & 1].view(torch.uint8)), "%s: call %d differs from its reference" % (name, k)
    finally:
        pp.close()


@pytest.mark.parametrize("name,cfg,dt", CASES[:3], ids=[c[0] for c in CASES[:3]])
def test_back_to_back_single_applies_are_ordered(name, cfg, dt):
    """the reference's calling pattern: Apply(left), Apply(right), next frame ... with no synchronisation"""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, frames = 480, 405, 640, 540, 12
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16}[dt]
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    try:
        src = _batch(dt, 3, 4, iw, ih)     # images 0,1 = frame type A (left, right); 2,3 = frame type B
        ref = []
        for i in range(4):
            o = pp.apply(i & 1, src[i], out_dtype=tdt)
            torch.cuda.synchronize()
            ref.append(o.clone())
        outs = []
        torch.cuda.synchronize()
        for f in range(frames):
            for eye in (0, 1):
                o = torch.zeros((oh, ow, 4), dtype=tdt, device="cuda")
                pp.apply(eye, src[2 * (f & 1) + eye], out=o)
                outs.append((2 * (f & 1) + eye, o.clone()))   # consumer on the caller's stream right behind the call
        torch.cuda.current_stream().synchronize()
        for k, (i, o) in enumerate(outs):
            assert torch.equal(o.view(torch.uint8), ref[i].view(torch.uint8)), "%s: apply %d differs from its reference" % (name, k)
    finally:
        pp.close()


def test_two_ctxs_on_two_streams_are_independent():
    """"distinct ctxs are independent" (include/openvr_fsr_amd.h): two ctxs with different configs, each on its own caller stream,
    calls interleaved from one host thread with no synchronisation; each ctx owns its intermediate, tile lists and auxiliary stream"""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, n, calls = 960, 810, 1280, 1080, 4, 10
    cfgs = [dict(radius=0.5, sharpness=0.9), dict(radius=2.0, sharpness=0.4)]
    pps = [A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **c) for c in cfgs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    try:
        ins = [_batch(np.uint8, 5, n, iw, ih), _batch(np.uint8, 6, n, iw, ih)]
        torch.cuda.synchronize()
        ref = {}
        for c in range(2):
            for b in range(2):
                o = torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda")
                with torch.cuda.stream(streams[c]):
                    pps[c].apply_batch(ins[b], o)
                torch.cuda.synchronize()
                ref[c, b] = o.clone()
        outs, snaps = [], []
        torch.cuda.synchronize()
        for k in range(calls):
            for c in range(2):
                o = torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda")
                torch.cuda.current_stream().synchronize()          # the zero fill ran on the default stream
                with torch.cuda.stream(streams[c]):
                    pps[c].apply_batch(ins[(k + c) & 1], o)
                    snaps.append(o.clone())
                outs.append((c, (k + c) & 1, o))
        for s in streams:
            s.synchronize()
        for i, (c, b, o) in enumerate(outs):
            assert torch.equal(o, ref[c, b]), "ctx %d call %d differs" % (c, i)
            assert torch.equal(snaps[i], ref[c, b]), "ctx %d call %d: consumer saw other data" % (c, i)
    finally:
        for p in pps:
            p.close()


def test_two_host_threads_with_their_own_ctxs():
    """no hidden global state: two host threads, each with its own ctx and caller stream on the SAME device, run different
    pipelines at the same time (ctypes releases the GIL during the calls); each thread's results equal its references"""
    import threading
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, n, calls = 640, 540, 853, 720, 4, 40
    cfgs = [dict(radius=0.5, sharpness=0.9), dict(radius=0.45, sharpness=0.6, use_nis=1)]
    ins = [_batch(np.uint8, 7, n, iw, ih), _batch(np.uint8, 8, n, iw, ih)]
    torch.cuda.synchronize()
    errors = []

    def work(c):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfgs[c])
                try:
                    ref = []
                    for b in range(2):
                        o = torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda")
                        pp.apply_batch(ins[b], o)
                        s.synchronize()
                        ref.append(o.clone())
                    outs = [torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda") for _ in range(calls)]
                    s.synchronize()
                    for k in range(calls):
                        pp.apply_batch(ins[k & 1], outs[k])
                    s.synchronize()
                    for k in range(calls):
                        if not torch.equal(outs[k], ref[k & 1]):
                            errors.append("thread %d call %d differs" % (c, k))
                finally:
                    pp.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (c, e))

    ts = [threading.Thread(target=work, args=(c,)) for c in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("cfg", [dict(radius=0.5, sharpness=0.9), dict(radius=2.0, sharpness=0.9), dict(radius=0.5, sharpness=0.9, use_nis=1)],
                         ids=["sorted two-pass", "unmasked two-pass", "NVScaler masked"])
def test_back_to_back_with_changing_batch_sizes(cfg):
    """the ctx-owned intermediate follows the largest batch seen: growing it between two unsynchronised calls must not pull it
    from under the kernels of the previous call"""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh = 640, 540, 853, 720
    sizes = [2, 6, 3, 8, 1, 8, 5, 12, 4, 12]
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    try:
        src = _batch(np.uint8, 9, max(sizes), iw, ih)
        ref = torch.zeros((max(sizes), oh, ow, 4), dtype=torch.uint8, device="cuda")
        for i in range(max(sizes)):   # one image per call, eye parity as in a batch, fully synchronised
            pp.apply_batch(src[i:i + 1], ref[i:i + 1], first_eye=i & 1)
            torch.cuda.synchronize()
        pp.reset()                    # the intermediate starts small again
        outs = [torch.zeros((n, oh, ow, 4), dtype=torch.uint8, device="cuda") for n in sizes]
        torch.cuda.synchronize()
        snaps = []
        for n, o in zip(sizes, outs):
            pp.apply_batch(src[:n], o)
            snaps.append(o.clone())
        torch.cuda.current_stream().synchronize()
        for k, (n, o) in enumerate(zip(sizes, outs)):
            assert torch.equal(o, ref[:n]), "call %d (batch of %d) differs" % (k, n)
            assert torch.equal(snaps[k], ref[:n]), "call %d (batch of %d): consumer saw other data" % (k, n)
    finally:
        pp.close()


LIFE_CASES = [
    ("sorted two-pass RGBA8", dict(radius=0.5, sharpness=0.9), np.uint8),
    ("fused + outside RGBA16F (forked)", dict(radius=0.5, sharpness=0.9), np.float16),
    ("NVScaler masked", dict(radius=0.45, sharpness=0.8, use_nis=1), np.uint8),
    ("unmasked two-pass, debug mode", dict(radius=2.0, sharpness=0.9, debug_mode=1), np.uint8),
]


@pytest.mark.parametrize("name,cfg,dt", LIFE_CASES, ids=[c[0] for c in LIFE_CASES])
def test_reset_set_config_and_destroy_with_work_in_flight(name, cfg, dt):
    """Reset(), a hotkey's set_config and the destructor release tile lists, tap tables, the intermediate, events and the auxiliary stream
    (PostProcessor.cpp:166-194, 659-709) -- here with the previous call's kernels possibly still running, as a render thread does it: no host
    synchronisation anywhere between the calls.  Every output equals the one the same configuration gave with full synchronisation; the
    last ctx is destroyed with work in flight and its outputs are read afterwards."""
    import copy
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, n = 640, 540, 853, 720, 4
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16}[dt]
    src = _batch(dt, 11, n, iw, ih)
    alt = dict(cfg, sharpness=0.3, radius=0.7 if cfg["radius"] < 2 else 2.0)      # what a hotkey changes
    ref = {}
    for key, c in (("a", cfg), ("b", alt)):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **c)
        ref[key] = torch.zeros((n, oh, ow, 4), dtype=tdt, device="cuda")
        pp.apply_batch(src, ref[key])
        torch.cuda.synchronize()
        pp.close()
    assert not torch.equal(ref["a"], ref["b"])
    pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    cfg_a, cfg_b = copy.copy(pp.cfg), copy.copy(pp.cfg)
    cfg_b.sharpness, cfg_b.radius = alt["sharpness"], alt["radius"]
    script = ["a", "reset", "a", "b", "b", "reset", "b", "a", "a", "reset", "a", "b"]
    outs, want, cur = [], [], "a"
    torch.cuda.synchronize()
    for step in script:
        if step == "reset":
            pp.reset()
            continue
        if step != cur:
            pp.set_config(cfg_a if step == "a" else cfg_b)
            cur = step
        o = torch.zeros((n, oh, ow, 4), dtype=tdt, device="cuda")
        pp.apply_batch(src, o)
        outs.append(o); want.append(step)
    pp.close()                                    # destructor with the last calls in flight
    torch.cuda.current_stream().synchronize()
    for k, (o, w) in enumerate(zip(outs, want)):
        assert torch.equal(o, ref[w]), "call %d (configuration %s) differs from the synchronised run" % (k, w)


# ------------------------------------------------------------------------------------------------
# round 5: cfg.pair_submit -- apply(LEFT) is recorded, apply(RIGHT) launches both eyes as one batch of two
# ------------------------------------------------------------------------------------------------
PAIR_CASES = [
    ("two-pass unmasked RGBA8", dict(radius=2.0, sharpness=0.9), np.uint8),
    ("sorted two-pass, radius 0.5, different mask centres per eye", dict(radius=0.5, sharpness=0.9, proj_centre=(0.45, 0.5, 0.56, 0.48)), np.uint8),
    ("fused + outside, RGBA16F, radius 0.5 (forked)", dict(radius=0.5, sharpness=0.9), np.float16),
    ("NVScaler + DirectCopy, radius 0.5", dict(radius=0.5, sharpness=0.9, use_nis=1), np.uint8),
    ("EASU only", dict(radius=2.0, stage_mask=1), np.uint8),
]


@pytest.mark.parametrize("name,cfg,dt", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_pair_submit_is_the_same_pixels_in_two_launches(name, cfg, dt):
    """Frames submitted L, R, L, R ... without synchronisation through a pair_submit ctx equal, bit for bit, what a plain ctx writes -- with the
    right eye's texture above OR below the left one's in memory (image 0 of the batch of two is the lower one), into caller-owned outputs."""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, frames = 480, 405, 640, 540, 6
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16}[dt]
    src = _batch(dt, 5, 4, iw, ih)      # frames of type A = images (0, 1), type B = (2, 3)
    plain = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    ref = []
    for i in range(4):
        ref.append(plain.apply(i & 1, src[i], out_dtype=tdt).clone())
    torch.cuda.synchronize()
    plain.close()
    for order in ("left first", "right first"):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, pair_submit=1, **cfg)
        try:
            res = []
            for f in range(frames):
                a = 2 * (f & 1)
                outs = torch.zeros((2, oh, ow, 4), dtype=tdt, device="cuda")
                # "right first": the right eye's images sit BELOW the left eye's in memory (input and output alike)
                li, ri = (a, a + 1)
                lo, ro = (0, 1) if order == "left first" else (1, 0)
                pair_in = torch.stack([src[li], src[ri]] if order == "left first" else [src[ri], src[li]])
                il, ir = (pair_in[0], pair_in[1]) if order == "left first" else (pair_in[1], pair_in[0])
                o_l = pp.apply(A.EYE_LEFT, il, out=outs[lo])
                assert o_l.data_ptr() == outs[lo].data_ptr()
                pp.apply(A.EYE_RIGHT, ir, out=outs[ro])
                res.append((a, outs[lo].clone(), outs[ro].clone(), pair_in))   # consumers on the caller's stream right behind the RIGHT call
            torch.cuda.current_stream().synchronize()
            for k, (a, gl, gr, _) in enumerate(res):
                assert torch.equal(gl.view(torch.uint8), ref[a].view(torch.uint8)), "%s, %s: left eye of frame %d differs" % (name, order, k)
                assert torch.equal(gr.view(torch.uint8), ref[a + 1].view(torch.uint8)), "%s, %s: right eye of frame %d differs" % (name, order, k)
        finally:
            pp.close()


@pytest.mark.parametrize("name,cfg,dt", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_pair_submit_pairs_by_arrival_order(name, cfg, dt):
    """Round 6: a game that submits R, L, R, L ...  The first eye of a frame is recorded whichever it is, the other eye
    launches both: after the second call of frame f BOTH results of frame f are there (in stream order) -- until round 6 only LEFT was
    recorded, so LEFT(f) was batched with RIGHT(f + 1) and its output stayed unwritten for a whole frame.  Caller-owned outputs laid out either
    way round relative to the inputs (the strides of the batch of two are differences modulo 2^64), and ctx-owned ones."""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh, frames = 480, 405, 640, 540, 5
    tdt = {np.uint8: torch.uint8, np.float16: torch.float16}[dt]
    src = _batch(dt, 11, 4, iw, ih)
    plain = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, **cfg)
    ref = [plain.apply(i & 1, src[i], out_dtype=tdt).clone() for i in range(4)]
    torch.cuda.synchronize()
    plain.close()
    for layout in ("outputs like inputs", "outputs reversed", "ctx-owned"):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, pair_submit=1, **cfg)
        try:
            res = []
            for f in range(frames):
                a = 2 * (f & 1)
                outs = torch.zeros((2, oh, ow, 4), dtype=tdt, device="cuda")
                lo, ro = (0, 1) if layout == "outputs like inputs" else (1, 0)
                if layout == "ctx-owned":
                    o_r = pp.apply(A.EYE_RIGHT, src[a + 1])          # recorded: nothing launched yet
                    o_l = pp.apply(A.EYE_LEFT, src[a])               # both eyes, one batch of two
                    assert o_r.data_ptr() != o_l.data_ptr()
                    res.append((a, o_l.clone(), o_r.clone()))
                else:
                    o_r = pp.apply(A.EYE_RIGHT, src[a + 1], out=outs[ro])
                    assert o_r.data_ptr() == outs[ro].data_ptr()
                    pp.apply(A.EYE_LEFT, src[a], out=outs[lo])
                    res.append((a, outs[lo].clone(), outs[ro].clone()))  # a consumer right behind the second call of the SAME frame
            torch.cuda.current_stream().synchronize()
            for k, (a, gl, gr) in enumerate(res):
                assert torch.equal(gl.view(torch.uint8), ref[a].view(torch.uint8)), "%s, %s: left eye of frame %d differs" % (name, layout, k)
                assert torch.equal(gr.view(torch.uint8), ref[a + 1].view(torch.uint8)), "%s, %s: right eye of frame %d differs" % (name, layout, k)
        finally:
            pp.close()


def test_pair_submit_ctx_owned_image_survives_a_size_change():
    """a recorded eye with a ctx-owned output, then a submission of another size: the recorded eye is processed with the old
    resources and the rebuild must not free the image its result went to (the caller was handed that pointer)."""
    import torch
    import openvr_fsr_amd as A
    iw, ih = 320, 270
    src = _batch(np.uint8, 13, 2, iw, ih)
    small = _batch(np.uint8, 14, 1, 160, 120)
    kw = dict(fsr_enabled=1, render_scale=0.75, precision=FP32, radius=0.6, sharpness=0.8)
    plain = A.PostProcessor(**kw)
    ref = plain.apply(0, src[0], out_dtype=torch.uint8).clone()
    ref_small = plain.apply(1, small[0], out_dtype=torch.uint8).clone()
    torch.cuda.synchronize()
    plain.close()
    pp = A.PostProcessor(pair_submit=1, **kw)
    try:
        o_l = pp.apply(0, src[0])                    # recorded, ctx-owned destination
        o_r = pp.apply(1, small[0])                  # another size: LEFT flushed with the old resources, ctx rebuilt, this eye recorded
        junk = [torch.full((427 * 360 * 4,), 77, dtype=torch.uint8, device="cuda") for _ in range(8)]   # would reuse a freed block
        o_l2 = pp.apply(0, small[0])                 # pairs with the recorded small RIGHT
        torch.cuda.synchronize()
        assert torch.equal(o_l, ref), "the left image was freed or overwritten by the rebuild"
        assert torch.equal(o_r, ref_small) and o_l2.shape == o_r.shape
        del junk
    finally:
        pp.close()


def test_pair_submit_falls_back_to_single_launches():
    """A LEFT that no matching RIGHT follows is processed on its own by the next call: two LEFTs in a row, outputs laid out the other way round
    than the inputs, a texture of another size, ovrfsr_apply_batch in between, ctx-owned outputs (two images in this mode); reset drops it."""
    import torch
    import openvr_fsr_amd as A
    iw, ih, ow, oh = 320, 270, 427, 360
    src = _batch(np.uint8, 7, 4, iw, ih)
    kw = dict(fsr_enabled=1, out_width=ow, out_height=oh, precision=FP32, radius=0.6, sharpness=0.8)
    plain = A.PostProcessor(**kw)
    ref = [plain.apply(i & 1, src[i], out_dtype=torch.uint8).clone() for i in range(4)]
    torch.cuda.synchronize()
    plain.close()
    pp = A.PostProcessor(pair_submit=1, **kw)
    try:
        outs = torch.zeros((4, oh, ow, 4), dtype=torch.uint8, device="cuda")
        # LEFT, LEFT, RIGHT: the first LEFT is flushed by the second, which is processed at once too (the same eye twice stops the recording
        # until the other eye is seen: a one-eye-per-frame host must not see every result a call late); the RIGHT is then processed at once
        pp.apply(0, src[0], out=outs[0]); assert pp.pair_pending()
        pp.apply(0, src[2], out=outs[2]); assert not pp.pair_pending()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[2], ref[2])
        pp.apply(0, src[0], out=outs[1]); assert not pp.pair_pending()      # still one eye only: immediate
        pp.apply(1, src[3], out=outs[3]); assert not pp.pair_pending()      # the other eye is back: immediate, pairing resumes with the next call
        torch.cuda.synchronize()
        assert torch.equal(outs[1], ref[0]) and torch.equal(outs[3], ref[3])
        pp.apply(0, src[0], out=outs[0]); assert pp.pair_pending()
        pp.apply(1, src[1], out=outs[1]); assert not pp.pair_pending()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])
        # outputs ordered against the inputs: still one batch of two since round 6 (strides modulo 2^64), same pixels
        outs.zero_()
        pp.apply(0, src[0], out=outs[1]); pp.apply(1, src[1], out=outs[0])
        torch.cuda.synchronize()
        assert torch.equal(outs[1], ref[0]) and torch.equal(outs[0], ref[1])
        # a batch call between LEFT and RIGHT: the LEFT goes first; the RIGHT -- a frame's SECOND eye (order learned from the pairs above) that
        # finds nothing recorded -- is a single apply at once, and the next frame pairs again (no standing lag)
        outs.zero_()
        b = torch.zeros((2, oh, ow, 4), dtype=torch.uint8, device="cuda")
        pp.apply(0, src[0], out=outs[0]); pp.apply_batch(src[2:4], b); assert not pp.pair_pending()
        pp.apply(1, src[1], out=outs[1]); assert not pp.pair_pending()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]) and torch.equal(b[0], ref[2]) and torch.equal(b[1], ref[3])
        pp.apply(0, src[2], out=outs[2]); assert pp.pair_pending()
        pp.apply(1, src[3], out=outs[3]); assert not pp.pair_pending()
        torch.cuda.synchronize()
        assert torch.equal(outs[2], ref[2]) and torch.equal(outs[3], ref[3])
        # ctx-owned outputs: two distinct images, both valid after the RIGHT call
        o_l = pp.apply(0, src[2]); o_r = pp.apply(1, src[3])
        torch.cuda.synchronize()
        assert o_l.data_ptr() != o_r.data_ptr() and torch.equal(o_l, ref[2]) and torch.equal(o_r, ref[3])
        # a RIGHT of another size: the recorded LEFT is processed with the old resources, then the ctx rebuilds
        small = _batch(np.uint8, 9, 1, 160, 120)
        outs.zero_()
        pp.apply(0, src[0], out=outs[0])
        pp2cfg = A.Config.default(pair_submit=1, **dict(kw, out_width=0, out_height=0, render_scale=0.75))
        pp.set_config(pp2cfg)          # set_config / reset DROP a recorded LEFT (documented)
        torch.cuda.synchronize()
        assert int(outs[0].max()) == 0
    finally:
        pp.close()


def test_hours_of_rebuilds_do_not_leak():
    """A VR session lasts hours and every hotkey, every resolution change and every recovered failure rebuilds the ctx's device resources (tile
    lists, tap tables, coefficient banks, the intermediate, events, the auxiliary stream).  2 400 rebuilds through every path that triggers one
    -- set_config, reset, an input-size change, pair_submit's retired image, ctx create / destroy -- and then: the device has as much free memory
    as before (hipMemGetInfo) and the process has not grown (RSS), within the allocators' granularity."""
    import copy
    import gc
    import psutil
    import torch
    import openvr_fsr_amd as A
    ow, oh = 427, 360
    srcs = {s: _batch(np.uint8, 21, 2, *s) for s in ((320, 270), (256, 200), (300, 240))}
    srcs16 = _batch(np.float16, 22, 2, 320, 270)
    out = torch.zeros((2, oh, ow, 4), dtype=torch.uint8, device="cuda")
    out16 = torch.zeros((2, oh, ow, 4), dtype=torch.float16, device="cuda")

    def cycle(n):
        pp = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, radius=0.5, sharpness=0.9, debug_mode=1)
        ph = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, radius=0.5, sharpness=0.9)                    # half: fused + forked outside kernel
        pq = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, radius=0.6, sharpness=0.9, pair_submit=1)     # ctx-owned outputs, retired images
        cfgs = []
        for r, nis in ((0.5, 0), (0.7, 0), (2.0, 0), (0.45, 1)):
            c = copy.copy(pp.cfg); c.radius = r; c.use_nis = nis; cfgs.append(c)
        sizes = list(srcs)
        for i in range(n):
            pp.set_config(cfgs[i % 4])                       # hotkey: rebuild on the next apply
            pp.apply_batch(srcs[sizes[i % 3]], out)          # ... and an input-size change two times out of three
            if i % 5 == 0:
                pp.reset()
            ph.apply_batch(srcs16, out16)
            if i % 7 == 0:
                ph.reset()
            t = srcs[sizes[(i // 2) % 3]]
            pq.apply(0, t[0]); pq.apply(1, t[1])             # the size changes every other frame: the recorded eye's image is retired
            if i % 50 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for p in (pp, ph, pq):
            p.close()
        for _ in range(n // 8):                              # ctx life cycle
            p = A.PostProcessor(fsr_enabled=1, out_width=ow, out_height=oh, radius=0.5)
            p.apply_batch(srcs[(320, 270)], out)
            p.close()
        torch.cuda.synchronize()

    cycle(40)                                                # warm-up: code objects, torch's caches, the runtime's pools
    gc.collect()
    free0, _ = torch.cuda.mem_get_info()
    rss0 = psutil.Process().memory_info().rss
    cycle(int(os.environ.get("OVRFSR_LEAK_CYCLES", "600")))   # 600 x (1 + ~0.67 + ...) rebuilds on pp, ph resets, 300 retired images, 75 ctxs
    gc.collect()
    free1, _ = torch.cuda.mem_get_info()
    rss1 = psutil.Process().memory_info().rss
    print("leak test: device free %+.2f MiB, host RSS %+.2f MiB over the cycle" % ((free1 - free0) / 2 ** 20, (rss1 - rss0) / 2 ** 20))
    assert free0 - free1 <= 16 << 20, "device memory shrank by %.1f MiB over the cycle" % ((free0 - free1) / 2 ** 20)
    assert rss1 - rss0 <= 48 << 20, "host RSS grew by %.1f MiB over the cycle" % ((rss1 - rss0) / 2 ** 20)
