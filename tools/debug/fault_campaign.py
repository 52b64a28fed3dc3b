#!/usr/bin/env python3
"""tools/debug/fault_campaign.py -- FAULT INJECTION into the launch manager of a checked build (OVRFSR_LIB=ab/bounds.so).

ovrfsr_debug_fail_resource(n) makes the n-th device allocation / stream / event creation of the library fail, once.  For every pipeline
configuration below and n = 1, 2, 3 ... until a call goes through unharmed, a fresh ctx gets one apply with the failure armed, and the claims of
include/openvr_fsr_amd.h about failures are checked (reference behaviour: PostProcessor.cpp:23-28, 145-152 -- a failed resource build disables
the processor and the game's texture goes to the compositor untouched; the reference has no way to exercise it):
  * the call returns a status (OUT_OF_MEMORY or HIP), never crashes, and ovrfsr_last_error() says something;
  * the caller's OUTPUT IMAGE IS UNTOUCHED -- every byte still holds the fill pattern (the stream is synchronised first);
  * a failure during the (re)build leaves the ctx DISABLED: the next apply returns OVRFSR_ERR_DISABLED without touching anything;
  * a failure the library can absorb (the auxiliary stream and its events: everything then runs in order on the caller's stream) yields a
    CORRECT result instead;
  * after ovrfsr_reset the same ctx, with nothing armed, produces exactly the pixels of a ctx that never saw a failure.

    OVRFSR_LIB=$PWD/ab/bounds.so python tools/debug/fault_campaign.py        (GPU)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import openvr_fsr_amd as A  # noqa: E402
from tests import synth  # noqa: E402

DEV = torch.device("cuda")
FILL = 0x5A


def arm(n):
    A.library().ovrfsr_debug_fail_resource(int(n))      # AttributeError: not a checked build


def tensor(img8, fmt):
    t = torch.from_numpy(img8).to(DEV)
    if fmt == "f16":
        return (t.float() / 255.0).to(torch.float16), None
    if fmt == "bgra8":
        return t[..., [2, 1, 0, 3]].contiguous(), A.FORMAT_BGRA8
    return t, None


def main():
    iw, ih, ow, oh = 150, 110, 200, 147
    img8 = synth.structured_u8(iw, ih, 3)
    configs = [("two-pass RGBA8, unmasked", "u8", torch.uint8, dict(radius=2.0)),
               ("mask-sorted RGBA8 (tile lists, tap tables)", "u8", torch.uint8, dict(radius=0.5)),
               ("two-pass RGBA8, masked, plain (fused=0)", "u8", torch.uint8, dict(radius=0.5, fused=0)),
               ("fused half, masked (auxiliary stream)", "f16", torch.float16, dict(radius=0.5)),
               ("fused on request", "u8", torch.uint8, dict(radius=2.0, fused=1)),
               ("EASU only", "u8", torch.uint8, dict(radius=2.0, stage_mask=1)),
               ("NVScaler, masked", "u8", torch.uint8, dict(radius=0.5, use_nis=1)),
               ("NVScaler half, masked (auxiliary stream)", "f16", torch.float16, dict(radius=0.5, use_nis=1)),
               ("BGRA8 submission (swizzle buffer)", "bgra8", torch.uint8, dict(radius=2.0)),
               ("debug mode (timestamp events)", "u8", torch.uint8, dict(radius=0.5, debug_mode=1)),
               ("pair_submit, masked", "u8", torch.uint8, dict(radius=0.5, pair_submit=1))]
    total_inj = total_abs = 0
    bad = []
    for name, fmt, odt, cfg in configs:
        tex, ifmt = tensor(img8, fmt)
        kw = dict(fsr_enabled=1, out_width=ow, out_height=oh, sharpness=0.9)
        kw.update(cfg)
        arm(0)
        clean = A.PostProcessor(**dict(kw, pair_submit=0))   # (the reference pixels come from a plain ctx: same pixels by contract)
        ref = clean.apply(0, tex, out_dtype=odt, in_format=ifmt).clone()
        torch.cuda.synchronize()
        clean.close()
        injected = absorbed = 0
        for n in range(1, 40):
            pp = A.PostProcessor(**kw)
            out = torch.full((oh, ow, 4), 0, dtype=odt, device=DEV)
            out.view(torch.uint8).fill_(FILL)
            tex2, out_r = tex.clone(), torch.empty_like(out)
            arm(n)
            try:
                got = pp.apply(0, tex, out=out, in_format=ifmt)
                if cfg.get("pair_submit"):   # the LEFT apply only records: the RIGHT one launches both (its own texture and output)
                    pp.apply(1, tex2, out=out_r, in_format=ifmt)
                torch.cuda.synchronize()
                status = 0
            except A.OvrFsrError as e:
                status = e.status
                msg = str(e)
            arm(0)
            torch.cuda.synchronize()
            if status == 0:
                # either the failure was absorbed (auxiliary stream: in-order fallback) or n exceeds the creations of this call: correct pixels both ways
                if not torch.equal(got.view(torch.uint8), ref.view(torch.uint8)):
                    bad.append((name, n, "apply returned OK with wrong pixels"))
                pp.close()
                # did anything fire?  a second, unarmed ctx-creation count is not observable: stop when two consecutive n go through
                absorbed += 1
                if absorbed >= 3:
                    break
                continue
            absorbed = 0
            injected += 1
            if status not in (3, 6):
                bad.append((name, n, "status %d" % status))
            if "status" not in msg or len(msg) < 20:
                bad.append((name, n, "no error text: %r" % msg))
            if not bool((out.view(torch.uint8) == FILL).all()):
                bad.append((name, n, "the output image was written to by a failed apply"))
            # the next apply: DISABLED (failed rebuild) or a working ctx (failure outside the rebuild) -- never wrong pixels
            out2 = torch.full((oh, ow, 4), 0, dtype=odt, device=DEV)
            out2.view(torch.uint8).fill_(FILL)
            try:
                g2 = pp.apply(0, tex, out=out2, in_format=ifmt)
                if cfg.get("pair_submit"):
                    pp.apply(1, tex2, out=out_r, in_format=ifmt)
                torch.cuda.synchronize()
                if not torch.equal(g2.view(torch.uint8), ref.view(torch.uint8)):
                    bad.append((name, n, "the apply after a failed one returned wrong pixels"))
            except A.OvrFsrError as e2:
                if e2.status != 5:
                    bad.append((name, n, "the apply after a failed one: status %d" % e2.status))
                if not bool((out2.view(torch.uint8) == FILL).all()):
                    bad.append((name, n, "a DISABLED ctx wrote to the output image"))
            pp.reset()
            g3 = pp.apply(0, tex, out=out2, in_format=ifmt)
            if cfg.get("pair_submit"):
                pp.apply(1, tex2, out=out_r, in_format=ifmt)
            torch.cuda.synchronize()
            if not torch.equal(g3.view(torch.uint8), ref.view(torch.uint8)):
                bad.append((name, n, "after reset: wrong pixels"))
            pp.close()
        total_inj += injected
        print("%-48s %2d failures injected, each: status + text, output untouched, DISABLED or working afterwards, reset -> correct pixels" % (name, injected), flush=True)
    print("TOTAL %d injected failures over %d configurations, %d violations" % (total_inj, len(configs), len(bad)))
    for b in bad[:30]:
        print("VIOLATION", b)
    sys.exit(1 if bad or total_inj < 20 else 0)


if __name__ == "__main__":
    main()
