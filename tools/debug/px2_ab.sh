#!/bin/bash
# tools/debug/px2_ab.sh -- A/B of tools/variants/rcas_px2.patch (profiles/r05_sched_ab.txt section 7): PX2_LIB=ab/px2.so, built by
#   PATCHES=rcas_px2 tools/variants/build.sh px2 "-DOVRFSR_RCAS_PX2"; one library, the form chosen per process by OVRFSR_RCAS_PX2=0|16|32
export OVRFSR_LIB=$PWD/${PX2_LIB:-ab/px2.so}
run() { python bench.py --no-cpu --no-extras --pmc off --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pairs/s', d['value'], 'dominant_ms', r['launch_ms'], 'parity', (d.get('parity_check') or {}).get('ok'))"; }
for v in 0 16 32; do
OVRFSR_RCAS_PX2=$v python - <<'PY'
import os, torch, hashlib
import openvr_fsr_amd as A, bench
dev = torch.device("cuda")
hs = []
for (w,h) in ((2244,2492),(1000,777),(124,40),(125,33),(63,17),(250,64),(372,31)):
    x = bench.random_batch(2, w, h, torch.uint8, dev, 7)
    o = torch.zeros_like(x)
    pp = A.PostProcessor(fsr_enabled=1, out_width=w, out_height=h, sharpness=0.9, radius=2.0)
    pp.apply_batch(x, o, first_eye=A.EYE_LEFT, alternate_eyes=True); torch.cuda.synchronize(); pp.close()
    hs.append(hashlib.md5(o.cpu().numpy().tobytes()).hexdigest()[:8])
print("PX2=%s" % os.environ["OVRFSR_RCAS_PX2"], " ".join(hs))
PY
done
for i in 1 2 3; do for v in 0 16 32; do echo -n "C2s PX2=$v  "; OVRFSR_RCAS_PX2=$v run --workload C2s; done; done
for i in 1 2; do for v in 0 16 32; do echo -n "C2 PX2=$v  "; OVRFSR_RCAS_PX2=$v run --workload C2; done; done
