#!/bin/bash
# tools/bench_all.sh -- one short bench line per workload (run on the GPU box via gpurun); no CPU leg, no PMC passes.
# Usage: tools/bench_all.sh [out-file] [extra bench args...]
OUT=${1:-gpurun_out/bench_all.txt}; shift || true
: > "$OUT"
for W in C2 C2r C3 C3r C4 C5 C2s C3s C2sbs C2sbsr; do
  python bench.py --no-cpu --no-extras --pmc off --steps 30 --warmup 10 --workload $W "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
p=d.get('parity_check') or {}
print('%-6s %9.1f pairs/s  step %.4f ms  dominant %-58s %.4f ms  %6.1f GB/s (%.3f of HBM peak, %.3f of copy)  parity ok=%s max_lsb=%s max_abs=%s' % ('$W', d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['achieved'], r['frac'], r['frac_of_copy'], p.get('ok'), p.get('max_lsb'), p.get('max_abs')))" | tee -a "$OUT"
done
python bench.py --no-cpu --no-extras --pmc off --steps 30 --warmup 10 --workload C2 --fused 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 fused=1 %9.1f pairs/s  step %.4f ms' % (d['value'], d['ms_per_step']))" | tee -a "$OUT"
python bench.py --no-cpu --no-extras --pmc off --steps 30 --warmup 10 --workload C3 --content random 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 random  %9.1f pairs/s  step %.4f ms' % (d['value'], d['ms_per_step']))" | tee -a "$OUT"
