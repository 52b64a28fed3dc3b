#!/bin/bash
# tools/ab.sh -- interleaved A/B of two builds of libopenvr_fsr_amd.so on ONE GPU box (same chip, same clocks).
# Usage (via gpurun): tools/ab.sh ab/libA.so ab/libB.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-4}; shift 3 || true
for i in $(seq 1 $R); do
  for L in $A $B; do
    OVRFSR_LIB=$PWD/$L python bench.py --no-cpu --no-extras --pmc off --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', 'pairs/s', d['value'], 'dominant_ms', r['launch_ms'], 'pipe_ms', r['pipeline_ms_per_step_events'], 'parity', (d.get('parity_check') or {}).get('ok'))"
  done
done
