#!/bin/bash
# tools/abn.sh ROUNDS lib1.so lib2.so ... [-- bench args] -- interleaved A/B/.. of N builds of the library on ONE GPU box
# (same chip, same clocks; box-to-box spread is ~4 %).  Prints one line per (round, lib).
R=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq 1 $R); do
  for L in "${LIBS[@]}"; do
    OVRFSR_LIB=$PWD/$L python bench.py --no-cpu --no-extras --pmc off --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s' % '$L', 'pairs/s', d['value'], 'dominant_ms', r['launch_ms'], 'pipe_ms', r['pipeline_ms_per_step_events'], 'parity', (d.get('parity_check') or {}).get('ok'))"
  done
done
