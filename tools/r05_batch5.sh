#!/bin/bash
O=gpurun_out/r05_fork_threshold.txt
for W in C2r C3r C5; do for P in 16 64; do for S in 1 0 1 0; do
  OVRFSR_SERIAL=$S python bench.py --no-cpu --no-extras --no-verify --pmc off --steps 40 --warmup 10 --pairs $P --workload $W 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$W images/call %d  %s  %.4f ms/step  %.1f pairs/s' % (2*$P, 'serial' if $S else 'forked', d['ms_per_step'], d['value']))" | tee -a $O
done; done; done
for S in 1 0; do OVRFSR_SERIAL=$S python tools/debug/frame_figure.py --workload C5 2>/dev/null | tail -1 | cut -c1-2000 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C5 frame figure serial=$S', {k:v for k,v in d.items() if k.endswith('per_frame')})" | tee -a $O; done
