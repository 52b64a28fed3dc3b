#!/bin/bash
O=gpurun_out/r05_fork_threshold.txt; : > $O
for W in C2r C5 C3r; do for P in 1 2 4; do for S in 1 0; do
  OVRFSR_SERIAL=$S python bench.py --no-cpu --no-extras --no-verify --pmc off --steps 200 --warmup 20 --pairs $P --workload $W 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$W images/call %d  %s  %.4f ms/step  %.1f pairs/s' % (2*$P, 'serial' if $S else 'forked', d['ms_per_step'], d['value']))" | tee -a $O
done; done; done
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_gputests_1.txt 2>&1; tail -5 gpurun_out/r05_gputests_1.txt
