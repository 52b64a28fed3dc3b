#!/bin/bash
# tools/debug/serial_vs_overlap.sh <workload> -- masked pass: stand-alone kernel durations (OVRFSR_SERIAL=1) vs the
# overlapped default, via rocprofv3 kernel-trace stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-C3r}
for S in 1 0; do
  rm -rf /tmp/svo_$S
  OVRFSR_SERIAL=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/svo_$S -o kt -- python bench.py --no-cpu --no-extras --no-verify --pmc off --workload $W --steps 10 --warmup 2 --pairs 8 > /tmp/svo_$S.log 2>&1
  echo "== OVRFSR_SERIAL=$S"; grep '^{"metric"' /tmp/svo_$S.log | tail -1 | python -c "
import sys, json
s = sys.stdin.read().strip()
if s:
    d = json.loads(s); print('pairs/s', d['value'], 'ms/step', d['ms_per_step'])
else:
    print('(no bench line: see /tmp/svo_$S.log)')"
  find /tmp/svo_$S -name '*kernel_stats.csv' -exec grep ovrfsr {} \; | cut -c1-160
done
