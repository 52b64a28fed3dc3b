#!/bin/bash
# tools/profile.sh -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# for FETCH_SIZE / WRITE_SIZE on bench.py, summaries written (small, CSV/text) under gpurun_out/profile/.
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile/$TAG
rm -rf "$OUT" /tmp/prof_$TAG; mkdir -p "$OUT" /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o kt -- python bench.py --no-cpu "$@" > "$OUT/bench_under_kernel_trace.log" 2>&1
find /tmp/prof_$TAG/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$TAG/$C -o pmc -- python bench.py --no-cpu --steps 2 --warmup 1 --pairs 4 > "$OUT/bench_under_$C.log" 2>&1
  F=$(find /tmp/prof_$TAG/$C -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then
    python - "$F" "$C" > "$OUT/pmc_$C.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r.get("Kernel_Name", "?")].append(float(r.get("Counter_Value", 0)))
print("# %s per dispatch (rocprofv3 --pmc %s), raw counter units (KiB per rocprofv3's derived metric)" % (sys.argv[2], sys.argv[2]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-90s n=%4d mean=%14.1f min=%14.1f max=%14.1f" % (k[:90], len(v), sum(v) / len(v), min(v), max(v)))
PY
  fi
done
ls -la "$OUT"
cat "$OUT/kernel_stats.csv" | cut -c1-220 | head -12
cat "$OUT"/pmc_*.txt | cut -c1-200
tail -1 "$OUT/bench_under_kernel_trace.log" | cut -c1-600
