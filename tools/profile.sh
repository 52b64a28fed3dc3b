#!/bin/bash
# tools/profile.sh -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# for FETCH_SIZE / WRITE_SIZE on bench.py, summaries written (small, CSV/text) under gpurun_out/profile/.
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile/$TAG
rm -rf "$OUT" /tmp/prof_$TAG; mkdir -p "$OUT" /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o kt -- python bench.py --no-cpu --no-extras --no-verify --pmc off "$@" > "$OUT/bench_under_kernel_trace.log" 2>&1
find /tmp/prof_$TAG/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$TAG/$C -o pmc -- python bench.py --no-cpu --no-extras --no-verify --pmc off "$@" --steps 2 --warmup 1 --pairs 4 > "$OUT/bench_under_$C.log" 2>&1
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
# HBM bytes per eye image of each ovrfsr kernel: WRITE_SIZE (KiB, calibrated 1.0 on 4-B/lane stores) +
# 2 x FETCH_SIZE (KiB; gfx950 rocprofv3 reports half the bytes of a coalesced read -- MI355X_MICROARCH.md, confirmed
# here: RCAS reads exactly its input).  The PMC passes ran with --pairs 4 -> 8 eye images per launch.
python - "$OUT" <<'PY'
import json, re, sys, os
out = sys.argv[1]
def parse(fn):
    d = {}
    for line in open(os.path.join(out, fn)):
        m = re.match(r"void (ovrfsr_\w+::\w+)<.*?mean=\s*([\d.]+)", line)
        if m: d[m.group(1)] = d.get(m.group(1), 0.0) + float(m.group(2)) * 1024.0  # template instances summed
    return d
f, w = parse("pmc_FETCH_SIZE.txt"), parse("pmc_WRITE_SIZE.txt")
res = {k: {"fetch_bytes_per_eye": 2 * f[k] / 8, "write_bytes_per_eye": w.get(k, 0) / 8,
           "hbm_bytes_per_eye": (2 * f[k] + w.get(k, 0)) / 8} for k in f}
json.dump(res, open(os.path.join(out, "traffic_per_eye.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
ls -la "$OUT"
cat "$OUT/kernel_stats.csv" | cut -c1-220 | head -12
cat "$OUT"/pmc_*.txt | cut -c1-200
tail -1 "$OUT/bench_under_kernel_trace.log" | cut -c1-600
