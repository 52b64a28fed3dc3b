#!/bin/bash
# tools/profile.sh -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# for FETCH_SIZE / WRITE_SIZE on bench.py, summaries written (small, CSV/text) under gpurun_out/profile/.
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile/$TAG
rm -rf "$OUT" /tmp/prof_$TAG; mkdir -p "$OUT" /tmp/prof_$TAG
OVRFSR_BENCH_TIMED_GAP_MS=3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o kt -- python bench.py --no-cpu --no-extras --no-verify --pmc off "$@" > "$OUT/bench_under_kernel_trace.log" 2>&1
find /tmp/prof_$TAG/kt -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# kernel_stats.csv averages EVERY dispatch of the process -- the clock-ramp launches (a cold chip, ~50 ms of them), the warm-up and the roofline
# probes behind the timed steps included.  timed_kernel_stats.csv: the same trace restricted to the dispatches of the TIMED steps -- bench.py
# idles the device for 3 ms in front of and behind them under OVRFSR_BENCH_TIMED_GAP_MS (outside the timed region), and the segment between two
# such gaps whose length matches steps x ms_per_step is taken -- which is what ms_per_step of the same run can be compared with.
KT=$(find /tmp/prof_$TAG/kt -name '*kernel_trace.csv' | head -1)
[ -n "$KT" ] && python - "$KT" "$OUT/bench_under_kernel_trace.log" > "$OUT/timed_kernel_stats.csv" <<'PY'
import csv, json, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])) if "ovrfsr" in r.get("Kernel_Name", "")]
rows.sort()
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
steps, want_ns = int(line["steps"]), line["ms_per_step"] * 1e6 * int(line["steps"])
segs, cur, last_end = [], [], None
for s, e, k in rows:
    if last_end is not None and s - last_end > 1_000_000:   # > 1 ms of idle device: a segment boundary
        segs.append(cur); cur = []
    cur.append((s, e, k)); last_end = max(last_end or 0, e)
segs.append(cur)
best = min(segs, key=lambda g: abs((max(e for _, e, _ in g) - g[0][0]) - want_ns))
span = max(e for _, e, _ in best) - best[0][0]
by = collections.defaultdict(list)
for s, e, k in best:
    by[k].append(e - s)
print("Name,TimedDispatches,DispatchesPerStep,AverageNs,MinNs,MaxNs,NsPerStep")
total = 0.0
for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    total += sum(d) / steps
    print('"%s",%d,%.2f,%.1f,%d,%d,%.1f' % (k, len(d), len(d) / steps, sum(d) / len(d), min(d), max(d), sum(d) / steps))
print("# the dispatches of the %d timed steps of this run: first start to last end %.4f ms per step; sum of kernel durations %.4f ms per step "
      "(concurrent kernels count twice); the traced run's own ms_per_step %.4f" % (steps, span / steps / 1e6, total / 1e6, line["ms_per_step"]))
PY
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
# per-wave instruction counters -> VALU issue-cycle interval of every kernel (tools/isa_costs.py; what bench.py reports as roofline.valu.issue)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 --output-format csv -d /tmp/prof_$TAG/issue -o pmc -- python bench.py --no-cpu --no-extras --no-verify --pmc off "$@" --steps 2 --warmup 1 --pairs 4 > "$OUT/bench_under_issue_counters.log" 2>&1
F=$(find /tmp/prof_$TAG/issue -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && python - "$F" > "$OUT/issue_roof.txt" <<'PY'
import csv, sys, collections, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
import isa_costs
doc = isa_costs.load()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "ovrfsr" in r.get("Kernel_Name", ""):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per-wave instruction counters (rocprofv3 --pmc, launches of 8 eye images) and the VALU issue-cycle interval they imply (tools/isa_costs.py)")
print("# class costs (true cycles):", isa_costs.COST)
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    w = m.get("SQ_WAVES")
    cfg = isa_costs.find_kernel(doc, k)
    if not w or cfg is None:
        continue
    pw = {a: m[c] / w for a, c in (("valu", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR"),
                                   ("trans", "SQ_INSTS_VALU_TRANS_F32"), ("smem", "SQ_INSTS_SMEM"), ("salu", "SQ_INSTS_SALU")) if c in m}
    b = isa_costs.issue_bounds(cfg, pw)
    print("%s\n   waves %d  per wave: %s\n   issue cycles per wave: %s" % (k, w, " ".join("%s=%.3f" % kv for kv in pw.items()),
          "[%.1f, %.1f] (mean %.3f..%.3f cycles per VALU instruction; counters used: %s, tolerance %.2f)" % (b["lo"], b["hi"], b["mean_cost_lo"], b["mean_cost_hi"], ",".join(b["constraints"]), b["tolerance"]) if b else "no profile fits"))
PY
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
