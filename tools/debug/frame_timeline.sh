#!/bin/bash
# tools/debug/frame_timeline.sh [lib.so] -- per-launch timeline of the one-pair-per-frame loop (tools/debug/frame_loop.py) from rocprofv3's kernel trace:
# mean duration of every kernel of a frame, the idle gaps between consecutive kernels, and the frame period.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$1" ] && export OVRFSR_LIB=$PWD/$1
for R in 2.0 0.5; do
  rm -rf /tmp/ft_run
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ft_run -o ft -- python tools/debug/frame_loop.py $R 300 > /tmp/ft_run.log 2>&1
  F=$(find /tmp/ft_run -name '*kernel_trace.csv' | head -1)
  python - "$F" $R <<'PY'
import csv, sys, re, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ovrfsr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]            # steady state
name = lambda r: re.search(r"ovrfsr_\w+::(\w+)", r["Kernel_Name"]).group(1)
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[name(a)].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[name(a) + " -> " + name(b)].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
print("== radius %s: %d launches in steady state" % (sys.argv[2], len(rows)))
for k, v in dur.items(): print("   kernel %-26s n=%4d mean %7.2f us  min %7.2f  max %7.2f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
for k, v in gap.items(): print("   gap    %-50s n=%4d mean %7.2f us  min %7.2f" % (k, len(v), sum(v) / len(v), min(v)))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
per_eye = len(rows) / (2.0 if sys.argv[2] == "2.0" else 3.0)
print("   busy span per frame (2 eyes): %.2f us" % (span / per_eye * 2))
PY
done
