#!/usr/bin/env python3
"""tools/collect_profile.py <tag> <workload> -- copy the small summaries tools/profile.sh left under
gpurun_out/profile/<tag>/ into profiles/<tag>/ and merge its per-kernel HBM traffic into profiles/traffic_per_eye.json
under <workload> (what bench.py reports as roofline.traffic)."""
import json, os, shutil, sys
tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "profile", tag), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in ("kernel_stats.csv", "pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "traffic_per_eye.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
if os.path.exists(os.path.join(src, "issue_roof.txt")):   # per-wave instruction counters + the VALU issue-cycle interval (tools/isa_costs.py)
    shutil.copy(os.path.join(src, "issue_roof.txt"), os.path.join(dst, "issue_roof.txt"))
# keep only the rows of our kernels + the header in the committed kernel stats
rows = open(os.path.join(dst, "kernel_stats.csv")).read().splitlines()
open(os.path.join(dst, "kernel_stats.csv"), "w").write("\n".join([rows[0]] + [r for r in rows[1:] if "ovrfsr" in r]) + "\n")
for f in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt"):
    rows = open(os.path.join(dst, f)).read().splitlines()
    open(os.path.join(dst, f), "w").write("\n".join(r for r in rows if r.startswith("#") or "ovrfsr" in r) + "\n")
log = open(os.path.join(src, "bench_under_kernel_trace.log")).read().splitlines()
line = [l for l in log if l.startswith("{")]
if line:
    open(os.path.join(dst, "bench_line_under_kernel_trace.json"), "w").write(line[-1] + "\n")
# HBM bytes per eye image per kernel (template instances of one kernel summed): WRITE_SIZE + 2 x FETCH_SIZE, both KiB per
# dispatch (gfx950: FETCH_SIZE reports half the bytes of a coalesced read, MI355X_MICROARCH.md); PMC passes ran --pairs 4
import re
def parse(fn):
    d = {}
    for l in open(os.path.join(dst, fn)):
        m = re.match(r"void (ovrfsr_\w+::\w+)<.*?mean=\s*([\d.]+)", l)
        if m: d[m.group(1)] = d.get(m.group(1), 0.0) + float(m.group(2)) * 1024.0
    return d
f, w = parse("pmc_FETCH_SIZE.txt"), parse("pmc_WRITE_SIZE.txt")
traffic = {k: {"fetch_bytes_per_eye": 2 * f[k] / 8, "write_bytes_per_eye": w.get(k, 0) / 8,
               "hbm_bytes_per_eye": (2 * f[k] + w.get(k, 0)) / 8} for k in f}
json.dump(traffic, open(os.path.join(dst, "traffic_per_eye.json"), "w"), indent=1)
path = os.path.join(root, "profiles", "traffic_per_eye.json")
allw = json.load(open(path))
allw[workload] = traffic
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(allw[workload], indent=1))
