#!/bin/bash
# tools/ubench/run_rcas_dpp_ab.sh -- run the RCAS direct-loads vs DPP-neighbours A/B on the GPU box (via gpurun), then once more
# under rocprofv3 --pmc for the instruction counts; everything lands in gpurun_out/rcas_dpp_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/ubench/rcas_dpp_ab | tee gpurun_out/rcas_dpp_ab.txt
rm -rf /tmp/dppab
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/dppab -o pmc -- tools/ubench/rcas_dpp_ab > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/rcas_dpp_ab.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("/tmp/dppab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
px = (2244 - 128) * (2492 - 8) * 32
for k, cs in sorted(agg.items()):
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print("%-42s %-18s mean per launch %14.0f  = %.1f per 64 px" % (k, c, m, m / (px / 64.0)))
PY
