#!/bin/bash
# tools/pmc_valu.sh -- per-workload VALU / LDS / vector-memory counters of the ovrfsr kernels (run on the GPU box via gpurun):
# two rocprofv3 --pmc passes per workload (counters only), summarised per kernel into gpurun_out/profile/<tag>/sq_<workload>.txt
# Usage: tools/pmc_valu.sh <tag> [workloads...]
set -u
TAG=${1:-sq}; shift || true
WL=${@:-C2 C2r C3 C3r C4 C5 C2s C3s}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile/$TAG
mkdir -p "$OUT"
for W in $WL; do
  : > "$OUT/sq_$W.txt"
  i=0
  for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rm -rf /tmp/sqv_${W}_$i
    rocprofv3 --pmc $SET --output-format csv -d /tmp/sqv_${W}_$i -o pmc -- python bench.py --no-cpu --no-extras --no-verify --pmc off --steps 2 --warmup 1 --pairs 4 --workload $W ${PMC_EXTRA_ARGS:-} > /dev/null 2>&1
    F=$(find /tmp/sqv_${W}_$i -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python - "$F" "$W" >> "$OUT/sq_$W.txt" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "ovrfsr" not in k: continue
    agg[re.sub(r"\(.*", "", k.replace("void ", ""))[:64]][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for k, cs in agg.items():
    for c, v in sorted(cs.items()):
        print("%-8s %-66s %-22s n=%3d mean per launch (8 eye images) %16.1f" % (sys.argv[2], k, c, len(v), sum(v) / len(v)))
    if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
        a, g = sum(cs["SQ_ACTIVE_INST_VALU"]) / len(cs["SQ_ACTIVE_INST_VALU"]), sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        print("%-8s %-66s %-22s %.3f  (4*SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))" % (sys.argv[2], k, "valu_active_ratio", 4 * a / (g / 8 * 1024)))
PY
  done
  cat "$OUT/sq_$W.txt"
done
