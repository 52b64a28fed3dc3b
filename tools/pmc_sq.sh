#!/bin/bash
# tools/pmc_sq.sh -- SQ/LDS counter passes for the ovrfsr kernels (run on the GPU box via gpurun).
# Usage: tools/pmc_sq.sh <tag> [bench args...]
set -u
TAG=${1:-sq}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile/$TAG
mkdir -p "$OUT"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/sq_${TAG}_$i
  rocprofv3 --pmc $SET --output-format csv -d /tmp/sq_${TAG}_$i -o pmc -- python bench.py --no-cpu --no-extras --no-verify --pmc off --steps 2 --warmup 1 --pairs 4 "$@" > "$OUT/sq_pass$i.log" 2>&1
  F=$(find /tmp/sq_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python - "$F" >> "$OUT/sq_counters.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "ovrfsr" not in k: continue
    agg[k[:70]][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for k, cs in agg.items():
    for c, v in sorted(cs.items()):
        print("%-72s %-26s n=%3d mean=%16.1f" % (k, c, len(v), sum(v) / len(v)))
PY
done
cat "$OUT/sq_counters.txt"
