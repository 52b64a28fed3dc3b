#!/bin/bash
# tools/debug/fused_pf_counters.sh -- stand-alone kernel durations (rocprofv3 --kernel-trace, outside-tile kernel serialised) and SQ counters of
# the fused kernel, one tile per workgroup against the persistent prefetching form (tools/variants/fused_prefetch.patch: ab/pf6.so = 80 VGPRs,
# ab/pf8.so = 64; build them as tools/debug/fused_pf_ab.sh says), workload C5, 16 eyes per launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for CFG in "shipped openvr_fsr_amd/libopenvr_fsr_amd.so 0" "prefetch80 ab/pf6.so 1" "prefetch64 ab/pf8.so 1"; do
  set -- $CFG
  echo "== $1 (OVRFSR_LIB=$2 OVRFSR_FUSED_PF=$3), C5, OVRFSR_SERIAL=1"
  rm -rf /tmp/kt_$1
  OVRFSR_SERIAL=1 OVRFSR_FUSED_PF=$3 OVRFSR_LIB=$PWD/$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$1 -o kt -- python bench.py --workload C5 --no-cpu --no-extras --no-verify --pmc off --steps 10 --warmup 2 --pairs 8 > /tmp/kt_$1.log 2>&1
  find /tmp/kt_$1 -name '*kernel_stats.csv' -exec grep ovrfsr {} \; | cut -c1-170
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
    i=$((i+1)); rm -rf /tmp/sq_$1_$i
    OVRFSR_SERIAL=1 OVRFSR_FUSED_PF=$3 OVRFSR_LIB=$PWD/$2 rocprofv3 --pmc $SET --output-format csv -d /tmp/sq_$1_$i -o pmc -- python bench.py --workload C5 --no-cpu --no-extras --no-verify --pmc off --steps 2 --warmup 1 --pairs 4 > /tmp/sq_$1_$i.log 2>&1
    F=$(find /tmp/sq_$1_$i -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "?")
    if "fused" not in k: continue
    agg[k[:60]][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for k, cs in agg.items():
    for c, v in sorted(cs.items()):
        print("   %-62s %-24s n=%3d mean=%16.1f" % (k, c, len(v), sum(v) / len(v)))
PY
  done
done
