#!/bin/bash
# tools/debug/rcas_occ_ab.sh -- RCAS at reduced occupancy: an experiment build (PATCHES=rcas_lds_cap tools/variants/build.sh rlds "") passes OVRFSR_RCAS_LDS bytes of (unused) dynamic LDS to rcas_dpp_kernel's
# batch launch, which caps the workgroups a CU holds at floor(160 KB / bytes) (profiles/r05_sched_ab.txt section 8)
export OVRFSR_LIB=$PWD/ab/rlds.so
run() { python bench.py --no-cpu --no-extras --pmc off --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pairs/s', d['value'], 'dominant_ms', r['launch_ms'], 'parity', (d.get('parity_check') or {}).get('ok'))"; }
for i in 1 2; do for b in 0 23000 27000 40000 53000 65000; do echo -n "C2s lds=$b  "; OVRFSR_RCAS_LDS=$b run --workload C2s; done; done
