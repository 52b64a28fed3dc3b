#!/bin/bash
# tools/debug/fused_pf_ab.sh [rounds] -- round 6, VERDICT r5 next #6: the fused kernel as a persistent workgroup that prefetches the next tile's
# texels (tools/variants/fused_prefetch.patch), interleaved against the one-tile-per-workgroup form on ONE box: C5 (masked half pipeline: fused
# kernel on the inside tiles || outside-tile kernel) and C2 --fused 1 (the fused kernel alone over full frames).  Build the variants first:
#   for W in 8 6 5; do PATCHES=fused_prefetch tools/variants/build.sh pf$W "-DOVRFSR_FUSED_PF_WAVES=$W"; done     (64 / 80 / 96 VGPRs)
# (the record was taken while the kernel still sat in the product sources behind OVRFSR_FUSED_PF: same code)  -> profiles/r06_fused_prefetch.txt
R=${1:-3}
run() { # label lib env... -- workload args
  local label=$1 lib=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env OVRFSR_LIB=$PWD/$lib "${envs[@]}" python bench.py --no-cpu --no-extras --pmc off --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s' % '$label', '%-14s' % ' '.join('$*'.split()[1:3]), 'pairs/s %9.1f' % d['value'], ' step_ms %.4f' % d['ms_per_step'], ' parity', (d.get('parity_check') or {}).get('ok'))"
}
L=openvr_fsr_amd/libopenvr_fsr_amd.so
for i in $(seq 1 $R); do
  for W in "--workload C5" "--workload C2 --fused 1"; do
    run "one tile per workgroup (shipped)" $L OVRFSR_FUSED_PF=0 -- $W
    run "prefetch, 64 VGPR, 4 tiles/wg"  ab/pf8.so OVRFSR_FUSED_PF=1 -- $W
    run "prefetch, 64 VGPR, 2 tiles/wg"  ab/pf8.so OVRFSR_FUSED_PF=1 OVRFSR_FUSED_TPW=2 -- $W
    run "prefetch, 64 VGPR, 8 tiles/wg"  ab/pf8.so OVRFSR_FUSED_PF=1 OVRFSR_FUSED_TPW=8 -- $W
    run "prefetch, 80 VGPR, 4 tiles/wg"  ab/pf6.so OVRFSR_FUSED_PF=1 -- $W
    run "prefetch, 96 VGPR, 4 tiles/wg"  ab/pf5.so OVRFSR_FUSED_PF=1 -- $W
  done
done
