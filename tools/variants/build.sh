#!/bin/bash
# tools/variants/build.sh NAME "EXTRA FLAGS" [--syntax-only] -- rebuild a measurement variant of rounds 3-4 that no longer lives in
# the product sources: copies openvr_fsr_amd/csrc to a scratch directory, applies tools/variants/*.patch there (the patches put the
# #ifdef'd variant code back; without EXTRA flags the patched tree compiles to the shipped code objects), and builds ab/NAME.so with
# the given -D flags.  Run variants interleaved on one GPU box with tools/abn.sh.  The patches apply to the sources of commit
# "round 5: measurement variants moved out of the kernel files"; `--syntax-only` only type-checks the two kernel TUs.
#   tools/variants/build.sh scalar   "-DOVRFSR_EASU_SCALAR"        profiles/r04_easu_scalar_ab.txt
#   tools/variants/build.sh px1      "-DOVRFSR_EASU_1PX"           profiles/r03_easu_2px_ab.txt
#   tools/variants/build.sh hacc     "-DOVRFSR_HALF_ACC"           profiles/r03_half_acc.txt
#   tools/variants/build.sh items    "-DOVRFSR_FUSED_ITEMS=1"      profiles/r04_fused_variants.txt   (=2 [-DOVRFSR_FUSED_SKIP_VRING]: prefetching form / strip bound)
#   tools/variants/build.sh narrow   "-DOVRFSR_FUSED_NARROW=1"     profiles/r04_fused_variants.txt (5)
#   tools/variants/build.sh compact  "-DOVRFSR_NIS_COMPACT"        profiles/r04_nis_compaction.txt   ([-DOVRFSR_NIS_DENSE_LANES=N])
#   PATCHES=easu_fs_bundle tools/variants/build.sh fsb "-DOVRFSR_EASU_FS_BUNDLE [-DOVRFSR_EASU_OCC5]" | mme "-DOVRFSR_EASU_MM_EARLY"   profiles/r05_sched_ab.txt
#   PATCHES=rcas_px2 tools/variants/build.sh px2 "-DOVRFSR_RCAS_PX2 [-DOVRFSR_RCAS_PX2_WAVES=N]"  (run with OVRFSR_RCAS_PX2=32|16)   profiles/r05_sched_ab.txt (7)
#   tools/variants/build.sh nishalf  "-DOVRFSR_NIS_HALF_LDS"       profiles/r03_nis_variants.txt
#   PATCHES=fused_prefetch tools/variants/build.sh pf6 "-DOVRFSR_FUSED_PF_WAVES=6"  (run with OVRFSR_FUSED_PF=1 [OVRFSR_FUSED_TPW=N])       profiles/r06_fused_prefetch.txt
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
NAME=$1; EXTRA=$2; MODE=${3:-}
S=$(mktemp -d /tmp/ovrfsr_variant_XXXXXX)
trap 'rm -rf "$S"' EXIT
mkdir -p "$S/include" "$S/openvr_fsr_amd/csrc"
cp "$ROOT"/include/*.h "$S/include/"
cp "$ROOT"/openvr_fsr_amd/csrc/*.{inc,hip,h,hpp,cpp} "$ROOT"/openvr_fsr_amd/csrc/Makefile "$S/openvr_fsr_amd/csrc/"
for P in ${PATCHES:-fsr_variants nis_variants}; do (cd "$S" && patch -s -p1 < "$ROOT/tools/variants/$P.patch"); done
cd "$S/openvr_fsr_amd/csrc"
if [ "$MODE" == "--syntax-only" ]; then
  for TU in fsr_kernels.hip nis_kernels.hip; do
    ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize $EXTRA -fsyntax-only $TU
  done
  echo "syntax ok: $NAME ($EXTRA)"
else
  mkdir -p "$ROOT/ab"
  make -j8 EXTRA="$EXTRA" BUILD="build_$NAME" OUT="$ROOT/ab/$NAME.so" >/dev/null
  echo "built ab/$NAME.so ($EXTRA)"
fi
