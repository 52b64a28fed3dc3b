#!/bin/bash
# tools/variants/regen.sh -- carry the variant patches over a change of the product kernel sources (3-way merge).
# tools/variants/BASE holds the commit the patches currently apply to.  For every patched file: base = that commit's file, theirs = base + patch,
# ours = the file in the working tree (commit it first); `git merge-file` merges the variant code into ours and the patch is re-cut as
# diff(ours, merged).  Conflicts (a product change inside variant code) stop the script: resolve by hand in the printed scratch directory.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"
BASE=$(cat tools/variants/BASE)
S=$(mktemp -d /tmp/ovrfsr_regen_XXXXXX)
for P in ${PATCH_LIST:-fsr_variants nis_variants easu_fs_bundle easu_soa rcas_pipe rcas_px2 rcas_lds_cap fused_prefetch}; do
  FILES=$(grep '^+++ ' tools/variants/$P.patch | sed 's#^+++ [ab]/##; s#\t.*##')
  mkdir -p $S/$P/base/openvr_fsr_amd/csrc $S/$P/theirs/openvr_fsr_amd/csrc $S/$P/a/openvr_fsr_amd/csrc $S/$P/b/openvr_fsr_amd/csrc
  for F in $FILES; do git show $BASE:$F > $S/$P/base/$F; cp $S/$P/base/$F $S/$P/theirs/$F; done
  (cd $S/$P/theirs && patch -s -p1 < "$ROOT/tools/variants/$P.patch")
  : > $S/$P.patch
  for F in $FILES; do
    cp $F $S/$P/a/$F; cp $F $S/$P/b/$F
    git merge-file -q $S/$P/b/$F $S/$P/base/$F $S/$P/theirs/$F || { echo "CONFLICT in $F for $P: resolve in $S/$P/b/$F"; exit 1; }
    (cd $S/$P && diff -u a/$F b/$F >> $S/$P.patch) || true
  done
  cp $S/$P.patch tools/variants/$P.patch
  echo "regenerated tools/variants/$P.patch"
done
git rev-parse HEAD > tools/variants/BASE
rm -rf $S
