#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS" -- build an A/B variant of the library into ab/NAME.so (own object dir, so
# variants do not invalidate each other's objects).  Run the variants interleaved on one GPU box with tools/ab.sh.
set -e
cd "$(dirname "$0")/../openvr_fsr_amd/csrc"
mkdir -p ../../ab
make -j8 EXTRA="$2" BUILD="build_$1" OUT="../../ab/$1.so" >/dev/null
echo "built ab/$1.so ($2)"
