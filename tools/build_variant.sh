#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS" -- build an A/B variant of the library into ab/NAME.so (own object dir, so
# variants do not invalidate each other's objects).  Run the variants interleaved on one GPU box with tools/ab.sh.
# A successful build leaves ab/NAME.so.stamp = hash of (flags, every source of csrc/, the public header): tools/variant_fresh.py NAME
# tells a test whether the library it is about to drive was built from the sources it sits next to -- the object directories do
# not travel to the GPU box (.gpurunignore), so "make says up to date" cannot be asked there.
set -e
cd "$(dirname "$0")/../openvr_fsr_amd/csrc"
mkdir -p ../../ab
# variants travel to the GPU box: their fat binaries are zstd-compressed (--offload-compress: 17 MB -> 4 MB for the checked build)
make -j8 EXTRA="$2 ${OVRFSR_VARIANT_COMPRESS---offload-compress}" BUILD="build_$1" OUT="../../ab/$1.so" >/dev/null
python3 ../../tools/variant_fresh.py --stamp "$1" "$2"
echo "built ab/$1.so ($2)"
