#!/bin/bash
# tools/debug/kt.sh <lib.so> <bench args...> -- rocprofv3 kernel-trace durations of the ovrfsr kernels for one library build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$1; shift
rm -rf /tmp/kt_run
OVRFSR_LIB=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o kt -- python bench.py --no-cpu --no-extras --no-verify --pmc off --steps 10 --warmup 2 "$@" > /tmp/kt_run.log 2>&1
echo "== $L $* (OVRFSR_SERIAL=${OVRFSR_SERIAL:-0})"
find /tmp/kt_run -name '*kernel_stats.csv' -exec grep ovrfsr {} \; | cut -c1-150
