#!/bin/bash
# tools/build_tsan.sh -- the RACE-DETECTION build of SURVEY.md section 5 (the reference has none: one render thread, one D3D11 immediate
# context): the library's host translation units under clang's ThreadSanitizer -> ab/tsan.so, and tests/debug/thread_stress.c (T host threads,
# each creating / driving / destroying its own ctxs through every launch form at once) -> ab/thread_stress_tsan.  Device code is not
# instrumented.  Run:   TSAN_OPTIONS="suppressions=tools/tsan.supp" ab/thread_stress_tsan --threads 4 --rounds 1
# (tests/test_gpu_threads.py does; tools/tsan.supp drops the reports that lie inside the uninstrumented HIP / HSA runtimes).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ROCM=${ROCM_PATH:-/opt/rocm}
CLANG=$ROCM/lib/llvm/bin/clang
RT=$(ls $ROCM/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
if [ "$1" = "--runtime" ]; then echo "$RT"; exit 0; fi
SAN="-fsanitize=thread -fno-omit-frame-pointer -gline-tables-only"
mkdir -p "$ROOT/ab"
make -C "$ROOT/openvr_fsr_amd/csrc" -j8 EXTRA="$SAN --offload-compress" LDEXTRA="-fsanitize=thread" BUILD=build_tsan OUT=../../ab/tsan.so 2>&1 | grep -v "option-ignored\|^/opt/rocm\|^make" || true
test -f "$ROOT/ab/tsan.so"
$CLANG -std=c11 -O1 $SAN -shared-libsan -pthread -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ "$ROOT/tests/debug/thread_stress.c" -I"$ROOT/include" -I"$ROCM/include" \
    "$ROOT/ab/tsan.so" -L"$ROCM/lib" -lamdhip64 -lm -Wl,-rpath,"\$ORIGIN" -Wl,-rpath,"$ROCM/lib" -Wl,-rpath,"$(dirname "$RT")" -o "$ROOT/ab/thread_stress_tsan"
python3 "$ROOT/tools/variant_fresh.py" --stamp tsan "$SAN"
echo "built ab/tsan.so ab/thread_stress_tsan"
