#!/bin/bash
# tools/build_asan.sh -- the SANITIZER build of SURVEY.md section 5 ("-fsanitize=address on host tests"): the library's host translation
# units (postprocessor.cpp, capi.cpp, constants.cpp, nis_config.cpp, config_json.cpp and the host halves of the kernel files) compiled
# with AddressSanitizer + UndefinedBehaviorSanitizer into ab/asan.so, and the two plain-C drivers of the ABI (examples/headless.c,
# examples/bench_node.c) likewise into ab/headless_asan / ab/bench_node_asan.  Device code is not instrumented (GPU ASan needs xnack+
# code objects, which this pool refuses); the device side has its own checked build (fsr_bounds.h, ab/bounds.so).
#   python processes load ab/asan.so with   LD_PRELOAD=$(tools/build_asan.sh --runtime) OVRFSR_LIB=$PWD/ab/asan.so ASAN_OPTIONS=detect_leaks=0
#   (tests/test_sanitizers.py does; leaks are not checked: the interpreter and the HIP runtime hold allocations until exit)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CLANG=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin/clang
RT=$(ls ${ROCM_PATH:-/opt/rocm}/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
if [ "$1" = "--runtime" ]; then echo "$RT"; exit 0; fi
if [ "$1" = "--runtime-gcc" ]; then echo "$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"; exit 0; fi
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
mkdir -p "$ROOT/ab"
make -C "$ROOT/openvr_fsr_amd/csrc" -j8 EXTRA="$SAN --offload-compress" LDEXTRA="-fsanitize=address,undefined" BUILD=build_asan OUT=../../ab/asan.so 2>&1 | grep -v "option-ignored\|^/opt/rocm\|^make" || true
test -f "$ROOT/ab/asan.so"
ROCM=${ROCM_PATH:-/opt/rocm}
for ex in headless bench_node thread_stress; do
    SRC="$ROOT/examples/$ex.c"; [ -f "$SRC" ] || SRC="$ROOT/tests/debug/$ex.c"   # (thread_stress.c lives with the tests)
    # (-shared-libasan: the drivers use the runtime as a shared object -- 30 KB each instead of 3.2 MB of statically linked runtime to push to the GPU box)
    $CLANG -std=c11 -O1 -g $SAN -shared-libasan -pthread -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ "$SRC" -I"$ROOT/include" -I"$ROCM/include" \
        "$ROOT/ab/asan.so" -L"$ROCM/lib" -lamdhip64 -lm -Wl,-rpath,"\$ORIGIN" -Wl,-rpath,"$ROCM/lib" -Wl,-rpath,"$(dirname "$RT")" -o "$ROOT/ab/${ex}_asan"
done
# The same host translation units under GCC's AddressSanitizer + UBSan, linked with the PRODUCT's kernel objects -> ab/asan_gcc.so: the one
# a Python process that also holds torch can load on the GPU box (LD_PRELOAD of gcc's libasan / libubsan).  ROCm's clang ASan runtime
# intercepts hsa_amd_memory_pool_allocate for GPU-ASan and cannot be preloaded into a process whose HIP runtime is torch's private,
# uninstrumented copy ("AddressSanitizer: out of memory" at the first device allocation) -- the plain-C drivers above, which link
# /opt/rocm's runtime directly, are fine with it.
# (the product's kernel translation units once more with a zstd-compressed fat binary: 0.8 instead of 3.4 MB to push)
make -C "$ROOT/openvr_fsr_amd/csrc" -j8 EXTRA=--offload-compress BUILD=build_z build_z/fsr_kernels.o build_z/nis_kernels.o >/dev/null
mkdir -p "$ROOT/openvr_fsr_amd/csrc/build_asan_gcc"
for f in postprocessor constants nis_config config_json capi; do
    g++ -std=c++17 -O1 -g1 -fPIC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ -I"$ROCM/include" -ffp-contract=off $SAN \
        -c "$ROOT/openvr_fsr_amd/csrc/$f.cpp" -o "$ROOT/openvr_fsr_amd/csrc/build_asan_gcc/$f.o"
done
${HIPCC:-$ROCM/bin/hipcc} --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/asan_gcc.so" "$ROOT"/openvr_fsr_amd/csrc/build_asan_gcc/*.o \
    "$ROOT/openvr_fsr_amd/csrc/build_z/fsr_kernels.o" "$ROOT/openvr_fsr_amd/csrc/build_z/nis_kernels.o"
python3 "$ROOT/tools/variant_fresh.py" --stamp asan "$SAN"
echo "built ab/asan.so ab/asan_gcc.so ab/headless_asan ab/bench_node_asan ab/thread_stress_asan"
