#!/bin/bash
# tools/build_pp.sh NAME [EXTRA] -- A/B build with the ISA peephole (tools/isa_peephole.py) applied to both kernel TUs -> ab/NAME.so
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/openvr_fsr_amd/csrc"
NAME=$1; EXTRA=$2; B=build_$NAME; L=/opt/rocm/lib/llvm/bin
F="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=on $EXTRA"
mkdir -p $B ../../ab
for TU in fsr_kernels nis_kernels; do
 (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -fno-slp-vectorize --cuda-device-only -S $TU.hip -o $B/$TU.s 2>/dev/null
  python3 "$ROOT/tools/variants/isa_peephole.py" $B/$TU.s $B/$TU.pp.s
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/$TU.pp.s -o $B/$TU.dev.o
  $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/$TU.hsaco $B/$TU.dev.o
  $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/$TU.hsaco -output=$B/$TU.hipfb
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -fno-slp-vectorize --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/$TU.hipfb -c $TU.hip -o $B/$TU.o
 ) &
done
for C in postprocessor constants nis_config config_json capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip $F -ffp-contract=off -c $C.cpp -o $B/$C.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/$NAME.so $B/fsr_kernels.o $B/nis_kernels.o $B/postprocessor.o $B/constants.o $B/nis_config.o $B/config_json.o $B/capi.o
echo "built ab/$NAME.so"
