#!/usr/bin/env python3
"""tools/debug/kernel_res.py LIB.so [regex] -- VGPR / SGPR / spill / LDS of the gfx950 kernels in a library build (the metadata tests/test_kernel_resources.py reads)"""
import os, re, struct, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
lib, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
data = open(fat, "rb").read()
n = 0
for m in re.finditer(re.escape(MAGIC), data):
    p = m.start()
    (nb,) = struct.unpack_from("<Q", data, p + 24)
    o = p + 32
    for _ in range(nb):
        off, size, tl = struct.unpack_from("<QQQ", data, o); o += 24
        triple = data[o:o + tl].decode(); o += tl
        if "gfx950" not in triple or size == 0:
            continue
        elf = os.path.join(tmp, "co%d.elf" % n); n += 1
        open(elf, "wb").write(data[p + off:p + off + size])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            if not re.search(pat, dem):
                continue
            g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1))
            print("vgpr %3d sgpr %3d spill %d/%d scratch %d lds %6d  %s" % (g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                                                                         g("private_segment_fixed_size"), g("group_segment_fixed_size"), dem[:110]))
