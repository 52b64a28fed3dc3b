"""Resource budgets of the compiled gfx950 kernels, read from the code objects inside libopenvr_fsr_amd.so (no GPU needed).

Round 3 measured that these numbers are part of the design, not an implementation detail (masked section;
profiles/r03_fused_threads.txt, profiles/r03_outside_raw_fence.txt):
  * C5 runs the fused kernel and the outside-tile kernel concurrently.  A SIMD has 512 VGPRs and 8 wave slots: three fused
    waves (LDS-limited) x 64 VGPRs leave room for five outside waves only if BOTH kernels stay at <= 64 allocated VGPRs.
    A fused kernel with an unchanged instruction stream but 73 VGPRs cost C5 21 %.
  * no product kernel may spill or use scratch (a spilled outside kernel ran 28 % slower stand-alone),
  * EASU must keep 5 workgroups per CU (LDS- and VGPR-wise), the fused kernel's static LDS must stay tiny (its dynamic LDS
    sits just under the 3-per-CU limit).
The metadata is the NT_AMDGPU_METADATA note of each code object (llvm-readelf --notes).
"""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "openvr_fsr_amd", "libopenvr_fsr_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{mangled kernel name: {field: int}} for every gfx950 kernel in the library."""
    objcopy, readelf = _tool("llvm-objcopy"), _tool("llvm-readelf")
    if not (objcopy and readelf):
        pytest.skip("llvm-objcopy / llvm-readelf not available")
    if not os.path.exists(LIB):
        pytest.fail("libopenvr_fsr_amd.so is not built: run __graft_entry__.build()")
    tmp = tmp_path_factory.mktemp("co")
    fat = str(tmp / "fat.bin")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    data = open(fat, "rb").read()
    out = {}
    n = 0
    for m in re.finditer(re.escape(MAGIC), data):  # one bundle per translation unit
        p = m.start()
        (nb,) = struct.unpack_from("<Q", data, p + 24)
        o = p + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" not in triple or size == 0:
                continue
            elf = str(tmp / ("co%d.elf" % n))
            n += 1
            open(elf, "wb").write(data[p + off:p + off + size])
            notes = subprocess.run([readelf, "--notes", elf], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out[name] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1)) for f in FIELDS}
    assert n >= 2 and out, "no gfx950 code objects found in the library"
    return out


def _alloc(v):
    return (v + 7) // 8 * 8  # VGPR allocation granule on gfx950 (512 per SIMD lane, unified with AGPRs)


def _pick(kernels, pattern):
    sel = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert sel, "no kernel matches %r" % pattern
    return sel


def test_library_is_gfx950_only_and_complete(kernels):
    names = "\n".join(kernels)
    for k in ("easu_fast_kernel", "rcas_dpp_kernel", "rcas_direct_kernel", "fused_kernel", "easu_outside_kernel",
              "outside_staged_kernel", "nis_scaler_kernel", "nis_sharpen_kernel"):
        assert ("ovrfsr_fast" in names) and (k in names), k
    # the strict validation build is a second instantiation of the same kernel text
    assert "ovrfsr_strict" in names


def test_no_product_kernel_spills_or_uses_scratch(kernels):
    bad = {k: v for k, v in _pick(kernels, r"ovrfsr_fast").items()
           if v["vgpr_spill_count"] or v["sgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad


def test_no_kernel_uses_agprs(kernels):
    # no MFMA on this path: an AGPR would only be a spill target
    assert not {k: v["agpr_count"] for k, v in kernels.items() if v["agpr_count"]}


def test_c5_pair_fits_one_simd(kernels):
    """3 fused waves + 5 outside waves per SIMD: both kernels at <= 64 allocated VGPRs (RGBA16F in / mid / out)."""
    fused = _pick(kernels, r"ovrfsr_fast12fused_kernelILi1ELi1ELi1ELi(32|40)ELi256E")
    outside = _pick(kernels, r"ovrfsr_fast19easu_outside_kernelILi1ELi1ELi1E")
    for k, v in {**fused, **outside}.items():
        assert _alloc(v["vgpr_count"]) <= 64, (k, v["vgpr_count"])
    worst_f = max(_alloc(v["vgpr_count"]) for v in fused.values())
    worst_o = max(_alloc(v["vgpr_count"]) for v in outside.values())
    assert 3 * worst_f + 5 * worst_o <= 512
    # LDS: three workgroups per CU.  Dynamic LDS at C5 (2370 -> 3160: the 34-row block's footprint is 29 texel rows at pitch 32):
    # colour + analysis + luma planes (36 B per cell) + kLumPadRows luma rows + the 34x34 float4 intermediate; static LDS of the shipped
    # kernel = the per-wave list counters and the footprint-maximum word of the HDR half guard (32 B; the 1.1 KB row / column tables belong to
    # the item-walking measurement variant, tools/variants/fsr_variants.patch, not to this library)
    dynamic_c5 = 32 * 29 * 36 + 32 * 5 * 4 + 34 * 34 * 16
    for k, v in fused.items():
        assert v["group_segment_fixed_size"] <= 64, (k, v["group_segment_fixed_size"])
        assert 3 * (dynamic_c5 + v["group_segment_fixed_size"]) <= 160 * 1024, (k, v["group_segment_fixed_size"])


def test_every_fused_instance_leaves_room_for_outside_waves(kernels):
    # all product instances of the fused kernel (any format triple): never above 72 -> at least 4 outside waves beside 3 fused
    for k, v in _pick(kernels, r"ovrfsr_fast12fused_kernel").items():
        assert _alloc(v["vgpr_count"]) <= 72, (k, v["vgpr_count"])


def test_easu_keeps_five_workgroups_per_cu(kernels):
    """easu_fast_kernel: LDS allows 5 workgroups of 4 waves per CU (pitch 28: 28 cells x 28 rows x 36 B + pad + static
    lists < 32 KiB); the registers must allow the same 5 waves per SIMD."""
    for k, v in _pick(kernels, r"ovrfsr_fast16easu_fast_kernelILi0ELi0ELi28ELb[01]E").items():
        assert _alloc(v["vgpr_count"]) * 5 <= 512, (k, v["vgpr_count"])
        assert v["group_segment_fixed_size"] <= 2560, (k, v["group_segment_fixed_size"])


def test_memory_bound_kernels_keep_eight_waves(kernels):
    """RCAS (DPP form) is the closest kernel to the HBM roof (52 %): 8 waves per SIMD."""
    for k, v in _pick(kernels, r"ovrfsr_fast15rcas_dpp_kernel").items():
        assert _alloc(v["vgpr_count"]) <= 64, (k, v["vgpr_count"])


def test_workgroup_sizes(kernels):
    for k, v in _pick(kernels, r"ovrfsr_(fast|strict)").items():
        assert v["max_flat_workgroup_size"] in (192, 256, 512, 1024), (k, v["max_flat_workgroup_size"])


def test_isa_waits_tool_flags_a_wait_between_loads(tmp_path):
    """tools/isa_waits.py on a synthetic listing: a vmcnt wait between two loads of a block is reported, one after them is not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_waits", os.path.join(ROOT, "tools", "isa_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lst = tmp_path / "k.s"
    lst.write_text("_Z4badkPf:\n\tglobal_load_dwordx4 v[0:3], v8, s[0:1]\n\tglobal_load_dwordx4 v[4:7], v9, s[0:1]\n"
                   "\ts_waitcnt vmcnt(1)\n\tv_add_f32 v0, v0, v1\n\tglobal_load_dwordx4 v[10:13], v9, s[0:1]\n\ts_endpgm\n"
                   "_Z5goodkPf:\n\tglobal_load_dword v0, v8, s[0:1]\n\tglobal_load_dword v1, v9, s[0:1]\n"
                   ".LBB1_2:\n\ts_waitcnt vmcnt(0)\n\tv_add_f32 v0, v0, v1\n\ts_endpgm\n")
    hits = mod.scan(str(lst))
    assert [(k, n, before, total) for k, _, n, before, total in hits] == [("_Z4badkPf", "1", 2, 3)]
