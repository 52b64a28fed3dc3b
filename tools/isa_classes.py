#!/usr/bin/env python3
"""tools/isa_classes.py -- ISA-derived instruction-class table of one gfx950 kernel.

  hipcc --offload-arch=gfx950 <flags of csrc/Makefile> --cuda-device-only -S fsr_kernels.hip -o /tmp/fsr.s
  tools/isa_classes.py /tmp/fsr.s easu_fast_kernelILi0ELi0ELi28 [--weights LABEL=W,...] [--blocks]

Splits the kernel into basic blocks (labels / branches), classifies every instruction by its measured issue class
(profiles/r01_valu_issue_rates.txt) and prints count x cost per class.  Without --weights every block of an innermost
loop is weighted by --loop-trips (default 1) and straight-line code by 1: use --blocks once to see the blocks, then give
the trip count of each hot block (per thread) and 0 for cold ones (bilinear fallbacks, border paths).
Costs are cycles per wave-instruction per SIMD at nominal 2.4 GHz as measured by tools/ubench/valu_rates.
"""
import argparse
import collections
import re
import sys

# measured issue classes (cycles / wave-instruction / SIMD at 2.4 GHz nominal)
COST = {"valu_fast": 2.6, "valu_slow": 4.2, "valu_trans": 8.2, "valu_pk": 4.3, "lds_b128": 16.5, "lds_b64": 8.8, "lds_b32": 8.7,
        "lds_write": 8.8, "vmem": 0.0, "salu": 0.0, "other": 0.0}
FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32",
        "v_mul_legacy_f32", "v_not_b32", "v_accvgpr", "v_addc_co_u32", "v_fmamk_f32", "v_fmaak_f32", "v_madmk_f32", "v_madak_f32",
        "v_add_i32", "v_sub_i32", "v_sub_co_u32", "v_subb_co_u32")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        if op.endswith(("_dpp", "_sdwa")):
            return "valu_slow"   # measured: a DPP-modified op issues in the slow class whatever the base op
        base = re.sub(r"_(e32|e64)$", "", op)
        return "valu_fast" if base.startswith(FAST) else "valu_slow"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_b128" if "b128" in op else "lds_b64" if "b64" in op or "2addr" in op else "lds_b32"
    if op.startswith("ds_"):
        return "lds_write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_text(path, name):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[start:end]


def blocks_of(text):
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    loops = set()
    for l in text[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            if "Loop Header" in m.group(2) or "in Loop" in m.group(2):
                loops.add(cur)
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s.split()[0])
    return blocks, loops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--weights", default="", help="LABEL=W,... (labels as printed by --blocks, without the .LBB prefix); unlisted blocks weigh 0")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--px-per-thread", type=float, default=4.0)
    a = ap.parse_args()
    blocks, loops = blocks_of(kernel_text(a.asm, a.kernel))
    if a.blocks:
        phase = 0
        for b, ops in blocks.items():
            c = collections.Counter(classify(o) for o in ops)
            valu = sum(v for k, v in c.items() if k.startswith("valu"))
            lds = sum(v for k, v in c.items() if k.startswith("lds"))
            print("%-14s %s phase%d  n=%4d valu=%4d lds=%3d vmem=%3d  %s" % (b.replace(".LBB", ""), "LOOP" if b in loops else "    ", phase, len(ops), valu, lds,
                                                                       c.get("vmem", 0), "BARRIER" if "s_barrier" in ops else ""))
            if "s_barrier" in ops:
                phase += 1
        return
    w = {}
    for item in filter(None, a.weights.split(",")):
        k, v = item.split("=")
        w[".LBB" + k if not k.startswith("entry") else k] = float(v)
    tot = collections.Counter()
    ops_by_class = collections.defaultdict(collections.Counter)
    for b, ops in blocks.items():
        wt = w.get(b, 0.0)
        if wt == 0.0:
            continue
        for o in ops:
            c = classify(o)
            tot[c] += wt
            ops_by_class[c][o] += wt
    px = a.px_per_thread
    print("# %s -- weighted per thread, shown per output pixel (/%g) = per 64 px of one wave-instruction stream" % (a.kernel, px))
    print("%-12s %10s %8s %12s" % ("class", "instr/px", "cost", "cycles/px"))
    cyc = 0.0
    for c in sorted(tot, key=lambda k: -tot[k] * COST[k]):
        print("%-12s %10.1f %8.1f %12.1f   %s" % (c, tot[c] / px, COST[c], tot[c] / px * COST[c],
                                                  " ".join("%s:%.1f" % (o, n / px) for o, n in ops_by_class[c].most_common(8))))
        if c.startswith("valu"):
            cyc += tot[c] / px * COST[c]
    nv = sum(v for k, v in tot.items() if k.startswith("valu")) / px
    nl = sum(v * COST[k] for k, v in tot.items() if k.startswith("lds")) / px
    print("VALU: %.1f instr per px (= wave-instr per 64 px), %.0f issue cycles per px-wave; LDS: %.0f cycles" % (nv, cyc, nl))


if __name__ == "__main__":
    sys.exit(main())
