#!/usr/bin/env python3
"""tools/isa_costs.py -- per-kernel issue-cost tables of the SHIPPED gfx950 code objects, and the issue-cycle bounds they imply.

Build step (called by __graft_entry__.build()):

    tools/isa_costs.py --emit [--lib openvr_fsr_amd/libopenvr_fsr_amd.so] [--out openvr_fsr_amd/kernel_issue_costs.json]

extracts the gfx950 ELFs from the library's fat binary, disassembles them with llvm-objdump and writes, for every
`ovrfsr_fast::` kernel, its control-flow graph: basic blocks (split at branch targets and behind branches) with their
instruction counts per issue class, and the edges between them.  Nothing is typed by hand: blocks, classes and edges come from
the disassembly.

Run step (bench.py): how often each block executes is NOT in the code object.  It is recovered from the hardware counters of the
run itself: with x_b the executions of block b per wave and y_e the traversals of edge e,

    flow conservation      x_b = [b is the entry] + sum of y_e into b = sum of y_e out of b (+ exits at s_endpgm)
    counters (per wave)    sum_b x_b * valu_b = SQ_INSTS_VALU / SQ_WAVES, and the same for SQ_INSTS_LDS, SQ_INSTS_VMEM_RD,
                           SQ_INSTS_VMEM_WR (each within a small tolerance)

is a linear system; `issue_bounds()` minimises and maximises  sum_b x_b * (issue cycles of block b)  over it (scipy linprog).
The result is an INTERVAL [lo, hi] of VALU issue cycles per wave that every execution profile consistent with the counters must
lie in -- narrow when the alternatives the counters cannot tell apart have similar instruction mixes, which is the case here.

Issue costs per class (true shader cycles per wave-instruction per SIMD at 8 waves per SIMD; tools/ubench/valu_rates.hip
measures each instruction's time AND, with s_memtime / s_memrealtime inside the kernel, the clock it ran at,
profiles/r04_valu_issue_rates.txt):
    fast   2.35  v_fma/mul/add/sub_f32 (also with clamp / |x| / -x modifiers), v_mov, 2-operand integer add/sub/shift/and/or
    slow   4.15  v_min/max/med3, v_cvt_*, v_cmp, v_cndmask, 3-operand integer ops, v_mul_lo/hi, every DPP-modified op, v_fma_mix ...
    pk     4.15  v_pk_* (f32 and f16)
    trans  8.15  v_rcp/rsq/sqrt/exp/log/sin/cos
MODEL ERROR.  The costs are those of homogeneous streams and the model adds them up.  Mixed streams measured in the same file
deviate: fast and slow ops alternating cost 0.69-0.81 of the sum (they overlap: 2.26-2.64 instead of 3.25 cycles per
instruction), fast + packed 1.18 (3.84 instead of 3.25), fast + transcendental 1.17, slow + packed and slow / packed +
transcendental 1.00, and a stream shaped like EASU's pair block (3 packed, 4 fast, 1 slow per 8) 1.06.  So a kernel's issue
cycles carry about -15 % ... +6 % of model error on top of the LP interval: an issue_frac of 0.9-1.1 reads "the VALU issues
back to back", 0.7 reads "it does not" -- which is the distinction the figure is for.
"""
import argparse
import gzip
import json
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

COST = {"fast": 2.35, "slow": 4.15, "pk": 4.15, "trans": 8.15}
VALU_CLASSES = ("fast", "slow", "pk", "trans")
COUNT_KEYS = VALU_CLASSES + ("salu", "lds", "vmem_rd", "vmem_wr", "smem", "branch", "other")

FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32",
        "v_mul_legacy_f32", "v_not_b32", "v_accvgpr", "v_addc_co_u32", "v_fmamk_f32", "v_fmaak_f32", "v_madmk_f32", "v_madak_f32",
        "v_add_i32", "v_sub_i32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_nop")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        if op.endswith(("_dpp", "_sdwa")):
            return "slow"   # measured: a DPP-modified op issues in the slow class whatever the base op
        base = re.sub(r"_(e32|e64)$", "", op)
        return "fast" if base.startswith(FAST) else "slow"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_rd"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_wr"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def extract_code_objects(lib, tmp):
    """The gfx950 ELFs of a fat binary (one bundle per translation unit)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        (nb,) = struct.unpack_from("<Q", data, p + 24)
        o = p + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                elf = os.path.join(tmp, "co%d.elf" % len(out))
                open(elf, "wb").write(data[p + off:p + off + size])
                out.append(elf)
    return out


_INS = re.compile(r"^\s+(\S+)(.*?)//\s*([0-9A-Fa-f]+):")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_TGT = re.compile(r"<(\S+?)\+0x([0-9a-fA-F]+)>\s*$|<(\S+?)>\s*$")


def parse_disassembly(text, want):
    """{mangled name: [(addr, mnemonic, branch target addr or None)]} for the symbols `want` accepts."""
    kernels, cur, base = {}, None, {}
    lines = text.splitlines()
    for ln in lines:
        m = _SYM.match(ln)
        if m:
            base[m.group(2)] = int(m.group(1), 16)
    for ln in lines:
        m = _SYM.match(ln)
        if m:
            cur = m.group(2) if want(m.group(2)) else None
            if cur:
                kernels[cur] = []
            continue
        if cur is None:
            continue
        m = _INS.match(ln)
        if not m:
            continue
        op, addr = m.group(1), int(m.group(3), 16)
        tgt = None
        if op.startswith(("s_branch", "s_cbranch")):
            t = _TGT.search(ln)
            if t:
                sym, off = (t.group(1), int(t.group(2), 16)) if t.group(1) else (t.group(3), 0)
                tgt = base.get(sym, 0) + off
        kernels[cur].append((addr, op, tgt))
    return kernels


def build_cfg(ins):
    """Basic blocks and edges of one kernel's instruction list."""
    addrs = [a for a, _, _ in ins]
    index = {a: i for i, a in enumerate(addrs)}
    leaders = {addrs[0]}
    for i, (a, op, tgt) in enumerate(ins):
        if classify(op) == "branch":
            if tgt is not None and tgt in index:
                leaders.add(tgt)
            if i + 1 < len(ins):
                leaders.add(addrs[i + 1])
    starts = sorted(leaders)
    bidx = {a: k for k, a in enumerate(starts)}
    blocks = [dict.fromkeys(COUNT_KEYS, 0) for _ in starts]
    edges, exits = [], []
    k = -1
    for i, (a, op, tgt) in enumerate(ins):
        if a in bidx:
            k = bidx[a]
        blocks[k][classify(op)] += 1
        last = i + 1 == len(ins) or addrs[i + 1] in bidx
        if not last:
            continue
        nxt = bidx.get(addrs[i + 1]) if i + 1 < len(ins) else None
        if op.startswith("s_endpgm"):
            exits.append(k)
        elif op.startswith("s_branch"):
            if tgt in bidx:
                edges.append((k, bidx[tgt]))
        elif op.startswith("s_cbranch"):
            if tgt in bidx:
                edges.append((k, bidx[tgt]))
            if nxt is not None:
                edges.append((k, nxt))
        elif nxt is not None:
            edges.append((k, nxt))
        else:
            exits.append(k)
    return {"blocks": [[b[c] for c in COUNT_KEYS] for b in blocks], "edges": sorted(set(edges)), "exits": sorted(set(exits))}


def demangle(names):
    filt = _tool("llvm-cxxfilt") or shutil.which("c++filt")
    if not filt or not names:
        return {n: n for n in names}
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def emit(lib, out_path):
    tmp = tempfile.mkdtemp(prefix="ovrfsr_isa_")
    try:
        table = {}
        for elf in extract_code_objects(lib, tmp):
            text = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", elf], capture_output=True, text=True, check=True).stdout
            ks = parse_disassembly(text, lambda s: "ovrfsr_fast" in s and "kernel" in s)
            names = demangle(list(ks))
            for mangled, ins in ks.items():
                if not ins:
                    continue
                cfg = build_cfg(ins)
                cfg["mangled"] = mangled
                table[re.sub(r"^void ", "", names[mangled])] = cfg
        doc = {"count_keys": list(COUNT_KEYS), "cost": COST, "source": "llvm-objdump -d of the gfx950 code objects in " + os.path.basename(lib),
               "kernels": table}
        with gzip.open(out_path, "wt", compresslevel=9) as f:   # (gzip under the same name: 30 KB instead of 0.7 MB to push to the GPU box)
            json.dump(doc, f, separators=(",", ":"))
        return len(table)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load(path=None):
    path = path or os.path.join(ROOT, "openvr_fsr_amd", "kernel_issue_costs.json")
    with open(path, "rb") as f:
        raw = f.read()
    return json.loads(gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw)


def find_kernel(doc, demangled):
    """Entry of a kernel by the name rocprofv3 reports ("void ns::kernel<...>(args)")."""
    key = re.sub(r"^void ", "", demangled.strip())
    if key in doc["kernels"]:
        return doc["kernels"][key]
    squeeze = lambda s: re.sub(r"\s+", "", s)  # noqa: E731
    for k, v in doc["kernels"].items():
        if squeeze(k) == squeeze(key):
            return v
    return None


CORE_COUNTERS = ("valu", "lds", "vmem_rd", "vmem_wr")          # instruction categories with an unambiguous membership
OPTIONAL_COUNTERS = ("trans", "smem", "branch", "salu")          # used when they fit: dropped (in reverse order) when no profile satisfies them


def issue_bounds(cfg, per_wave, cost=None, tolerances=(0.01, 0.02, 0.05, 0.10)):
    """[lo, hi] VALU issue cycles per wave over every block-execution profile that satisfies flow conservation and the measured
    per-wave instruction counts `per_wave` = {"valu": SQ_INSTS_VALU / SQ_WAVES, "lds": .., "vmem_rd": .., "vmem_wr": .., and
    optionally "trans" (SQ_INSTS_VALU_TRANS_F32), "smem", "branch", "salu"}; missing keys are not constrained.
    Returns {"lo", "hi", "tolerance", "constraints", "valu_per_wave", "mean_cost_lo", "mean_cost_hi"} or None when no profile fits."""
    import numpy as np
    from scipy.optimize import linprog
    cost = cost or COST
    B, E = cfg["blocks"], cfg["edges"]
    nb, ne = len(B), len(E)
    nx = len(cfg["exits"])
    nvar = nb + ne + nx            # x_b | y_e | t_exit
    key = {c: i for i, c in enumerate(COUNT_KEYS)}
    col = lambda name: np.array([b[key[name]] for b in B], float)  # noqa: E731
    cyc = np.array([sum(b[key[c]] * cost[c] for c in VALU_CLASSES) for b in B])
    valu = np.array([sum(b[key[c]] for c in VALU_CLASSES) for b in B], float)
    vectors = {"valu": valu, "lds": col("lds"), "vmem_rd": col("vmem_rd"), "vmem_wr": col("vmem_wr"), "trans": col("trans"),
               "smem": col("smem"), "branch": col("branch"), "salu": col("salu")}
    rows, rhs = [], []
    for b in range(nb):            # inflow
        r = np.zeros(nvar)
        r[b] = 1.0
        for j, (u, v) in enumerate(E):
            if v == b:
                r[nb + j] -= 1.0
        rows.append(r)
        rhs.append(1.0 if b == 0 else 0.0)
    for b in range(nb):            # outflow
        r = np.zeros(nvar)
        r[b] = 1.0
        for j, (u, v) in enumerate(E):
            if u == b:
                r[nb + j] -= 1.0
        if b in cfg["exits"]:
            r[nb + ne + cfg["exits"].index(b)] -= 1.0
        rows.append(r)
        rhs.append(0.0)
    A_eq, b_eq = np.array(rows), np.array(rhs)
    c = np.zeros(nvar)
    c[:nb] = cyc
    core = [n for n in CORE_COUNTERS if per_wave.get(n) is not None]
    optional = [n for n in OPTIONAL_COUNTERS if per_wave.get(n) is not None]

    def solve(names, tol):
        A_ub, b_ub = [], []
        for n in names:
            m = float(per_wave[n])
            r = np.zeros(nvar)
            r[:nb] = vectors[n]
            slack = tol * m + 0.02   # absolute floor: tiny counts (a fraction of a store per wave) carry sampling noise
            A_ub.append(r); b_ub.append(m + slack)
            A_ub.append(-r); b_ub.append(-(m - slack))
        kw = dict(A_ub=np.array(A_ub) if A_ub else None, b_ub=np.array(b_ub) if b_ub else None, A_eq=A_eq, b_eq=b_eq, bounds=(0, None), method="highs")
        lo = linprog(c, **kw)
        if lo.status != 0:
            return None
        hi = linprog(-c, **kw)
        if hi.status != 0:
            return None
        v = per_wave.get("valu") or float(valu @ lo.x[:nb])
        return {"lo": float(lo.fun), "hi": float(-hi.fun), "tolerance": tol, "constraints": list(names), "valu_per_wave": v,
                "mean_cost_lo": float(lo.fun) / v if v else None, "mean_cost_hi": float(-hi.fun) / v if v else None}

    # A counter whose category membership is not certain (does SQ_INSTS_SALU count s_waitcnt? s_nop?) is dropped before the
    # tolerance on the certain ones is widened: for each tolerance, the largest prefix of the optional counters that still fits.
    for tol in tolerances:
        for k in range(len(optional), -1, -1):
            r = solve(core + optional[:k], tol)
            if r is not None:
                return r
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit", action="store_true")
    ap.add_argument("--lib", default=os.path.join(ROOT, "openvr_fsr_amd", "libopenvr_fsr_amd.so"))
    ap.add_argument("--out", default=os.path.join(ROOT, "openvr_fsr_amd", "kernel_issue_costs.json"))
    ap.add_argument("--show", help="print the block table of the kernels whose demangled name contains this string")
    a = ap.parse_args()
    if a.emit:
        n = emit(a.lib, a.out)
        print("wrote %s: %d kernels" % (a.out, n))
    if a.show:
        doc = load(a.out)
        for name, cfg in doc["kernels"].items():
            if a.show in name:
                print(name)
                for i, b in enumerate(cfg["blocks"]):
                    d = dict(zip(doc["count_keys"], b))
                    print("  block %3d  fast %3d slow %3d pk %3d trans %2d | salu %3d lds %2d vmem %d/%d  -> %s%s" % (
                        i, d["fast"], d["slow"], d["pk"], d["trans"], d["salu"], d["lds"], d["vmem_rd"], d["vmem_wr"],
                        [v for u, v in cfg["edges"] if u == i], " EXIT" if i in cfg["exits"] else ""))


if __name__ == "__main__":
    sys.exit(main())
