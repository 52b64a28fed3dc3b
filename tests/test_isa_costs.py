"""tools/isa_costs.py: the control-flow graphs it reads out of the shipped code objects and the issue-cycle interval it derives from
per-wave instruction counters by linear programming (what bench.py reports as roofline.valu.issue)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_costs as I  # noqa: E402

LIB = os.path.join(ROOT, "openvr_fsr_amd", "libopenvr_fsr_amd.so")


def _cfg(blocks, edges, exits):
    key = {c: i for i, c in enumerate(I.COUNT_KEYS)}
    rows = []
    for b in blocks:
        r = [0] * len(I.COUNT_KEYS)
        for k, v in b.items():
            r[key[k]] = v
        rows.append(r)
    return {"blocks": rows, "edges": edges, "exits": exits}


def test_classification_follows_the_measured_issue_classes():
    assert I.classify("v_fma_f32") == "fast" and I.classify("v_mul_f32_e32") == "fast" and I.classify("v_add_u32_e32") == "fast"
    assert I.classify("v_min3_f32") == "slow" and I.classify("v_cvt_f32_ubyte0_e32") == "slow" and I.classify("v_cndmask_b32_e64") == "slow"
    assert I.classify("v_mov_b32_dpp") == "slow"          # a DPP modifier moves any op into the slow class
    assert I.classify("v_pk_fma_f32") == "pk" and I.classify("v_rcp_f32_e32") == "trans"
    assert I.classify("ds_read_b128") == "lds" and I.classify("global_load_dwordx4") == "vmem_rd" and I.classify("global_store_dword") == "vmem_wr"
    assert I.classify("s_cbranch_execz") == "branch" and I.classify("s_waitcnt") == "salu" and I.classify("s_load_dwordx2") == "smem"


def test_cfg_from_a_disassembly_listing():
    text = ("0000000000001000 <_Z1kPf>:\n"
            "\ts_load_dwordx2 s[0:1], s[0:1], 0x0                 // 000000001000: C0060000 00000000\n"
            "\tv_mov_b32_e32 v1, 0                                 // 000000001008: 7E020280\n"
            "\tv_fma_f32 v1, v1, v2, v3                            // 00000000100C: D1CB0001 040E0501\n"     # loop body (target of the back edge)
            "\tds_read_b128 v[4:7], v0                             // 000000001014: D9FE0000 04000000\n"
            "\ts_cbranch_scc1 65533                                // 00000000101C: BF85FFFD <_Z1kPf+0xc>\n"
            "\tglobal_store_dword v0, v1, s[0:1]                   // 000000001020: DC708000 00000100\n"
            "\ts_endpgm                                            // 000000001028: BF810000\n")
    ks = I.parse_disassembly(text, lambda s: True)
    cfg = I.build_cfg(ks["_Z1kPf"])
    key = {c: i for i, c in enumerate(I.COUNT_KEYS)}
    assert len(cfg["blocks"]) == 3 and cfg["exits"] == [2]
    assert sorted(cfg["edges"]) == [(0, 1), (1, 1), (1, 2)]
    assert cfg["blocks"][1][key["fast"]] == 1 and cfg["blocks"][1][key["lds"]] == 1 and cfg["blocks"][2][key["vmem_wr"]] == 1
    # the loop ran 10 times: 1 + 10 VALU, 10 LDS, 1 store per wave -> the block counts are fully determined, so is the cost
    b = I.issue_bounds(cfg, {"valu": 11.0, "lds": 10.0, "vmem_wr": 1.0, "vmem_rd": 0.0})
    assert b["lo"] == pytest.approx(11 * I.COST["fast"], rel=0.03) and b["hi"] == pytest.approx(11 * I.COST["fast"], rel=0.03)


def test_bounds_bracket_the_truth_when_alternatives_are_indistinguishable():
    # entry -> (A | B) -> exit, A and B with the same VALU count but different classes: counters that only see the VALU total cannot
    # tell them apart; the interval spans both, and an extra counter (LDS, only in A) closes it
    cfg = _cfg([{"fast": 2}, {"fast": 10, "lds": 2}, {"pk": 10}, {"slow": 1}], [(0, 1), (0, 2), (1, 3), (2, 3)], [3])
    truth_A = 2 * I.COST["fast"] + 10 * I.COST["fast"] + I.COST["slow"]
    truth_B = 2 * I.COST["fast"] + 10 * I.COST["pk"] + I.COST["slow"]
    b = I.issue_bounds(cfg, {"valu": 13.0})
    assert b["lo"] <= truth_A * 1.02 and b["hi"] >= truth_B * 0.98
    b = I.issue_bounds(cfg, {"valu": 13.0, "lds": 2.0})
    assert b["hi"] <= truth_A * 1.03
    b = I.issue_bounds(cfg, {"valu": 13.0, "lds": 0.0})
    assert b["lo"] >= truth_B * 0.97
    # an optional counter that contradicts the graph is dropped instead of making the problem infeasible
    b = I.issue_bounds(cfg, {"valu": 13.0, "lds": 2.0, "salu": 999.0})
    assert b is not None and "salu" not in b["constraints"] and "lds" in b["constraints"]
    assert I.issue_bounds(cfg, {"valu": 500.0}) is None    # no path through this graph executes 500 VALU instructions


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_table_of_the_shipped_library(tmp_path):
    out = str(tmp_path / "costs.json")
    n = I.emit(LIB, out)
    doc = I.load(out)
    assert n == len(doc["kernels"]) and n >= 100
    cfg = I.find_kernel(doc, "void ovrfsr_fast::easu_fast_kernel<0, 0, 28, false>(ovrfsr::EasuArgs)")
    assert cfg is not None and len(cfg["blocks"]) > 20 and cfg["exits"]
    for name in ("ovrfsr_fast::rcas_dpp_kernel<0, false, 32>(ovrfsr::RcasArgs)", "ovrfsr_fast::nis_scaler_kernel<0, 0, 32>(ovrfsr::NisArgs)",
                 "ovrfsr_fast::fused_kernel<1, 1, 1, 32, 256>(ovrfsr::FusedArgs)"):
        assert I.find_kernel(doc, name) is not None, name
    # every edge joins existing blocks, the entry block is block 0, strict kernels are not in the table
    nb = len(cfg["blocks"])
    assert all(0 <= u < nb and 0 <= v < nb for u, v in cfg["edges"]) and not any("ovrfsr_strict" in k for k in doc["kernels"])
    # round 3's measured per-wave counters of this kernel (profiles/r03_final_sq/sq_C2.txt): the interval contains the hand-weighted
    # table of profiles/r03_easu_isa_classes.txt (865 nominal = ~830 true cycles per 64 px) and is a few per cent wide
    W = 177216.0
    b = I.issue_bounds(cfg, {"valu": 174544156 / W, "lds": 12422708 / W, "vmem_rd": 200736 / W, "vmem_wr": 754196 / W})
    px64 = 8 * 2244 * 2492 / 64.0
    lo, hi = b["lo"] * W / px64, b["hi"] * W / px64
    assert 700 < lo < hi < 1000 and (hi - lo) / lo < 0.15
