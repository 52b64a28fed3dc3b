"""bench.py's self-check legs on CPU: `parity_check` (the oracle on the images the timed launch processed) and the
`cpu_baseline` sample that shares those oracle calls.  Tiny stand-in workloads are registered for the test; the GPU side is
replaced by the oracle's own output with a known perturbation, so the records' arithmetic and the ok / not-ok decision are what
is tested here (the GPU run of the same code is BENCH_rNN.json's parity_check)."""
import argparse

import numpy as np
import pytest
import torch

import bench
from tests import synth


@pytest.fixture()
def tiny_workloads():
    added = {"T2": (48, 40, 64, 53, torch.uint8, 2.0, 0), "T2r": (48, 40, 64, 53, torch.uint8, 0.6, 0),
             "T3": (48, 40, 64, 53, torch.uint8, 2.0, 1), "T5": (48, 40, 64, 53, torch.float16, 0.6, 0),
             "T2s": (64, 53, 64, 53, torch.uint8, 2.0, 0), "T3s": (64, 53, 64, 53, torch.uint8, 2.0, 1),
             "Tsbs": (96, 40, 128, 53, torch.uint8, 0.6, 0)}
    bench.WORKLOADS.update(added)
    bench.SHARED.add("Tsbs")
    yield added
    for k in added:
        bench.WORKLOADS.pop(k)
    bench.SHARED.discard("Tsbs")


def _args(workload, precision="fp32", pairs=3):
    return argparse.Namespace(workload=workload, precision=precision, pairs=pairs)


def _inputs(workload, n):
    inW, inH, outW, outH, dtype, radius, use_nis = bench.WORKLOADS[workload]
    imgs = [synth.structured_u8(inW, inH, 100 + i) for i in range(n)]
    if dtype == torch.float16:
        imgs = [(im.astype(np.float32) / 255.0).astype(np.float16) for im in imgs]
    return imgs


def test_check_indices_first_and_last_pair():
    assert bench.check_indices(128) == [0, 1, 126, 127]
    assert bench.check_indices(2) == [0, 1]
    assert bench.check_indices(1) == [0]
    assert bench.check_indices(3) == [0, 1, 2]


@pytest.mark.parametrize("workload", ["T2", "T2r", "T3", "T2s", "T3s", "Tsbs"])
def test_parity_check_unorm8(tiny_workloads, workload):
    ipp = bench.images_per_pair(workload)
    n = ipp * 3
    idx = bench.check_indices(n)
    imgs = _inputs(workload, n)
    want = [bench.oracle_expected(workload, imgs[i], i) for i in idx]
    assert want[0].dtype == np.uint8 and want[0].shape[:2] == bench.WORKLOADS[workload][3:1:-1]
    # eye of image i is i & 1: with a mask the two eyes of a pair differ only through their index (same centre here), but
    # the oracle must be called per image, not once per pair
    got = [w.copy() for w in want]
    got[-1][3, 5, 1] ^= 1                                   # one byte off by one LSB: inside the product tolerance
    par, cpu = bench.parity_and_cpu(_args(workload), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=True, budget_s=0.2, max_pairs=4)
    assert par["ok"] and par["max_lsb"] == 1 and par["n_diff"] == 1 and par["images"] == len(idx) and par["image_indices"] == idx
    assert par["n_total"] == sum(w.size for w in want)
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["single_thread_value"] > 0 and cpu["cores"] >= 1
    # the strict build is held to bit-exactness
    par, cpu = bench.parity_and_cpu(_args(workload, "strict"), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=False)
    assert not par["ok"] and cpu is None
    got[0][0, 0, 0] = np.uint8((int(got[0][0, 0, 0]) + 2) % 256)   # two LSB: outside every tolerance
    par, _ = bench.parity_and_cpu(_args(workload), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=False)
    assert not par["ok"] and par["max_lsb"] >= 2 and par["n_gt1"] >= 1


def test_parity_check_half(tiny_workloads):
    n = 4
    idx = bench.check_indices(n)
    imgs = _inputs("T5", n)
    want = [bench.oracle_expected("T5", imgs[i], i) for i in idx]
    assert want[0].dtype == np.float16
    got = [w.copy() for w in want]
    got[1][2, 2, 0] = np.float16(float(got[1][2, 2, 0]) + 4.0e-4)   # under 1e-3
    par, _ = bench.parity_and_cpu(_args("T5", pairs=2), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=False)
    assert par["ok"] and par["max_lsb"] is None and 0 < par["max_abs"] <= 1e-3 and par["n_gt_1e3"] == 0
    got[1][2, 2, 0] = np.float16(float(got[1][2, 2, 0]) + 4.0e-3)
    par, _ = bench.parity_and_cpu(_args("T5", pairs=2), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=False)
    assert not par["ok"] and par["n_gt_1e3"] == 1
    got[0][0, 0, 1] = np.float16(np.nan)
    par, _ = bench.parity_and_cpu(_args("T5", pairs=2), [(imgs[i], g) for i, g in zip(idx, got)], idx, want_cpu=False)
    assert not par["ok"] and par["n_nan"] == 1


def test_shared_workloads_count_one_image_per_pair():
    assert bench.images_per_pair("C2") == 2 and bench.images_per_pair("C2sbs") == 1 and bench.images_per_pair("C2sbsr") == 1
    inW, inH, outW, outH = bench.WORKLOADS["C2sbs"][:4]
    assert (inW, outW) == (2 * 1683, 2 * 2244) and (inH, outH) == (1869, 2492)   # both eyes side by side: same pixels per pair as C2
