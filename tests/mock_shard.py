"""A shard without a GPU for tests/test_bench_launcher.py: bench.py builds it instead of GpuShard when
OVRFSR_BENCH_SHARD_FACTORY=tests.mock_shard:make is set, so the launcher, the timing protocol and the JSON contract can be
exercised with the exact commands the round driver runs.  A step is a sleep whose length grows with the shard index (uneven
shards: the reported time must be the slowest one's)."""
import os
import threading
import time

import numpy as np

STEP_S = 0.004


class MockShard:
    def __init__(self, device_index, shard_index, args):
        self.device_index, self.shard_index = device_index, shard_index
        self.delay = STEP_S * (1 + shard_index)
        self.steps, self.thread, self.t0, self.t1 = 0, None, None, None

    def bind(self):
        self.thread = threading.get_ident()

    def step(self):
        self.steps += 1
        time.sleep(self.delay)

    def sync(self):
        pass

    def mark_start(self):
        self.t0 = time.perf_counter()

    def mark_end(self):
        self.t1 = time.perf_counter()

    def device_ms(self):
        return (self.t1 - self.t0) * 1e3

    # the per-shard parity leg of bench.py (every shard proves image 0 of its own batch): a tiny image of this shard's seed, "processed"
    # by the oracle itself -- so a healthy mock shard is bit-identical -- unless OVRFSR_MOCK_CORRUPT_SHARD names this shard, whose
    # output then carries one byte 2 LSB off (outside every tolerance): the run must fail and say which shard
    parity_workload = "Tmock"

    def fetch(self, indices):
        import bench   # the functions only (when bench.py runs as __main__ this is a second, stateless copy)
        from tests import synth
        inW, inH = bench.WORKLOADS[self.parity_workload][:2]
        out = []
        for i in indices:
            src = synth.structured_u8(inW, inH, (bench.shard_seed(4, self.shard_index) + i) & 0x7fffffff)
            got = bench.oracle_expected(self.parity_workload, src, i).copy()
            if os.environ.get("OVRFSR_MOCK_CORRUPT_SHARD", "") == str(self.shard_index):
                got[3, 5, 1] = np.uint8((int(got[3, 5, 1]) + 2) % 256)
            out.append((src, got))
        return out

    def close(self):
        pass


def make(device_index, shard_index, args):
    return MockShard(device_index, shard_index, args)
