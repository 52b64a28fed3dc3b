"""A shard without a GPU for tests/test_bench_launcher.py: bench.py builds it instead of GpuShard when
OVRFSR_BENCH_SHARD_FACTORY=tests.mock_shard:make is set, so the launcher, the timing protocol and the JSON contract can be
exercised with the exact commands the round driver runs.  A step is a sleep whose length grows with the shard index (uneven
shards: the reported time must be the slowest one's)."""
import threading
import time

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

    def close(self):
        pass


def make(device_index, shard_index, args):
    return MockShard(device_index, shard_index, args)
