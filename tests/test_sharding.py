"""The N>1 path of bench.py on CPU: two processes over gloo.  Batches shard with no data-path collective; the
only communication is the timing barrier and the max-over-ranks reduction, exercised here exactly as bench.py
runs them (the functions are imported from bench.py).  The per-rank 'hot path' is the CPU oracle on a tiny shard,
which also proves that rank-local results do not depend on the world size."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PAIRS = 2
STEPS = 3


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import time
    import bench
    from oracle import oracle as O
    from tests import synth
    base = bench.shard_seed(PAIRS, rank)
    seeds = [base + i for i in range(2 * PAIRS)]
    imgs = [synth.structured_u8(24, 20, s) for s in seeds]
    sums = []

    def step():
        sums.clear()
        for i, im in enumerate(imgs):  # image i of the shard is eye i & 1
            out = O.fsr_pipeline_u8(im, 32, 27, sharpness=0.9, radius=0.6, eye=i & 1, proj=(0.4, 0.5, 0.6, 0.5), nthreads=1)
            sums.append(int(out.astype(np.uint64).sum()))
        time.sleep(0.02 * (rank + 1))  # uneven ranks: the reported time must be the slowest one

    def barrier():
        dist.barrier()

    dt_local = bench.timed_region(step, STEPS, barrier)
    dt = bench.max_over_ranks(dt_local, world, torch.device("cpu"))
    q.put((rank, seeds, list(sums), dt_local, dt))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, seeds0, sums0, dtl0, dt0), (r1, seeds1, sums1, dtl1, dt1) = res
    # shards are disjoint, contiguous, and follow the documented seed rule 0x5EED0000 + 2*pair + eye
    assert seeds0 == [0x5EED0000 + i for i in range(2 * PAIRS)]
    assert seeds1 == [0x5EED0000 + 2 * PAIRS + i for i in range(2 * PAIRS)]
    # every rank reports the same, slowest, time
    # (the closing barrier is inside the timed region, so local times already include the wait for the slowest rank)
    assert dt0 == dt1 and dt0 >= max(dtl0, dtl1) - 1e-9 and dt0 >= STEPS * 0.04
    # whole-job value = units of all ranks / max time
    value = PAIRS * 2 * STEPS / dt0
    assert value > 0
    # rank-local results equal a single-process run of the same global images (no cross-rank dependence)
    from oracle import oracle as O
    from tests import synth
    for seeds, sums in ((seeds0, sums0), (seeds1, sums1)):
        for i, s in enumerate(seeds):
            out = O.fsr_pipeline_u8(synth.structured_u8(24, 20, s), 32, 27, sharpness=0.9, radius=0.6, eye=i & 1,
                                    proj=(0.4, 0.5, 0.6, 0.5), nthreads=1)
            assert int(out.astype(np.uint64).sum()) == sums[i]
