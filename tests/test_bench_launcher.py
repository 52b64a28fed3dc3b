"""bench.py's launcher on CPU (the step is mocked): `--gpus N` run directly drives N shards -- one per device, disjoint
contiguous seed ranges, one host thread each, EXACTLY K timed steps per shard -- and reports the slowest shard; under
torchrun every rank drives one shard.  (Round 1's bench ignored --gpus when invoked directly.)"""
import threading
import time

import bench


class MockShard:
    def __init__(self, device_index, shard_index, delay):
        self.device_index, self.shard_index, self.delay = device_index, shard_index, delay
        self.steps, self.thread, self.t0, self.t1 = 0, None, None, None
        self.seed = bench.shard_seed(16, shard_index)

    def bind(self):
        self.thread = threading.get_ident()

    def step(self):
        self.steps += 1
        time.sleep(self.delay)

    def sync(self):
        pass

    def mark_start(self):
        self.t0, self.steps_at_start = time.perf_counter(), self.steps

    def mark_end(self):
        self.t1 = time.perf_counter()

    def device_ms(self):
        return (self.t1 - self.t0) * 1e3


def test_plan_direct_invocation_uses_all_gpus():
    args = bench.parse_args(["--gpus", "4"])
    assert bench.plan(args, {}) == (1, 0, [0, 1, 2, 3], 0)
    assert bench.plan(bench.parse_args([]), {}) == (1, 0, [0], 0)


def test_plan_under_torchrun_is_one_device_per_rank():
    args = bench.parse_args(["--gpus", "8"])
    assert bench.plan(args, {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}) == (8, 5, [5], 5)


def test_direct_launcher_creates_and_times_n_shards():
    n, steps, warmup = 4, 5, 2
    shards = bench.build_shards(n, 0, lambda i, s: MockShard(i, s, 0.002 * (i + 1)))
    assert [s.device_index for s in shards] == [0, 1, 2, 3] and [s.shard_index for s in shards] == [0, 1, 2, 3]
    # disjoint, contiguous image seeds: shard s owns global pairs [16 s, 16 (s+1))
    assert [s.seed for s in shards] == [0x5EED0000 + 32 * i for i in range(n)]
    dt, dev_ms = bench.run_local(shards, steps, warmup)
    assert all(s.steps == warmup + steps and s.steps_at_start == warmup for s in shards)   # EXACTLY K timed steps each
    assert len({s.thread for s in shards}) == n                                          # one host thread per device
    assert len(dev_ms) == n and dev_ms[3] > dev_ms[0]
    assert dt >= steps * 0.008 and dt >= max(dev_ms) / 1e3 - 1e-3                         # wall time = the slowest shard
    assert dt < 0.5


def test_single_shard_runs_inline_and_calls_the_cross_rank_barrier_on_both_sides():
    calls = []
    shards = bench.build_shards(1, 3, lambda i, s: MockShard(i, s, 0.0))
    assert shards[0].shard_index == 3 and shards[0].seed == 0x5EED0000 + 32 * 3
    bench.run_local(shards, 3, 1, cross_barrier=lambda: calls.append(shards[0].steps))
    assert calls == [1, 4] and shards[0].thread == threading.get_ident()


def test_a_failing_shard_does_not_hang_the_others():
    class Bad(MockShard):
        def step(self):
            raise RuntimeError("boom")
    shards = [MockShard(0, 0, 0.0), Bad(1, 1, 0.0)]
    try:
        bench.run_local(shards, 2, 1)
    except (RuntimeError, threading.BrokenBarrierError):
        pass
    else:
        raise AssertionError("expected the failure to propagate")


# ------------------------------------------------------------------------------------------------
# the round driver's own commands, end to end, with the GPU shard mocked (tests/mock_shard.py)
# ------------------------------------------------------------------------------------------------
import json  # noqa: E402
import os  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=180, corrupt=None):
    env = dict(os.environ, OVRFSR_BENCH_SHARD_FACTORY="tests.mock_shard:make", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OVRFSR_MOCK_CORRUPT_SHARD"):
        env.pop(k, None)
    if corrupt is not None:
        env["OVRFSR_MOCK_CORRUPT_SHARD"] = str(corrupt)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    if corrupt is not None:   # a wrong shard: the line is still printed (it says which), the exit status is not 0
        assert p.returncode != 0, "a corrupted shard must fail the run"
        assert len(lines) == 1, lines
        return json.loads(lines[0]), p.stderr
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "exactly ONE line on stdout, got %r" % lines   # the driver parses stdout as one JSON object
    return json.loads(lines[0])


def _check_line(d, n, steps, warmup, pairs):
    from tests import mock_shard
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup and d["unit"] == "eye-pairs/s"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    per_dev = d["config"]["per_device_ms_per_step"]
    assert len(per_dev) == n                                               # one entry per GPU of the job, whatever the launcher
    assert per_dev == sorted(per_dev) and per_dev[-1] >= mock_shard.STEP_S * n * 1e3 * 0.9   # shard i sleeps (i+1) x STEP_S
    # whole-job value = N x pairs x steps / time of the slowest shard
    assert d["ms_per_step"] >= per_dev[-1] * 0.95
    assert d["value"] == pytest.approx(n * pairs * steps / (d["ms_per_step"] * 1e-3 * steps), rel=1e-3)
    assert d["config"]["pairs_per_gpu_per_step"] == pairs and d.get("mock_shards") is True
    # every shard of the job proved image 0 of its own batch against the oracle: N records, in shard order, all ok
    ps = d["parity_check"]["per_shard"]
    assert [r["shard"] for r in ps] == list(range(n)) and all(r["ok"] and r["n_diff"] == 0 and r["image"] == 0 for r in ps)
    assert d["parity_check"]["ok"] is True and d["parity_check"]["failed_shards"] == []


@pytest.mark.timeout(300)
def test_driver_command_direct_8_gpus():
    """`python bench.py --gpus 8 --steps K --warmup W`: one process, 8 shards."""
    d = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    _check_line(d, 8, 5, 2, 64)
    assert "one process, 8 device(s)" in d["config"]["launcher"]


@pytest.mark.timeout(300)
def test_driver_command_under_torch_distributed_run():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W` -- the driver's N > 1 launch: one rank per GPU, rank 0 prints the one line, every rank's device time
    arrives through the gloo gather."""
    n = 4
    port = 29600 + (os.getpid() % 300)
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", str(n), "--steps", "5", "--warmup", "2"])
    _check_line(d, n, 5, 2, 64)
    assert "torchrun" in d["config"]["launcher"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("k", [0, 5])
def test_direct_run_fails_and_names_a_corrupted_shard(k):
    """Shard k writes a wrong pixel (tests/mock_shard.py, OVRFSR_MOCK_CORRUPT_SHARD): the run exits non-zero, the line's
    parity_check.per_shard has 8 entries with exactly entry k not ok, and stderr names shard k."""
    d, err = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"], corrupt=k)
    ps = d["parity_check"]["per_shard"]
    assert len(ps) == 8 and [r["shard"] for r in ps if not r["ok"]] == [k] and ps[k]["max_lsb"] == 2
    assert d["parity_check"]["ok"] is False and d["parity_check"]["failed_shards"] == [k]
    assert "shard %d (device %d)" % (k, k) in err


@pytest.mark.timeout(300)
def test_torchrun_fails_and_names_a_corrupted_shard():
    """The same under the driver's torchrun launch: rank 2's record reaches rank 0 through the gloo gather and fails the job."""
    n, k = 4, 2
    port = 29900 + (os.getpid() % 90)
    d, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), "bench.py", "--gpus", str(n), "--steps", "3", "--warmup", "1"], corrupt=k)
    ps = d["parity_check"]["per_shard"]
    assert len(ps) == n and [r["shard"] for r in ps if not r["ok"]] == [k]
    assert "shard %d (device %d)" % (k, k) in err
