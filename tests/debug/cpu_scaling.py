# tests/debug/cpu_scaling.py -- how the CPU oracle scales on this host (threads vs wall time), and what the container allows
import os, sys, time; sys.path.insert(0, '.')
from oracle import oracle as O
from tests import synth
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", O.lib().ovo_max_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except OSError: pass
img = synth.structured_u8(1683, 1869, 1)
O.fsr_pipeline_u8(img[:64, :64].copy(), 85, 85)
for nt in (1, 2, 8, 16, 32, 64, 128, 0):
    t0 = time.perf_counter(); O.fsr_pipeline_u8(img, 2244, 2492, sharpness=0.9, nthreads=nt); dt = time.perf_counter() - t0
    print("threads %3d  %.3f s per eye" % (nt, dt))
