# tests/debug/fuzz_campaign.py [first_seed] [count] -- run the GPU fuzz tests of tests/test_gpu_fuzz.py over many more seeds than
# the suite does (hunting rare footprint / tile-list / tap-table corner cases).  Prints failures and continues.
import sys, traceback; sys.path.insert(0, '.')
from tests import test_gpu_fuzz as F
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 100
fails = 0
import time
t0 = time.time()
def bounds_counters():
    """device counters of a CHECKED build (OVRFSR_LIB=ab/bounds.so, csrc/fsr_bounds.h): (per-kind out-of-bounds, pad, checked, first record) or None"""
    try:
        import ctypes
        import openvr_fsr_amd as A
        lib = A.library()
        n = lib.ovrfsr_debug_bounds_slots()
        buf = (ctypes.c_ulonglong * n)()
        lib.ovrfsr_debug_bounds.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
        if lib.ovrfsr_debug_bounds(buf, n, 0) != 0:
            return None
        nk = (n - 5) // 3
        v = list(buf)
        return v[:nk], v[nk:2 * nk], v[2 * nk:3 * nk], v[3 * nk:]
    except AttributeError:
        return None


checked_build = bounds_counters() is not None
last_oob = 0
for seed in range(first, first + count):
    if (seed - first) % 25 == 0:   # progress survives a timeout: the summary line only prints at the end
        print("progress: seeds %d..%d done, %d failures, %.0f s" % (first, seed - 1, fails, time.time() - t0), flush=True)
    for fn in (F.test_fsr_fuzz_strict, F.test_nis_fuzz_strict, F.test_masked_product_fuzz, F.test_nis_masked_product_fuzz,
               F.test_masked_product_fuzz_half, F.test_ctx_lifecycle_stress):
        try:
            fn(None, seed)
        except Exception:
            fails += 1
            print("FAIL", fn.__name__, seed)
            traceback.print_exc(limit=2)
        if checked_build:   # round 6: under a checked build the campaign doubles as a bounds campaign -- name the instance that trips a counter
            c = bounds_counters()
            if sum(c[0]) != last_oob:
                print("OUT OF BOUNDS in %s seed %d: per-kind %s (cumulative), first record of the run %s" % (fn.__name__, seed, {i: x for i, x in enumerate(c[0]) if x}, c[3]), flush=True)
                last_oob = sum(c[0])
print("seeds %d..%d: %d failures" % (first, first + count - 1, fails))
if checked_build:
    c = bounds_counters()
    print("checked build: %d checked accesses, %d OUT OF BOUNDS, %d declared-pad accesses" % (sum(c[2]), sum(c[0]), sum(c[1])))
