# tests/debug/fuzz_campaign.py [first_seed] [count] -- run the GPU fuzz tests of tests/test_gpu_fuzz.py over many more seeds than
# the suite does (hunting rare footprint / tile-list / tap-table corner cases).  Prints failures and continues.
import sys, traceback; sys.path.insert(0, '.')
from tests import test_gpu_fuzz as F
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 100
fails = 0
import time
t0 = time.time()
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
print("seeds %d..%d: %d failures" % (first, first + count - 1, fails))
