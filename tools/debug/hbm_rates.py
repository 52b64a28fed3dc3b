# tools/debug/hbm_rates.py -- what plain torch kernels reach on this chip: fill (pure write), copy (read + write), reduce (pure read)
import time, torch
n = 1 << 30  # 1 GiB
a = torch.empty(n, dtype=torch.uint8, device="cuda").view(torch.float32)
b = torch.empty_like(a)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it
print("fill   (write %d MiB): %.2f TB/s" % (n >> 20, n / t(lambda: a.fill_(1.0)) / 1e12))
print("copy   (read+write)  : %.2f TB/s total" % (2 * n / t(lambda: b.copy_(a)) / 1e12))
print("sum    (read)        : %.2f TB/s" % (n / t(lambda: a.sum()) / 1e12))
