#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this card: device-to-device copy, read-only sum and write-only fill of 4 GiB (torch kernels)."""
import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
a.normal_()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: b.copy_(a)); print("copy  (4 GiB read + 4 GiB written): %.2f ms, %.2f TB/s" % (t * 1e3, 2 * 4 * n / t / 1e12))
t = timed(lambda: a.sum());    print("sum   (4 GiB read):                 %.2f ms, %.2f TB/s" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: b.fill_(1.0)); print("fill  (4 GiB written):              %.2f ms, %.2f TB/s" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: torch.add(a, b, out=b)); print("add   (8 GiB read + 4 GiB written): %.2f ms, %.2f TB/s" % (t * 1e3, 3 * 4 * n / t / 1e12))
