#!/usr/bin/env python3
"""PCIe ceiling of the box: pinned host <-> device copies of 4 GiB, one direction and both at once (torch, no engine code)."""
import time, torch
n = 4 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = t(lambda: d.copy_(h, non_blocking=True)); b = t(lambda: h.copy_(d, non_blocking=True))
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
c = t(both)
print("pinned H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f + %.1f GB/s" % (n / a / 1e9, n / b / 1e9, n / c / 1e9, n / c / 1e9))
