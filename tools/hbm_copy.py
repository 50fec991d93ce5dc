#!/usr/bin/env python3
"""Practical HBM ceiling of the box: device-to-device copy and a read-only reduction at several sizes
(bytes moved = read + written).  The 8 TB/s in MI355X_MICROARCH.md is the pin rate; this is what a perfectly
coalesced streaming kernel reaches, i.e. the realistic bound for the HBM-bound tiles."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda").normal_(); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); ta = t(lambda: torch.add(a, b, out=b)); tr = t(lambda: a.sum())
    print(f"{mb:5d} MB  copy {2*n*4/tc/1e12:5.2f} TB/s   add(2r1w) {3*n*4/ta/1e12:5.2f} TB/s   sum(read) {n*4/tr/1e12:5.2f} TB/s")
