#!/usr/bin/env python3
"""Winograd weight gradient in the persistent-slab ACCUMULATE phase (what 22 of the 23 BPTT steps run)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from refid_amd import ops
from bench_kernels import timeit, B
for name, H, Ca, Cb, Co in [("L0 res 64->64 @256", 256, 64, 0, 64), ("L1 res 128->128 @128", 128, 128, 0, 128),
                            ("L2 res 256->256 @64", 64, 256, 0, 256), ("L0 main 64+64->64 @256", 256, 64, 64, 64)]:
    a = torch.randn(B, H, H, Ca, device="cuda"); b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    g = torch.randn(B, H, H, Co, device="cuda")
    dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
    for algo in (1, 0):
        sl = ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=algo, phase=1, i_total=Ca + Cb)
        t1 = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=algo, phase=1, i_total=Ca + Cb, slabs=sl), iters=20)
        t2 = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=algo, phase=2, i_total=Ca + Cb, slabs=sl), iters=20)
        print(f"{name:24s} algo {algo}: overwrite {t1*1e6:7.1f} us   accumulate {t2*1e6:7.1f} us")
