#!/usr/bin/env python3
"""1x1 weight-gradient kernels at the config-2 shapes: REFID_PW_WGRAD=0 (LDS tile) vs 1 (register-operand tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from refid_amd import ops
from bench_kernels import timeit, B
for name, H, Ca, Cb, Co in [("EGACA 64->128 @128", 128, 64, 0, 128), ("EGACA 128->64 @128", 128, 128, 0, 64),
                            ("EGACA 64->64 @128", 128, 64, 0, 64), ("fuse L0 64+64->64 @256", 256, 64, 64, 64),
                            ("fuse L1 128+128->128 @128", 128, 128, 128, 128), ("fuse L2 256+256->256 @64", 64, 256, 256, 256),
                            ("identity 32->64 @256", 256, 32, 0, 64)]:
    a = torch.randn(B, H, H, Ca, device="cuda"); b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    g = torch.randn(B, H, H, Co, device="cuda")
    dw = torch.zeros(Co, Ca + Cb, 1, 1, device="cuda"); db = torch.zeros(Co, device="cuda")
    sl = ops.conv2d_wgrad(g, a, dw, kh=1, kw=1, in_b=b, db=db, phase=1, i_total=Ca + Cb)
    t = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=1, kw=1, in_b=b, db=db, phase=2, i_total=Ca + Cb, slabs=sl), iters=20)
    by = 4.0 * B * H * H * (Ca + Cb + Co)
    print(f"{name:28s} {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s (inputs only)")
