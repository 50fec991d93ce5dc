#!/usr/bin/env python3
"""Winograd tile with the residual / mask epilogue variants the train step uses (256^2 x 64 and 128^2 x 128)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from refid_amd import ops
from bench_kernels import timeit, B
for H, C in ((256, 64), (128, 128)):
    a = torch.randn(B, H, H, C, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, C, C)
    out = torch.empty(B, H, H, C, device="cuda"); r = torch.randn_like(out); m = torch.randn_like(out)
    bias = torch.randn(C, device="cuda")
    for name, kw in (("plain", {}), ("mask", dict(mask=m, slope_mask=0.1)), ("res", dict(res=r)), ("res+mask", dict(res=r, mask=m, slope_mask=0.1))):
        t = timeit(lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=C, cout_pad=C, bias=bias, slope_pre=0.1, algo=1, **kw), iters=20)
        print(f"{C:3d} ch @{H}^2 {name:9s} {t*1e6:7.1f} us")
