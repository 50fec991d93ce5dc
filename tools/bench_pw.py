#!/usr/bin/env python3
"""Micro-benchmark of the 1x1 conv tiles (algo 0 = LDS-tiled implicit GEMM, algo 3 = register-operand
pointwise tile) at the BASELINE config-2 shapes.  Prints time and effective HBM GB/s (in + out bytes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops
from bench_kernels import timeit, B

SHAPES = [  # name, H, Ca, Cb, Co
    ("EGACA 64->128 @128", 128, 64, 0, 128), ("EGACA 128->64 @128", 128, 128, 0, 64),
    ("EGACA 64->64 @128", 128, 64, 0, 64), ("fuse L0 64+64->32.. @256", 256, 32, 32, 32),
    ("fuse L1 64+64->64 @128", 128, 64, 64, 64), ("fuse L2 128+128->128 @64", 64, 128, 128, 128),
    ("identity 32->32 @256", 256, 32, 0, 32),
]
for name, H, Ca, Cb, Co in SHAPES:
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
    out = torch.empty(B, H, H, Co, device="cuda")
    bias = torch.randn(Co, device="cuda")
    res = []
    for algo in (0, 3):
        kc, bn = (ops.conv_kc(1, 1, 1, 0), ops.conv_bn(1, 1, 1, 0, Co)) if algo == 0 else (8, 32)
        wp = ops.pack_conv_weights(w, ops.ROLE_FWD, bn, kc, 1, 1, Co, Ci)
        cp = -(-Co // bn) * bn
        t = timeit(lambda: ops.conv2d(a, wp, out, kh=1, kw=1, cout=Co, cout_pad=cp, in_b=b, bias=bias,
                                      slope_pre=0.1, algo=algo), iters=20)
        res.append(t)
    by = 4.0 * B * H * H * (Ci + Co)
    print(f"{name:28s} algo0 {res[0]*1e6:7.1f} us {by/res[0]/1e9:7.0f} GB/s | algo3 {res[1]*1e6:7.1f} us {by/res[1]/1e9:7.0f} GB/s")
