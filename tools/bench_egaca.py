#!/usr/bin/env python3
"""Bandwidth of the HBM-bound EGACA kernels at the config-2 shape (B=8, 128x128, 64 channels): us per launch and
algorithmic TB/s (each tensor read / written once) against the 8 TB/s HBM3E peak."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops

B, H, W, C = int(os.environ.get("B", 8)), 128, 128, 64


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def row(name, t, tensors):
    nbytes = tensors * B * H * W * C * 4
    print(f"{name:34s} {t * 1e6:8.1f} us   {nbytes / t / 1e12:5.2f} TB/s   {nbytes / t / 8e12:5.2f} of 8 TB/s")


x = torch.randn(B, H, W, C, device="cuda")
g = torch.randn(B, H, W, C, device="cuda")
w9 = torch.randn(C, 1, 3, 3, device="cuda"); b = torch.randn(C, device="cuda")
lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
dw9 = torch.zeros_like(w9); db = torch.zeros_like(b); dlw = torch.zeros_like(lw); dlb = torch.zeros_like(lb)
gx = torch.empty_like(x)
row("ln_fwd (1 in, 1 out)", timeit(lambda: ops.layernorm2d_fwd(x, lw, lb, out=gx)), 2)
row("ln_bwd (g, x in; gx out)", timeit(lambda: ops.layernorm2d_bwd(g, x, lw, gx, dlw, dlb)), 3)
row("ln_bwd + res (3 in, 1 out)", timeit(lambda: ops.layernorm2d_bwd(g, x, lw, gx, dlw, dlb, res=g)), 4)
row("dw3x3+gelu fwd (1 in, 2 out)", timeit(lambda: ops.dwconv3x3_gelu_fwd(x, w9, b, want_pool=True)), 3)
row("dw3x3 bwd (2 in, 1 out)", timeit(lambda: ops.dwconv3x3_bwd(g, x, w9, dw9, db)), 3)
row("gelu_bwd (2 in, 1 out)", timeit(lambda: ops.gelu_bwd(g, x, out=gx)), 3)
row("add (2 in, 1 out)", timeit(lambda: ops.add(g, x, out=gx)), 3)
row("act_bwd (2 in, 1 out)", timeit(lambda: ops.act_bwd(g, x, 0.2, out=gx)), 3)
