#!/usr/bin/env python3
"""Winograd 3x3 tile, hot vs cold operands: the same launch over ONE buffer pair (operands stay in the 256 MB
MALL / L2 between launches) and cycling over many distinct buffer pairs (every launch streams from HBM, as
inside a training step where each activation is a fresh tensor)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from refid_amd import ops
from bench_kernels import B

def run(name, H, C, Co, nbuf):
    w = torch.randn(Co, C, 3, 3, device="cuda") * 0.05
    ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, C)
    bias = torch.randn(Co, device="cuda")
    ins = [torch.randn(B, H, H, C, device="cuda") for _ in range(nbuf)]
    outs = [torch.empty(B, H, H, Co, device="cuda") for _ in range(nbuf)]
    cw = -(-Co // 64) * 64
    def go(i):
        ops.conv2d(ins[i % nbuf], ww, outs[i % nbuf], kh=3, kw=3, pad=1, cout=Co, cout_pad=cw, bias=bias, slope_pre=0.1, algo=1)
    for i in range(nbuf): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(40, 2 * nbuf)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    fl = 2.0 * B * H * H * Co * C * 9
    print(f"{name:24s} buffers {nbuf:3d}: {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TF(eff)")

for nbuf in (1, 40):
    run("L0 64->64 @256", 256, 64, 64, nbuf)
    run("L1 128->128 @128", 128, 128, 128, nbuf)
    run("L2 256->256 @64", 64, 256, 256, nbuf)
