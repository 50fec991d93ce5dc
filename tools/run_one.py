#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc):  run_one.py fwd|wgrad algo H Ca Cb Co"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops
kind, algo, H, Ca, Cb, Co = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
B, Ci = 8, int(sys.argv[4]) + int(sys.argv[5])
a = torch.randn(B, H, H, Ca, device="cuda"); b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
out = torch.empty(B, H, H, Co, device="cuda"); bias = torch.randn(Co, device="cuda")
if kind == "fwd":
    if algo:
        wp = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci); cp = -(-Co // 64) * 64
    else:
        bn = ops.conv_bn(3, 3, 1, 0, Co); wp = ops.pack_conv_weights(w, ops.ROLE_FWD, bn, 8, 3, 3, Co, Ci); cp = -(-Co // bn) * bn
    for _ in range(5):
        ops.conv2d(a, wp, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=cp, in_b=b, bias=bias, slope_pre=0.1, algo=algo)
else:
    g = torch.randn(B, H, H, Co, device="cuda"); dw = torch.zeros(Co, Ci, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
    for _ in range(5):
        ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=algo)
torch.cuda.synchronize()
