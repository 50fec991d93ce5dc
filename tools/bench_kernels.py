#!/usr/bin/env python3
"""Micro-benchmark of the conv tile / wgrad entry points at the BASELINE config-2 shapes
(B=8, 256x256).  Prints achieved fp32 TFLOP/s per shape (peak 157.3)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops

B = int(os.environ.get("B", 8))


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def fwd(name, H, Ca, Cb, Co, k, s, p, mode=0):
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
    kc, bn = ops.conv_kc(k, k, s, mode), ops.conv_bn(k, k, s, mode, Co)
    wp = ops.pack_conv_weights(w, ops.ROLE_FWD, bn, kc, k, k, Co, Ci)
    Ho = (H + 2 * p - k) // s + 1
    out = torch.empty(B, Ho, Ho, Co, device="cuda")
    bias = torch.randn(Co, device="cuda")
    cp = -(-Co // bn) * bn
    t = timeit(lambda: ops.conv2d(a, wp, out, kh=k, kw=k, stride=s, pad=p, cout=Co, cout_pad=cp, in_b=b,
                                  bias=bias, slope_pre=0.1))
    fl = 2.0 * B * Ho * Ho * Co * Ci * k * k
    print(f"fwd  {name:28s} {t*1e6:9.1f} us  {fl/t/1e12:7.2f} TFLOP/s")
    return t


def wgrad(name, H, Ca, Cb, Co, k, s, p):
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    Ho = (H + 2 * p - k) // s + 1
    g = torch.randn(B, Ho, Ho, Co, device="cuda")
    dw = torch.zeros(Co, Ci, k, k, device="cuda")
    db = torch.zeros(Co, device="cuda")
    t = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=k, kw=k, stride=s, pad=p, in_b=b, db=db))
    fl = 2.0 * B * Ho * Ho * Co * Ci * k * k
    print(f"wgrd {name:28s} {t*1e6:9.1f} us  {fl/t/1e12:7.2f} TFLOP/s")
    return t


if __name__ == "__main__":
    print("CUs:", ops.lib().refid_device_cu_count())
    for fn in (fwd, wgrad):
        fn("L0 first 32->64 @256", 256, 32, 0, 64, 3, 1, 1)
        fn("L0 main.0 128->64 @256", 256, 64, 64, 64, 3, 1, 1)
        fn("L0 res 64->64 @256", 256, 64, 0, 64, 3, 1, 1)
        fn("L1 main.0 256->128 @128", 128, 128, 128, 128, 3, 1, 1)
        fn("L1 res 128->128 @128", 128, 128, 0, 128, 3, 1, 1)
        fn("L2 main.0 512->256 @64", 64, 256, 256, 256, 3, 1, 1)
        fn("L2 res 256->256 @64", 64, 256, 0, 256, 3, 1, 1)
        fn("bottleneck 256->256 @32", 32, 256, 0, 256, 3, 1, 1)
        fn("D2 res 32->32 @256", 256, 32, 0, 32, 3, 1, 1)
        fn("down 64 @256->128", 256, 64, 0, 64, 4, 2, 1)
        fn("down 256 @64->32", 64, 256, 0, 256, 4, 2, 1)
        fn("fuse1x1 128->64 @256", 256, 64, 64, 64, 1, 1, 0)
        fn("fuse1x1 512->256 @64", 64, 256, 256, 256, 1, 1, 0)
    fwd("head5x5 4->32 @256", 256, 4, 0, 32, 5, 1, 2)
