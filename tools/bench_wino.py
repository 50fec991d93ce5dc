#!/usr/bin/env python3
"""Direct vs Winograd 3x3 tile at the config-2 shapes (B=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit, B

def one(name, H, Ca, Cb, Co):
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, H, H, Co, device="cuda")
    bias = torch.randn(Co, device="cuda")
    fl = 2.0 * B * H * H * Co * Ci * 9
    bn = ops.conv_bn(3, 3, 1, 0, Co)
    wp = ops.pack_conv_weights(w, ops.ROLE_FWD, bn, 8, 3, 3, Co, Ci)
    cp = -(-Co // bn) * bn
    t0 = timeit(lambda: ops.conv2d(a, wp, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=cp, in_b=b, bias=bias, slope_pre=0.1))
    ref = out.clone()
    ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    cw = -(-Co // 64) * 64
    t1 = timeit(lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=cw, in_b=b, bias=bias, slope_pre=0.1, algo=1))
    err = (out - ref).abs().max().item()
    print(f"{name:28s} direct {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | winograd {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF(eff) "
          f"x{t0/t1:4.2f}  maxdiff {err:.2e}")

def wg(name, H, Ca, Cb, Co):
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    g = torch.randn(B, H, H, Co, device="cuda")
    dw = torch.zeros(Co, Ci, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
    fl = 2.0 * B * H * H * Co * Ci * 9
    t0 = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db))
    t1 = timeit(lambda: ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=1))
    print(f"wgrad {name:24s} direct {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | winograd {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF(eff) x{t0/t1:4.2f}")


if __name__ == "__main__":
    wg("L0 first 32->64 @256", 256, 32, 0, 64)
    wg("L0 main.0 128->64 @256", 256, 64, 64, 64)
    wg("L0 res 64->64 @256", 256, 64, 0, 64)
    wg("L1 main.0 256->128 @128", 128, 128, 128, 128)
    wg("L1 res 128->128 @128", 128, 128, 0, 128)
    wg("L2 main.0 512->256 @64", 64, 256, 256, 256)
    wg("L2 res 256->256 @64", 64, 256, 0, 256)
    wg("bottleneck 256->256 @32", 32, 256, 0, 256)
    one("L0 first 32->64 @256", 256, 32, 0, 64)
    one("L0 main.0 128->64 @256", 256, 64, 64, 64)
    one("L0 res 64->64 @256", 256, 64, 0, 64)
    one("L1 main.0 256->128 @128", 128, 128, 128, 128)
    one("L1 res 128->128 @128", 128, 128, 0, 128)
    one("L2 main.0 512->256 @64", 64, 256, 256, 256)
    one("L2 res 256->256 @64", 64, 256, 0, 256)
    one("bottleneck 256->256 @32", 32, 256, 0, 256)
    one("D2 res 32->32 @256", 256, 32, 0, 32)
