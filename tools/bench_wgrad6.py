#!/usr/bin/env python3
"""Winograd weight gradient: fp32 MFMA tile (algo 1) vs six bf16 products per fp32 product (algo 3) at the config-2 shapes
(B=8, 8 grouped time steps as in the train step): time per launch and the largest difference between the two results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit, B

G = int(os.environ.get("GROUPS", 8))


def one(name, H, Ca, Cb, Co):
    Ci = Ca + Cb
    steps = []
    for t in range(G):
        a = torch.randn(B, H, H, Ca, device="cuda")
        b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
        g = torch.randn(B, H, H, Co, device="cuda") * 0.01
        steps.append((g, a, b))
    fl = 2.0 * G * B * H * H * Co * Ci * 9
    res = {}
    for algo in (1, 3):
        dw = torch.zeros(Co, Ci, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
        g0, a0, b0 = steps[0]

        def run():
            return ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=algo, phase=1, more=steps[1:])
        slabs = run()
        t = timeit(run)
        dw.zero_(); db.zero_()
        slabs = ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=algo, phase=1, more=steps[1:])
        ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=algo, phase=3, slabs=slabs)
        res[algo] = (t, dw.clone(), db.clone())
    d = (res[1][1] - res[3][1]).abs().max().item() / res[1][1].abs().max().item()
    dbd = (res[1][2] - res[3][2]).abs().max().item() / res[1][2].abs().max().item()
    t1, t3 = res[1][0], res[3][0]
    print(f"{name:26s} fp32 {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF(eff) | x6 {t3*1e6:8.1f} us {fl/t3/1e12:6.1f} TF(eff) x{t1/t3:4.2f} | "
          f"rel diff dw {d:.1e} db {dbd:.1e}", flush=True)


if __name__ == "__main__":
    one("L0 res 64->64 @256", 256, 64, 0, 64)
    one("L0 main.0 128->64 @256", 256, 64, 64, 64)
    one("L1 res 128->128 @128", 128, 128, 0, 128)
    one("L1 main.0 256->128 @128", 128, 128, 128, 128)
    one("L2 res 256->256 @64", 64, 256, 0, 256)
    one("L2 main.0 512->256 @64", 64, 256, 256, 256)
    one("L0 first 32->64 @256", 256, 32, 0, 64)
    one("bottleneck 256->256 @32", 32, 256, 0, 256)
