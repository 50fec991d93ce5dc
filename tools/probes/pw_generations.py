#!/usr/bin/env python3
"""Is the pointwise tile bound by its one-generation launch structure?  Times the fp32 1x1 tile (refid_conv2d algo 3) on EGACA /
fuse_two_dir shapes with 1x, 2x, 4x and 8x the pixels of the train step's launch (8 samples): if bytes per second grow with the
number of wave generations, a persistent form (several pixel tiles per wave, loads of the next under the stores of the current)
has something to gain; if they do not, the tile is bound inside a wave's lifetime.  GPU box:  python tools/probes/pw_generations.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from refid_amd import ops
from bench_kernels import timeit

SHAPES = [("conv1_e 128->128 @128", 128, 128, 0, 128, False), ("conv3 256->128 @128 +res", 128, 128, 128, 128, True),
          ("conv4 128->256 @128", 128, 128, 0, 256, False), ("fuse 64->64 @256 +res", 256, 64, 0, 64, True),
          ("fuse 256->256 @64 +res", 64, 256, 0, 256, True)]
for name, H, Ca, Cb, Co, res in SHAPES:
    row = []
    for mult in (1, 2, 4, 8):
        N = 8 * mult
        a = torch.randn(N, H, H, Ca, device="cuda")
        b = torch.randn(N, H, H, Cb, device="cuda") if Cb else None
        w = torch.randn(Co, Ca + Cb, 1, 1, device="cuda") * 0.05
        wp = ops.pack_conv_weights(w, ops.ROLE_FWD, 32, 8, 1, 1, Co, Ca + Cb)
        out = torch.empty(N, H, H, Co, device="cuda")
        r = torch.randn(N, H, H, Co, device="cuda") if res else None
        bias = torch.randn(Co, device="cuda")
        t = timeit(lambda: ops.conv2d(a, wp, out, kh=1, kw=1, stride=1, pad=0, cout=Co, cout_pad=-(-Co // 32) * 32, in_b=b, bias=bias,
                                      res=r, slope_post=0.2, algo=3))
        nbytes = 4.0 * N * H * H * (Ca + Cb + Co * (2 if res else 1))
        fl = 2.0 * N * H * H * Co * (Ca + Cb)
        row.append(f"x{mult}: {t*1e6:7.1f} us {nbytes/t/1e12:5.2f} TB/s {fl/t/1e12:6.1f} TF")
        del a, b, out, r
    print(f"{name:26s} " + " | ".join(row), flush=True)
