#!/usr/bin/env python3
"""Pointwise (1x1) tile: fp32 MFMA products against six bf16 products per fp32 product (refid_conv2d algo 3, mfma_terms 6)
at the config-2 shapes (B=8): time per launch, HBM rate on the algorithmic bytes, largest deviation from the float64 conv."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit, B

SHAPES = [("EGACA conv1 128->128 @128", 128, 128, 0, 128, 0), ("EGACA conv3 256->128 @128 +res", 128, 128, 128, 128, 1),
          ("EGACA conv4 128->256 @128", 128, 128, 0, 256, 0), ("EGACA conv5 256->256 @128 +res", 128, 256, 0, 256, 1),
          ("fuse_two_dir 128->64 @256", 256, 64, 64, 64, 0), ("fuse_two_dir 256->128 @128", 128, 128, 128, 128, 0),
          ("fuse_two_dir 512->256 @64", 64, 256, 256, 256, 0), ("identity 64->128 @128", 128, 64, 0, 128, 0),
          ("dgrad 128->128 @128 +mask", 128, 128, 0, 128, 2)]

for name, H, Ca, Cb, Co, rm in SHAPES:
    Ci = Ca + Cb
    torch.manual_seed(0)
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
    bias = torch.randn(Co, device="cuda")
    res = torch.randn(B, H, H, Co, device="cuda") if rm == 1 else None
    mask = torch.randn(B, H, H, Co, device="cuda") if rm == 2 else None
    out = torch.empty(B, H, H, Co, device="cuda")
    kw = dict(kh=1, kw=1, cout=Co, cout_pad=-(-Co // 32) * 32, algo=3, in_b=b, bias=bias, res=res, mask=mask, slope_mask=0.2)
    wp = ops.pack_conv_weights(w, ops.ROLE_FWD, 32, 8, 1, 1, Co, Ci)
    w6 = ops.pack_conv_weights_split(w, ops.ROLE_FWD, 32, 1, 1, Co, Ci, planes=3)
    # float64 reference on a crop
    x = torch.cat([a, b], 3) if Cb else a
    crop = x[:1, :8, :8].double()
    ref = torch.einsum("nhwc,oc->nhwo", crop, w[:, :, 0, 0].double()) + bias.double()
    if res is not None:
        ref = ref + res[:1, :8, :8].double()
    if mask is not None:
        ref = ref * torch.where(mask[:1, :8, :8] > 0, 1.0, 0.2)
    row = []
    for tag, wt, terms in (("fp32", wp, 0), ("x6", w6, 6)):
        ops.conv2d(a, wt, out, terms=terms, **kw)
        err = float((out[:1, :8, :8].double() - ref).abs().max())
        t = timeit(lambda: ops.conv2d(a, wt, out, terms=terms, **kw))
        nbytes = 4.0 * B * H * H * (Ci + Co * (1 + (res is not None) + (mask is not None)))
        row.append((t, nbytes / t / 1e12, err))
    print(f"{name:34s} fp32 {row[0][0]*1e6:7.1f} us {row[0][1]:5.2f} TB/s | x6 {row[1][0]*1e6:7.1f} us {row[1][1]:5.2f} TB/s  x{row[0][0]/row[1][0]:4.2f} | "
          f"err vs fp64: fp32 {row[0][2]:.1e}  x6 {row[1][2]:.1e}", flush=True)
