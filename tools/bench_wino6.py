#!/usr/bin/env python3
"""fp32 Winograd tile (algo 1) vs Winograd x six bf16 products (algo 5) vs Winograd x three fp16 products (algo 5, terms 3) at
the config-2 shapes (B=8): time and -- on a small crop -- the largest deviation of each from the float64 convolution."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from refid_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit, B


def one(name, H, Ca, Cb, Co, res=False, mask=False):
    Ci = Ca + Cb
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (1.0 / (Ci * 9) ** 0.5)
    bias = torch.randn(Co, device="cuda")
    r = torch.randn(B, H, H, Co, device="cuda") if res else None
    m = torch.randn(B, H, H, Co, device="cuda") if mask else None
    fl = 2.0 * B * H * H * Co * Ci * 9
    kw = dict(kh=3, kw=3, pad=1, cout=Co, cout_pad=-(-Co // 64) * 64, in_b=b, bias=bias, slope_pre=0.1, res=r, mask=m,
              slope_mask=0.2 if mask else 1.0)
    w1 = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    w6 = ops.pack_conv_weights_wino6(w, ops.ROLE_WINO_FWD, Co, Ci)
    w3 = ops.pack_conv_weights_wino6(w, ops.ROLE_WINO_FWD, Co, Ci, f16=True)
    o1 = torch.empty(B, H, H, Co, device="cuda")
    o6 = torch.empty(B, H, H, Co, device="cuda")
    o3 = torch.empty(B, H, H, Co, device="cuda")
    t1 = timeit(lambda: ops.conv2d(a, w1, o1, algo=1, **kw))
    t6 = timeit(lambda: ops.conv2d(a, w6, o6, algo=5, **kw))
    t3 = timeit(lambda: ops.conv2d(a, w3, o3, algo=5, terms=3, **kw))
    # float64 reference on a crop of sample 0 (rows 0..15: includes the top border)
    crop = 18
    xa = torch.cat([a[:1, :crop], b[:1, :crop]], 3) if b is not None else a[:1, :crop]
    ref = F.conv2d(xa.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), bias.double().cpu(), 1, 1)
    ref = torch.where(ref > 0, ref, 0.1 * ref)[:, :, :crop - 2].permute(0, 2, 3, 1)
    if res:
        ref = ref + r[:1, :crop - 2].double().cpu()
    if mask:
        ref = ref * torch.where(m[:1, :crop - 2].cpu() > 0, 1.0, 0.2)
    e1, e6, e3 = ((o[:1, :crop - 2].double().cpu() - ref).abs().max().item() for o in (o1, o6, o3))
    print(f"{name:28s} wino fp32 {t1*1e6:7.1f} us | x6 bf16 {t6*1e6:7.1f} us | x3 fp16 {t3*1e6:7.1f} us {fl/t3/1e12:6.1f} TF(eff) "
          f"x{t6/t3:4.2f} vs x6 | err vs fp64: fp32 {e1:.1e}  x6 {e6:.1e}  x3 {e3:.1e}", flush=True)


if __name__ == "__main__":
    one("L0 first dgrad 64->32.. skip", 256, 64, 0, 64)
    one("L0 main.0 128->64 @256", 256, 64, 64, 64)
    one("L0 res 64->64 @256", 256, 64, 0, 64)
    one("L0 res 64->64 @256 +res", 256, 64, 0, 64, res=True)
    one("L0 res dgrad +res +mask", 256, 64, 0, 64, res=True, mask=True)
    one("L0 first 32->64 @256", 256, 32, 0, 64)
    one("L1 main.0 256->128 @128", 128, 128, 128, 128)
    one("L1 res 128->128 @128", 128, 128, 0, 128)
    one("L2 main.0 512->256 @64", 64, 256, 256, 256)
    one("L2 res 256->256 @64", 64, 256, 0, 256)
    one("L2 first 128->256 @64", 64, 128, 0, 256)
    one("bottleneck 256->256 @32", 32, 256, 0, 256)
    one("D1 main.0 128->64 @128", 128, 64, 64, 64)
    # 32 output channels: the tile's one-column-tile form (round 4)
    one("D2 main.0 64->32 @256", 256, 32, 32, 32)
    one("D2 res 32->32 @256", 256, 32, 0, 32)
    one("D2 res 32->32 @256 +res", 256, 32, 0, 32, res=True)
    one("D2 res dgrad +res +mask", 256, 32, 0, 32, res=True, mask=True)
    one("L0 first dgrad 64->32 @256", 256, 64, 0, 32)
