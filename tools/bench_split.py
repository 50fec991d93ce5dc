#!/usr/bin/env python3
"""fp32 Winograd tile (algo 1) vs the split-bf16 direct tile (algo 4, 6 and 3 products) at the config-2 shapes:
time per launch and the largest deviation from the fp64 convolution of the same inputs (B=8 unless env B)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from refid_amd import ops

B = int(os.environ.get("B", 8))


def timeit(fn, n=20):
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


def one(name, H, Ca, Cb, Co, res=True, mask=False):
    Ci = Ca + Cb
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(B, H, H, Ca, device="cuda", generator=g)
    b = torch.randn(B, H, H, Cb, device="cuda", generator=g) if Cb else None
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * Ci)) ** 0.5
    r = torch.randn(B, H, H, Co, device="cuda", generator=g) if res else None
    m = torch.randn(B, H, H, Co, device="cuda", generator=g) if mask else None
    bias = torch.randn(Co, device="cuda", generator=g)
    if os.environ.get("ZERO"):          # power experiment: same instruction stream, no data toggling
        for t in (a, b, w, r, m):
            if t is not None:
                t.zero_()
    # fp64 truth on the first sample (the whole batch in fp64 is slow)
    x0 = (a[:1] if b is None else torch.cat([a[:1], b[:1]], 3)).permute(0, 3, 1, 2).double()
    y = F.conv2d(x0, w.double(), bias.double(), padding=1)
    y = torch.where(y > 0, y, 0.1 * y)
    if res:
        y = y + r[:1].permute(0, 3, 1, 2).double()
    if mask:
        y = y * torch.where(m[:1].permute(0, 3, 1, 2) > 0, 1.0, 0.0).double()
    y = y.permute(0, 2, 3, 1)
    cw = -(-Co // 64) * 64
    bn = ops.conv_bn(3, 3, 1, 0, Co)
    cws = -(-Co // bn) * bn
    fl = 2.0 * B * H * H * Co * Ci * 9
    cols = []
    for label, algo, terms in (("wino-fp32", 1, 0), ("split x6", 4, 6), ("split x3", 4, 3), ("split x1", 4, 1), ("igemm-bf16", 2, 1)):
        out = torch.zeros(B, H, H, Co, device="cuda")
        if algo == 1:
            ww, pad = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci), cw
        elif algo == 2:
            ww, pad = ops.pack_conv_weights_bf16(w, ops.ROLE_FWD, bn, 16, 3, 3, Co, Ci), cws
        else:
            ww, pad = ops.pack_conv_weights_split(w, ops.ROLE_FWD, bn, 3, 3, Co, Ci, planes={6: 3, 3: 2, 1: 1}[terms]), cws
        run = lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=pad, in_b=b, bias=bias, res=r, mask=m,  # noqa: E731
                                 slope_pre=0.1, slope_mask=0.0, algo=algo, terms=terms)
        t = timeit(run)
        err = (out[:1].double() - y).abs().max().item()
        cols.append(f"{label} {t*1e6:6.1f} us err {err:.0e}")
        if algo != 1:
            cols[-1] += f" ({fl * terms / t / 1e12 / 2500:.2f})"
        else:
            cols[-1] += f" ({fl * 16 / 36 / t / 1e12 / 157.3:.2f})"
    print(f"{name:26s} " + " | ".join(cols), flush=True)


def down(name, H, C, dgrad):
    """conv_down (4x4 / stride 2) forward or input gradient: direct fp32 tile (algo 0) vs the split tile."""
    g = torch.Generator(device="cuda").manual_seed(2)
    w = torch.randn(C, C, 4, 4, device="cuda", generator=g) * (2.0 / (16 * C)) ** 0.5
    fl = 2.0 * B * (H // 2) ** 2 * C * C * 16
    if not dgrad:
        x = torch.randn(B, H, H, C, device="cuda", generator=g)
        out = torch.empty(B, H // 2, H // 2, C, device="cuda")
        ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1)
        role, mode = ops.ROLE_FWD, 0
    else:
        x = torch.randn(B, H // 2, H // 2, C, device="cuda", generator=g)
        out = torch.empty(B, H, H, C, device="cuda")
        ref = F.conv_transpose2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1)
        role, mode = ops.ROLE_DOWN_DGRAD, 2
    bn = ops.conv_bn(4, 4, 2, mode, C)
    pad = -(-C // bn) * bn
    cols = []
    for label, algo, terms in (("direct-fp32", 0, 0), ("split x6", 4, 6), ("split x3", 4, 3), ("split x1", 4, 1), ("igemm-bf16", 2, 1)):
        kc = ops.conv_kc(4, 4, 2, mode)
        if algo == 0:
            ww = ops.pack_conv_weights(w, role, bn, kc, 4, 4, C, C)
        elif algo == 2:
            ww = ops.pack_conv_weights_bf16(w, role, bn, 2 * kc, 4, 4, C, C)
        else:
            ww = ops.pack_conv_weights_split(w, role, bn, 4, 4, C, C, planes={6: 3, 3: 2, 1: 1}[terms])
        run = lambda: ops.conv2d(x, ww, out, kh=4, kw=4, stride=2, pad=1, mode=mode, cout=C, cout_pad=pad, algo=algo, terms=terms)  # noqa: E731
        t = timeit(run)
        err = (out[:1].double() - ref).abs().max().item()
        peak = 157.3 if algo == 0 else 2500.0
        cols.append(f"{label} {t*1e6:6.1f} us err {err:.0e} ({fl * max(terms, 1) / t / 1e12 / peak:.2f})")
    print(f"{name:26s} " + " | ".join(cols), flush=True)


if __name__ == "__main__":
    if os.environ.get("DOWN"):
        for H, C in ((256, 64), (128, 128), (64, 256)):
            down(f"down fwd {C}ch @{H}", H, C, False)
            down(f"down dgrad {C}ch @{H}", H, C, True)
        sys.exit(0)
    one("L0 first 32->64 @256", 256, 32, 0, 64, res=False)
    one("L0 main.0 128->64 @256", 256, 64, 64, 64, res=False)
    one("L0 res 64->64 @256", 256, 64, 0, 64)
    one("L0 dgrad 64->64 @256 r+m", 256, 64, 0, 64, res=True, mask=True)
    one("L1 main.0 256->128 @128", 128, 128, 128, 128, res=False)
    one("L1 res 128->128 @128", 128, 128, 0, 128)
    one("L2 main.0 512->256 @64", 64, 256, 256, 256, res=False)
    one("L2 res 256->256 @64", 64, 256, 0, 256)
    one("D2 res 32->32 @256", 256, 32, 0, 32)
    one("D2 main.0 64->32 @256", 256, 32, 32, 32, res=False)
    one("D1 res 64->64 @128", 128, 64, 0, 64)
