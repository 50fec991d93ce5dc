#!/usr/bin/env python3
"""(Needs a library built with REFID_EXPERIMENTAL_TILES=1 python -m refid_amd.build.)
2-waves-per-SIMD Winograd tile vs the persistent one-wave-per-SIMD tile at the config-2 shapes (B=8)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd import _lib
if os.environ.get('WINO_LIB'):
    _lib.LIB_PATH = os.environ['WINO_LIB']
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
    a = torch.randn(B, H, H, Ca, device="cuda")
    b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    r = torch.randn(B, H, H, Co, device="cuda") if res else None
    m = torch.randn(B, H, H, Co, device="cuda") if mask else None
    out = torch.empty(B, H, H, Co, device="cuda")
    bias = torch.randn(Co, device="cuda")
    fl = 2.0 * B * H * H * Co * Ci * 9
    ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    cw = -(-Co // 64) * 64
    ts = []
    for tile in (1, 2):
        ops.WINO_TILE = tile
        ts.append(timeit(lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=cw, in_b=b, bias=bias, res=r,
                                            mask=m, slope_pre=0.1, algo=1)))
    ops.WINO_TILE = 0
    f = lambda t: fl * 16 / 36 / t / 1e12 / 157.3      # noqa: E731   issued-FLOP fraction of the fp32 MFMA peak
    print(f"{name:28s} 2-wave {ts[0]*1e6:8.1f} us ({f(ts[0]):.3f})   persistent {ts[1]*1e6:8.1f} us ({f(ts[1]):.3f})   "
          f"x{ts[0]/ts[1]:.3f}")


if __name__ == "__main__":
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
