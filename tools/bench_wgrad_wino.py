#!/usr/bin/env python3
"""fp32 Winograd weight gradient at the config-2 shapes, 8 grouped time steps per launch as in the train step: the F(2x2,3x3)
tile (algo 1, "regs") against the 2x4-tile form (algo 5, "f24").  Not bit-equal: the largest relative difference is printed."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("L0 res 64->64 @256", 256, 64, 0, 64), ("L0 main.0 128->64 @256", 256, 64, 64, 64),
          ("L0 first 32->64 @256", 256, 32, 0, 64), ("L0 dec 64->32 @256", 256, 64, 0, 32),
          ("L1 res 128->128 @128", 128, 128, 0, 128), ("L1 main.0 256->128 @128", 128, 128, 128, 128),
          ("L2 res 256->256 @64", 64, 256, 0, 256), ("L2 main.0 512->256 @64", 64, 256, 256, 256),
          ("bottleneck 256->256 @32", 32, 256, 0, 256)]


def run(tag, algo):
    import torch
    from refid_amd import ops
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_kernels import timeit, B
    G = int(os.environ.get("GROUPS", 8))
    out = {}
    only = os.environ.get("ONLY_SHAPE")
    for name, H, Ca, Cb, Co in (SHAPES if only is None else [SHAPES[int(only)]]):
        torch.manual_seed(1)
        Ci = Ca + Cb
        steps = []
        for t in range(G):
            a = torch.randn(B, H, H, Ca, device="cuda")
            b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
            g = torch.randn(B, H, H, Co, device="cuda") * 0.01
            steps.append((g, a, b))
        dw = torch.zeros(Co, Ci, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
        g0, a0, b0 = steps[0]

        def go():
            return ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=algo, phase=1, more=steps[1:])
        go()
        if "REPS" in os.environ:                               # counter passes (tools/probes/w4_pmc.sh): a few launches, no timing loop
            for _ in range(int(os.environ["REPS"])):
                go()
            torch.cuda.synchronize()
            continue
        t = timeit(go)
        dw.zero_(); db.zero_()
        slabs = go()
        ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=algo, phase=3, slabs=slabs)
        fl = 2.0 * G * B * H * H * Co * Ci * ({5: 24 / 8, 6: 36 / 16}.get(algo, 16 / 4))
        print(f"{tag} {name:26s} {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF/s issued", flush=True)
        out[name] = (dw.cpu(), db.cpu())
    torch.save(out, f"/tmp/wgrad_wino_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], {"regs": 1, "f24": 5}[sys.argv[1]])
    else:
        other = "f24"
        for tag in ("regs", other):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), tag])
        import torch
        a, b = torch.load("/tmp/wgrad_wino_regs.pt"), torch.load(f"/tmp/wgrad_wino_{other}.pt")
        for k in a:
            dwd = ((a[k][0] - b[k][0]).abs().max() / a[k][0].abs().max()).item()
            dbd = ((a[k][1] - b[k][1]).abs().max() / a[k][1].abs().max()).item()
            print(f"{k:26s} rel diff dw {dwd:.1e} db {dbd:.1e}")
