#!/usr/bin/env python3
"""What does each piece of the 2x4-tile Winograd weight-gradient kernel cost?  Builds refid_amd/csrc/wgrad_wino24.hip with
-DREFID_W24_ABLATE=n (one piece removed, results wrong) next to the product build and times config-2 shapes with each
(8 grouped time steps per launch as in the train step).

  python tools/probes/w24_ablate.py --build      (CPU container: cross-compile the variants)
  python tools/probes/w24_ablate.py              (GPU box)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
VARIANTS = {0: "product tile (algo 5)", 1: "no DMA requests (stale tiles)", 2: "no MFMAs", 3: "no LDS reads", 4: "no barrier",
            5: "no DMA, no barrier", 6: "no DMA, no LDS reads, no barrier (transforms + MFMAs alone)",
            7: "every request re-fetches the first tile (cache hits)"}
if os.environ.get("W24_ONLY"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["W24_ONLY"].split(",")}


def lib_path(v):
    return os.path.join(BIN, f"librefid_w24abl{v}.so")


def build():
    from refid_amd.build import FLAGS, HIPCC, build as build_main
    build_main()
    os.makedirs(BIN, exist_ok=True)
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o"))) if os.path.basename(o) != "wgrad_wino24.o"]
    for v in VARIANTS:
        obj = os.path.join(BIN, f"wgrad_wino24_abl{v}.o")
        subprocess.check_call([HIPCC] + FLAGS + [f"-DREFID_W24_ABLATE={v}", "-I", os.path.join(ROOT, "refid_amd", "csrc"),
                                                 "-c", os.path.join(ROOT, "refid_amd", "csrc", "wgrad_wino24.hip"), "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(v), obj] + objs)
    print("built", len(VARIANTS), "variants")


def run(v):
    import torch
    from refid_amd import _lib
    if not os.path.exists(lib_path(v)):
        raise SystemExit(f"{lib_path(v)} is missing: python tools/probes/w24_ablate.py --build (CPU container)")
    _lib.LIB_PATH = lib_path(v)
    from refid_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_kernels import timeit, B
    G = 8
    shapes = [("64->64 @256", 256, 64, 0, 64), ("128->128 @128", 128, 128, 0, 128), ("512->256 @64", 64, 256, 256, 256)]
    row = []
    for name, H, Ca, Cb, Co in shapes:
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
            return ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=5, phase=1, more=steps[1:])
        go()
        t = timeit(go)
        fl = 2.0 * G * B * H * H * Co * Ci * 24 / 8           # issued: 24 multiplies per 2x4 outputs
        row.append(f"{name} {t*1e6:7.1f} us {fl/t/1e12:5.1f} TF")
    print(f"[{v}] {VARIANTS[v]:60s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--one" in sys.argv:
        run(int(sys.argv[sys.argv.index("--one") + 1]))
    else:
        for v in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", str(v)])
