#!/usr/bin/env python3
"""What does each piece of the fp32 Winograd weight-gradient tile cost?  Builds refid_amd/csrc/wgrad_wino.hip with
-DREFID_WW_ABLATE=n (one piece removed, results wrong) next to the product build and times config-2 shapes with each
(8 grouped time steps per launch as in the train step).

  python tools/probes/wgrad_wino_ablate.py --build      (CPU container: cross-compile the variants)
  python tools/probes/wgrad_wino_ablate.py              (GPU box)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
VARIANTS = {0: "product tile (algo 1)", 1: "tiles not re-staged (no ds_write pass, no 2nd barrier)", 2: "no global tile loads",
            3: "no loads, stores, barriers (NOT a bound: LDS reads hoisted)", 4: "3 + operands from registers"}
# (variants 100 / 5 / 6 / 11 / 12 priced the LDS-DMA form of this tile, refid_wgrad_desc.algo 4: removed with csrc/experimental/
#  in round 6; its numbers: profiles/r03_*)
if os.environ.get("WW_ONLY"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["WW_ONLY"].split(",")}


def lib_path(v):
    return os.path.join(BIN, f"librefid_wwabl{v}.so")


def check_fresh(path):
    """A variant library links the objects of the product build it was made from: refuse one that is older than any
    kernel source or header (a stale one fails with `undefined symbol` at best and measures yesterday's kernel at worst)."""
    if not os.path.exists(path):
        raise SystemExit(f"{path} is missing: run tools/build_probes.sh in the CPU container before gpurun")
    newest = max(os.path.getmtime(f) for pat in ("refid_amd/csrc/*.hip", "refid_amd/csrc/*.h", "include/*.h")
                 for f in glob.glob(os.path.join(ROOT, pat)))
    if os.path.getmtime(path) < newest:
        raise SystemExit(f"{os.path.basename(path)} is older than the kernel sources: run tools/build_probes.sh again")


def build():
    from refid_amd.build import FLAGS, HIPCC, build as build_main
    build_main()
    os.makedirs(BIN, exist_ok=True)
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o"))) if os.path.basename(o) != "wgrad_wino.o"]
    for v in VARIANTS:
        obj = os.path.join(BIN, f"wgrad_wino_abl{v}.o")
        subprocess.check_call([HIPCC] + FLAGS + [f"-DREFID_WW_ABLATE={v % 100}", "-I", os.path.join(ROOT, "refid_amd", "csrc"),
                                                 "-c", os.path.join(ROOT, "refid_amd", "csrc", "wgrad_wino.hip"), "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(v), obj] + objs)
    print("built", len(VARIANTS), "variants")


def run(v):
    import torch
    from refid_amd import _lib
    check_fresh(lib_path(v))
    _lib.LIB_PATH = lib_path(v)
    from refid_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_kernels import timeit, B
    G = 8
    shapes = [("64->64 @256", 256, 64, 0, 64), ("128->64 @256", 256, 64, 64, 64), ("128->128 @128", 128, 128, 0, 128),
              ("512->256 @64", 64, 256, 256, 256)]
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
            return ops.conv2d_wgrad(g0, a0, dw, kh=3, kw=3, pad=1, in_b=b0, db=db, algo=(4 if v >= 5 else 1), phase=1, more=steps[1:])
        go()
        t = timeit(go)
        fl = 2.0 * G * B * H * H * Co * Ci * 16 / 4          # issued: 16 multiplies per 2x2 outputs
        row.append(f"{name} {t*1e6:7.1f} us {fl/t/1e12:5.1f} TF")
    print(f"[{v}] {VARIANTS[v]:56s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--one" in sys.argv:
        run(int(sys.argv[sys.argv.index("--one") + 1]))
    else:
        for v in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", str(v)])
