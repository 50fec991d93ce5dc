#!/usr/bin/env python3
"""What does each piece of the split tile (csrc/conv_split.hip) cost?  VERDICT r5 #3 asked for the ablation of the bf16 mode's
dominant kernel, conv_split_kernel<2,2,1,1,0> (one bf16 product on the direct 3x3 tile: 69 us x 1070 launches per step at 0.32 of HBM
and 0.16 of the bf16 matrix peak -- bound by neither), written like r05_wino6_ablation.txt.  Builds conv_split.hip with
-DREFID_SPLIT_ABLATE=n (one piece removed, results wrong) next to the product build and times config-2 shapes with each;
SPLIT_TERMS=1 (default: bf16 mode's form), 6, 3 or 19 with SPLIT_MODE=down for conv_down's forward.

  python tools/probes/split_ablate.py --build      (CPU container: cross-compile the variants)
  python tools/probes/split_ablate.py              (GPU box)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
VARIANTS = {0: "product", 1: "no MFMAs", 2: "no operand split", 3: "global loads cache resident (chunk 0)", 4: "no global loads in the K loop",
            5: "no K loop", 6: "no epilogue traffic", 7: "no K loop, no epilogue traffic", 8: "no LDS fragment reads in the K loop"}
if os.environ.get("SPLIT_ONLY"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["SPLIT_ONLY"].split(",")}
TERMS = int(os.environ.get("SPLIT_TERMS", "1"))
DOWN = os.environ.get("SPLIT_MODE", "") == "down"


def lib_path(v):
    return os.path.join(BIN, f"librefid_spabl{v}.so")


def build():
    from refid_amd.build import FLAGS, HIPCC, build as build_main
    build_main()
    os.makedirs(BIN, exist_ok=True)
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o"))) if os.path.basename(o) != "conv_split.o"]
    for v in VARIANTS:
        obj = os.path.join(BIN, f"conv_split_abl{v}.o")
        subprocess.check_call([HIPCC] + FLAGS + [f"-DREFID_SPLIT_ABLATE={v}", "-I", os.path.join(ROOT, "refid_amd", "csrc"),
                                                 "-c", os.path.join(ROOT, "refid_amd", "csrc", "conv_split.hip"), "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(v), obj] + objs)
    print("built", len(VARIANTS), "variants")


def run(v):
    import torch
    from refid_amd import _lib
    if not os.path.exists(lib_path(v)):
        raise SystemExit(f"{lib_path(v)} is missing: python tools/probes/split_ablate.py --build in the CPU container first")
    _lib.LIB_PATH = lib_path(v)
    from refid_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_kernels import timeit, B
    if DOWN:
        shapes = [("down 64 @256", 256, 64, 64, 0), ("down 128 @128", 128, 128, 128, 0), ("down 256 @64", 64, 256, 256, 0)]
    else:
        shapes = [("64->64 @256", 256, 64, 64, 0), ("64->64 @256 +res+mask", 256, 64, 64, 1), ("32->64 @256", 256, 32, 64, 0),
                  ("128->128 @128", 128, 128, 128, 0), ("256->256 @64", 64, 256, 256, 0)]
    planes = {1: 1, 3: 2, 6: 3, 19: 2}[TERMS]
    row = []
    for name, H, Ci, Co, rm in shapes:
        a = torch.randn(B, H, H, Ci, device="cuda")
        w = torch.randn(Co, Ci, 4 if DOWN else 3, 4 if DOWN else 3, device="cuda") * 0.05
        k = 4 if DOWN else 3
        bn = ops.conv_bn(k, k, 2 if DOWN else 1, 0, Co)
        wp = ops.pack_conv_weights_split(w, ops.ROLE_FWD, bn, k, k, Co, Ci, planes=planes, f16=TERMS == 19)
        Ho = H // 2 if DOWN else H
        out = torch.empty(B, Ho, Ho, Co, device="cuda")
        r = torch.randn(B, Ho, Ho, Co, device="cuda") if rm else None
        m = torch.randn(B, Ho, Ho, Co, device="cuda") if rm else None
        bias = torch.randn(Co, device="cuda")
        t = timeit(lambda: ops.conv2d(a, wp, out, kh=k, kw=k, stride=2 if DOWN else 1, pad=1, cout=Co, cout_pad=-(-Co // bn) * bn,
                                      bias=bias, res=r, mask=m, slope_mask=0.2, slope_pre=0.1, algo=4, terms=TERMS))
        row.append(f"{name} {t * 1e6:7.1f}")
    print(f"[{v} terms {TERMS}{' down' if DOWN else ''}] {VARIANTS[v]:38s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:   # one process per variant: each loads its own library
        for v in VARIANTS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(v)])
