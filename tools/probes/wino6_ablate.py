#!/usr/bin/env python3
"""What does each piece of the Winograd x six-product tile cost?  Builds refid_amd/csrc/conv_wino6.hip with
-DREFID_WINO6_ABLATE=n (n = 1..5: one piece removed, results wrong) next to the product build and times a few config-2
shapes with each.

  python tools/probes/wino6_ablate.py --build      (CPU container: cross-compile the variants)
  python tools/probes/wino6_ablate.py              (GPU box)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
VARIANTS = {0: "product", 1: "U fragments cache resident", 2: "no three-plane split", 3: "no K loop",
            4: "no res / mask loads, no stores", 5: "raw halo cache resident", 6: "no U loads in the K loop",
            7: "no raw loads / LDS stores in the K loop", 8: "neither (6 + 7)", 9: "odd-slot workgroup starts 4 us late",
            10: "no K loop, no epilogue traffic (3 + 4)", 11: "odd-slot workgroup starts 8 us late",
            12: "U loads: same 16 bytes for every lane", 13: "half the U loads",
            14: "three products, two U planes", 15: "three products, two U planes, no split",
            16: "fp16 form: no max / rescale / scale", 17: "fp16 form: one-instruction max", 18: "fp16 form: no split",
            19: "prologue halo loads cache resident", 20: "workgroups return at once (dispatch only)", 21: "20 + one barrier", 3: "no K loop", 10: "no K loop, no epilogue traffic (3 + 4)"}
TERMS = int(os.environ.get("WINO6_TERMS", "0"))          # 3: time the three-fp16-product form (variants 0, 16-18)
if os.environ.get("WINO6_ONLY"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["WINO6_ONLY"].split(",")}
EXTRA = os.environ.get("WINO6_FLAGS", "").split()


TAG = os.environ.get("WINO6_TAG", "")


def lib_path(v):
    return os.path.join(BIN, f"librefid_w6abl{v}{TAG}.so")


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
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o"))) if os.path.basename(o) != "conv_wino6.o"]
    for v in VARIANTS:
        obj = os.path.join(BIN, f"conv_wino6_abl{v}{TAG}.o")
        subprocess.check_call([HIPCC] + FLAGS + EXTRA + [f"-DREFID_WINO6_ABLATE={v}", "-I", os.path.join(ROOT, "refid_amd", "csrc"),
                                                         "-c", os.path.join(ROOT, "refid_amd", "csrc", "conv_wino6.hip"), "-o", obj])
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
    shapes = [("64->64 @256", 256, 64, 0, 64, 0), ("64->64 @256 +res+mask", 256, 64, 0, 64, 1), ("128->64 @256", 256, 64, 64, 64, 0),
              ("128->128 @128", 128, 128, 0, 128, 0), ("512->256 @64", 64, 256, 256, 256, 0)]
    row = []
    for name, H, Ca, Cb, Co, rm in shapes:
        Ci = Ca + Cb
        a = torch.randn(B, H, H, Ca, device="cuda")
        b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        out = torch.empty(B, H, H, Co, device="cuda")
        r = torch.randn(B, H, H, Co, device="cuda") if rm else None
        m = torch.randn(B, H, H, Co, device="cuda") if rm else None
        bias = torch.randn(Co, device="cuda")
        w6 = ops.pack_conv_weights_wino6(w, ops.ROLE_WINO_FWD, Co, Ci, f16=TERMS == 3)
        t = timeit(lambda: ops.conv2d(a, w6, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=-(-Co // 64) * 64, in_b=b, bias=bias,
                                      res=r, mask=m, slope_mask=0.2, slope_pre=0.1, algo=5, terms=TERMS))
        row.append(f"{name} {t * 1e6:7.1f}")
    print(f"[{v}{TAG}{' fp16x3' if TERMS == 3 else ''}] {VARIANTS[v]:32s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:   # one process per variant: each loads its own library
        for v in VARIANTS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(v)])
