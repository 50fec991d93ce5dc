#!/usr/bin/env python3
"""Where does a split-bf16 conv tile's time go?  Builds refid_amd/csrc/conv_split.hip with -DREFID_SPLIT_TRACE (every
workgroup stamps the 100 MHz wall clock at its phase boundaries and records its CU; one workgroup stamps every K-loop
phase of waves 0 and 3) and prints the phase durations per shape and product count.

  python tools/probes/split_trace.py --build      (CPU container: cross-compile the traced library)
  python tools/probes/split_trace.py              (GPU box)
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
LIB = os.path.join(BIN, "librefid_split_trace.so")


def build():
    from refid_amd.build import FLAGS, HIPCC, build as build_main
    build_main()
    os.makedirs(BIN, exist_ok=True)
    obj = os.path.join(BIN, "conv_split_trace.o")
    subprocess.check_call([HIPCC] + FLAGS + ["-DREFID_SPLIT_TRACE", "-I", os.path.join(ROOT, "refid_amd", "csrc"),
                                             "-c", os.path.join(ROOT, "refid_amd", "csrc", "conv_split.hip"), "-o", obj])
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o"))) if os.path.basename(o) != "conv_split.o"]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + objs)
    print("built", LIB)


def main():
    import numpy as np
    import torch
    from refid_amd import _lib
    _lib.LIB_PATH = LIB
    from refid_amd import ops
    L = _lib.lib()
    L.refid_split_trace_set.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    B = int(os.environ.get("B", 8))
    shapes = [("L0 res 64->64 @256", 256, 64, 0, 64), ("L1 res 128->128 @128", 128, 128, 0, 128),
              ("L2 main.0 512->256 @64", 64, 256, 256, 256)]
    for name, H, Ca, Cb, Co in shapes:
        Ci = Ca + Cb
        a = torch.randn(B, H, H, Ca, device="cuda")
        b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        res = torch.randn(B, H, H, Co, device="cuda")
        out = torch.empty(B, H, H, Co, device="cuda")
        bias = torch.randn(Co, device="cuda")
        bn = ops.conv_bn(3, 3, 1, 0, Co)
        for terms in (6, 3, 1):
            ww = ops.pack_conv_weights_split(w, ops.ROLE_FWD, bn, 3, 3, Co, Ci, planes={6: 3, 3: 2, 1: 1}[terms])
            run = lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=-(-Co // bn) * bn, in_b=b, bias=bias,  # noqa: E731
                                     res=res, slope_pre=0.1, algo=4, terms=terms)
            for _ in range(3):
                run()
            nwg = 1 << 15
            buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
            kb = torch.zeros(4 * 64 * 8, dtype=torch.int64, device="cuda")
            L.refid_split_trace_set(C.c_void_p(buf.data_ptr()), None, -1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            t = buf.view(nwg, 8).cpu().numpy()
            t = t[t[:, 3] != 0]
            us = lambda x: x * 0.01                                 # noqa: E731  100 MHz ticks -> us
            d = [us((t[:, i + 1] - t[:, i]).astype(np.float64)) for i in range(3)]
            tot = us((t[:, 3] - t[:, 0]).astype(np.float64))
            hw = t[:, 7]
            cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
            print(f"\n{name} x{terms}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups on {len(np.unique(cu))} CUs")
            print("  per workgroup (us, mean / p10 / p90): " + "  ".join(
                f"{n} {x.mean():.2f}/{np.percentile(x, 10):.2f}/{np.percentile(x, 90):.2f}"
                for n, x in zip(("prologue", "k-loop", "epilogue", "total"), d + [tot])))
            in_loop = np.zeros(4); busy = 0.0
            for c in np.unique(cu):
                rows = t[cu == c]
                ev = []
                for r in rows:
                    ev += [(r[1], 1, 0), (r[2], -1, 0), (r[0], 0, 1), (r[3], 0, -1)]
                ev.sort()
                k = rn = 0; last = None
                for when, dk, dr in ev:
                    if last is not None and rn > 0:
                        in_loop[min(k, 3)] += when - last
                        busy += when - last
                    k += dk; rn += dr; last = when
            print("  CU busy time split by #workgroups in their K loop: " + "  ".join(f"{i}: {in_loop[i] / busy:.2f}" for i in range(4)) +
                  f";  kernel span by stamps {us(float(t[:, 3].max() - t[:, 0].min())):.1f} us")
            nchunks = min(Ci // 8, 64)
            for wg in (len(t) // 2 + 3,):
                kb.zero_()
                L.refid_split_trace_set(None, C.c_void_p(kb.data_ptr()), wg)
                run()
                torch.cuda.synchronize()
                k = kb.view(4, 64, 8).cpu().numpy()
                for wv in (0, 3):
                    r = k[wv, :nchunks].astype(np.float64) * 0.01
                    names = ("load issue", "reads+mfma", "barrier1", "vmcnt wait", "split+store", "barrier2")
                    dd = [r[:, i + 1] - r[:, i] for i in range(6)]
                    print(f"  wg {wg} wave {wv} per chunk (us): " + "  ".join(f"{n} {x[:-1].mean():.2f}" for n, x in zip(names, dd)) +
                          f"  | chunk {(r[-1, 6] - r[0, 0]) / len(r):.2f}")
            L.refid_split_trace_set(None, None, -1)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
