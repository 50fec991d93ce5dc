#!/usr/bin/env python3
"""Where does a Winograd tile's time go?  Builds refid_amd/csrc/conv_wino.hip with -DREFID_WINO_TRACE (every workgroup
stamps the 100 MHz wall clock at its phase boundaries and records its CU), runs a few config-2 shapes and prints, per
shape: mean duration of prologue / K loop / exchange / epilogue, the gap between consecutive workgroups on the same CU
slot, and how much of a CU's busy time has 0 / 1 / 2 workgroups inside their K loop (the MFMA section).

  python tools/probes/wino_trace.py --build      (CPU container: cross-compile the traced library)
  python tools/probes/wino_trace.py              (GPU box)
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "probes", "bin")
LIB = os.path.join(BIN, "librefid_trace.so")
KERNEL = os.environ.get("WINO_SRC", os.path.join(ROOT, "refid_amd", "csrc", "conv_wino.hip"))


def build():
    from refid_amd.build import FLAGS, HIPCC, build as build_main
    build_main()
    os.makedirs(BIN, exist_ok=True)
    traced = []
    for src in ("conv_wino", "experimental/conv_wino2"):
        obj = os.path.join(BIN, os.path.basename(src) + "_trace.o")
        subprocess.check_call([HIPCC] + FLAGS + ["-DREFID_WINO_TRACE", "-DREFID_EXPERIMENTAL_TILES", "-I",
                                                 os.path.join(ROOT, "refid_amd", "csrc"),
                                                 "-c", os.path.join(ROOT, "refid_amd", "csrc", src + ".hip"), "-o", obj])
        traced.append(obj)
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "refid_amd", "csrc", "*.o")))
            if os.path.basename(o) not in ("conv_wino.o", "conv_wino2.o")]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + traced + objs)
    print("built", LIB)


def main():
    import numpy as np
    import torch
    from refid_amd import _lib
    _lib.LIB_PATH = LIB
    from refid_amd import ops
    ops.WINO_TILE = 1                       # the traced kernel is the 2-waves tile
    L = _lib.lib()
    L.refid_wino_trace_set.argtypes = [C.c_void_p]
    B = int(os.environ.get("B", 8))
    shapes = [("L0 res 64->64 @256", 256, 64, 0, 64), ("L0 main.0 128->64 @256", 256, 64, 64, 64),
              ("L1 res 128->128 @128", 128, 128, 0, 128), ("L2 res 256->256 @64", 64, 256, 0, 256),
              ("D2 res 32->32 @256", 256, 32, 0, 32)]
    for name, H, Ca, Cb, Co in shapes:
        Ci = Ca + Cb
        a = torch.randn(B, H, H, Ca, device="cuda")
        b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        res = torch.randn(B, H, H, Co, device="cuda")
        out = torch.empty(B, H, H, Co, device="cuda")
        bias = torch.randn(Co, device="cuda")
        ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
        cw = -(-Co // 64) * 64
        run = lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=cw, in_b=b, bias=bias, res=res,  # noqa: E731
                                 slope_pre=0.1, algo=1)
        for _ in range(3):
            run()
        nwg = 1 << 16
        buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
        L.refid_wino_trace_set(C.c_void_p(buf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        L.refid_wino_trace_set(None)
        t = buf.view(nwg, 8).cpu().numpy()
        t = t[t[:, 4] != 0]
        us = lambda x: x * 0.01                                 # 100 MHz ticks -> us  # noqa: E731
        d = [us((t[:, i + 1] - t[:, i]).astype(np.float64)) for i in range(4)]
        tot = us((t[:, 4] - t[:, 0]).astype(np.float64))
        hw = t[:, 7]
        cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        print(f"\n{name}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups on {len(np.unique(cu))} CUs")
        print("  per workgroup (us, mean / p10 / p90): " + "  ".join(
            f"{n} {x.mean():.2f}/{np.percentile(x, 10):.2f}/{np.percentile(x, 90):.2f}"
            for n, x in zip(("prologue", "k-loop", "exchange", "epilogue", "total"), d + [tot])))
        # per CU: time with k WGs inside their K loop, and start gaps
        t0 = t[:, 0].min()
        in_loop = np.zeros(3); busy = 0.0; gaps = []
        for c in np.unique(cu):
            rows = t[cu == c]
            ev = []
            for r in rows:
                ev += [(r[1], 1, 0), (r[2], -1, 0), (r[0], 0, 1), (r[4], 0, -1)]
            ev.sort()
            k = res_n = 0; last = None
            for when, dk, dr in ev:
                if last is not None and res_n > 0:
                    in_loop[min(k, 2)] += when - last
                    busy += when - last
                k += dk; res_n += dr; last = when
            ends = np.sort(rows[:, 4]); starts = np.sort(rows[:, 0])
            # the i-th start after the first two (both slots filled at launch) follows the (i-2)-th end
            for i in range(2, len(starts)):
                gaps.append(us(float(starts[i] - ends[i - 2])))
        print(f"  CU busy time split by #workgroups in their K loop: 0: {in_loop[0] / busy:.2f}  1: {in_loop[1] / busy:.2f}  "
              f"2: {in_loop[2] / busy:.2f};  end -> next start on the CU: {np.mean(gaps):.2f} us (p90 {np.percentile(gaps, 90):.2f})")
        print(f"  kernel span by stamps: {us(float(t[:, 4].max() - t0)):.1f} us")
        # fine-grained: K-loop phases of one mid-kernel workgroup (shader cycles)
        nchunks = Ci // 8
        kb = torch.zeros(4 * 64 * 8, dtype=torch.int64, device="cuda")
        L.refid_wino_ktrace_set.argtypes = [C.c_void_p, C.c_int]
        for wg in (len(t) // 2 + 3, len(t) // 2 + 700):
            kb.zero_()
            L.refid_wino_ktrace_set(C.c_void_p(kb.data_ptr()), wg)
            run()
            torch.cuda.synchronize()
            k = kb.view(4, 64, 8).cpu().numpy()
            L.refid_wino_ktrace_set(None, -1)
            for w in (0, 3):
                rows = k[w, :min(nchunks, 64)]
                wait = rows[:, 1] - rows[:, 0]; issue = rows[:, 2] - rows[:, 1]; mfma = rows[:, 3] - rows[:, 2]; bar = rows[:, 4] - rows[:, 3]
                gap = rows[1:, 0] - rows[:-1, 4]
                print(f"  wg {wg} wave {w}: per chunk cycles  vmcnt-wait {wait.mean():.0f}  issue(ds_write+loads) {issue.mean():.0f}  "
                      f"lds+valu+mfma {mfma.mean():.0f}  barrier {bar.mean():.0f}  between-phases {gap.mean():.0f}  | chunk total "
                      f"{(rows[-1, 4] - rows[0, 0]) / len(rows):.0f}  (32 MFMAs = 2048)")
                print("     first chunks (wait, issue, mfma, barrier): " + "  ".join(
                    f"({a},{b},{c},{d})" for a, b, c, d in zip(wait[:6], issue[:6], mfma[:6], bar[:6])))


def persistent():
    """K-loop phase accounting of the persistent one-wave-per-SIMD tile (exact: the wave is alone on its SIMD)."""
    import numpy as np
    import torch
    from refid_amd import _lib
    _lib.LIB_PATH = LIB
    from refid_amd import ops
    ops.WINO_TILE = 2
    L = _lib.lib()
    L.refid_wino2_ktrace_set.argtypes = [C.c_void_p, C.c_int]
    B = 8
    for name, H, Ca, Cb, Co in [("L0 res 64->64 @256", 256, 64, 0, 64), ("L1 res 128->128 @128", 128, 128, 0, 128),
                                ("L2 main.0 512->256 @64", 64, 256, 256, 256)]:
        Ci = Ca + Cb
        a = torch.randn(B, H, H, Ca, device="cuda")
        b = torch.randn(B, H, H, Cb, device="cuda") if Cb else None
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        res = torch.randn(B, H, H, Co, device="cuda")
        out = torch.empty(B, H, H, Co, device="cuda")
        bias = torch.randn(Co, device="cuda")
        ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
        run = lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=-(-Co // 64) * 64, in_b=b, bias=bias,  # noqa: E731
                                 res=res, slope_pre=0.1, algo=1)
        for _ in range(3):
            run()
        kb = torch.zeros(4 * 64 * 8, dtype=torch.int64, device="cuda")
        nchunks = min(Ci // 8, 64)
        print(f"\n{name} (persistent tile, second tile of the workgroup; clock64 ticks)")
        for wg in (17, 130):
            kb.zero_()
            L.refid_wino2_ktrace_set(C.c_void_p(kb.data_ptr()), wg)
            run()
            torch.cuda.synchronize()
            L.refid_wino2_ktrace_set(None, -1)
            k = kb.view(4, 64, 8).cpu().numpy()
            for wv in (0, 3):
                r = k[wv, :nchunks]
                names = ("vmcnt-wait", "ds_write+load issue", "block0 (32 MFMA + reads/V of block1)", "barrier",
                         "block1 (32 MFMA + reads/V of next chunk)")
                d = [r[:, i + 1] - r[:, i] for i in range(5)]
                gap = r[1:, 0] - r[:-1, 5]
                tot = (r[-1, 5] - r[0, 0]) / len(r)
                print(f"  wg {wg} wave {wv}: " + "  ".join(f"{n} {x[1:-1].mean():.0f}" for n, x in zip(names, d)) +
                      f"  between {gap.mean():.0f}  | chunk {tot:.0f} ticks (64 MFMAs = 4096 cycles)")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--persistent" in sys.argv:
        persistent()
    else:
        main()
