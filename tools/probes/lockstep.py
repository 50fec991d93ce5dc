import ctypes as C, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from refid_amd import _lib
_lib.LIB_PATH = '/root/repo/tools/probes/bin/librefid_trace.so'
from refid_amd import ops
ops.WINO_TILE = 1
L = _lib.lib()
L.refid_wino_trace_set.argtypes = [C.c_void_p]
B, H, Ca, Co = 8, 256, 64, 64
a = torch.randn(B, H, H, Ca, device="cuda"); w = torch.randn(Co, Ca, 3, 3, device="cuda") * 0.05
res = torch.randn(B, H, H, Co, device="cuda"); out = torch.empty(B, H, H, Co, device="cuda"); bias = torch.randn(Co, device="cuda")
ww = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ca)
run = lambda: ops.conv2d(a, ww, out, kh=3, kw=3, pad=1, cout=Co, cout_pad=64, bias=bias, res=res, slope_pre=0.1, algo=1)
for _ in range(3): run()
nwg = 1 << 16
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
L.refid_wino_trace_set(C.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); L.refid_wino_trace_set(None)
t = buf.view(nwg, 8).cpu().numpy(); t = t[t[:, 4] != 0]
t0 = t[:, 0].min()
for name, col in (("start", 0), ("k-loop end", 2), ("end", 4)):
    x = (t[:, col] - t0) * 0.01
    h, _ = np.histogram(x, bins=np.arange(0, 300, 4.0))
    print(f"{name:11s}", " ".join(f"{v:3d}" for v in h[:70]))
# concurrency of epilogue phases: number of WGs between stamp 2 and 4 over time
ev = sorted([(r[2], 1) for r in t] + [(r[4], -1) for r in t])
cur = 0; samples = []
last = ev[0][0]
acc = {}
for when, d in ev:
    acc[cur] = acc.get(cur, 0) + (when - last); cur += d; last = when
tot = sum(acc.values())
print("time share by #WGs in exchange/epilogue (of 512 resident):", {k: round(v / tot, 3) for k, v in sorted(acc.items()) if v / tot > 0.02})
import collections
b = collections.Counter()
for k, v in acc.items(): b[min(k // 64, 8)] += v
print("binned by 64:", {k * 64: round(v / tot, 3) for k, v in sorted(b.items())})
