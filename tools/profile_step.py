#!/usr/bin/env python3
"""One instrumented train step; per (kernel, shape) time / TFLOP/s table (HIP events)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from refid_amd import ops
from refid_amd.train import TwoImageEventRecurrentRestorationModel

class A: pass
a = A(); a.img_chn = 26; a.batch = int(os.environ.get("B", 8)); a.T = int(os.environ.get("T", 23)); a.size = 256; a.dtype = os.environ.get("DTYPE", "fp32")
from refid_amd import engine
engine.OVERLAP_WGRAD = False          # kernels run alone: per-kernel times mean something
engine.PIPELINE = False               # (forward wavefront over four streams off as well)
model = TwoImageEventRecurrentRestorationModel(bench.options(a))
x, ev, gt = bench.synthetic_batch(a.batch, a.T, a.size, a.size, 26, 1, torch.device("cuda"))
model.feed_data({"lq": x, "voxel": ev, "gt": gt})
for it in (1, 2):
    model.update_learning_rate(it); model.optimize_parameters(it)
torch.cuda.synchronize()
ops.PROFILE = []
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); model.update_learning_rate(3); model.optimize_parameters(3); e1.record()
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
step_ms = e0.elapsed_time(e1)
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for name, fl, s, e, shape, _nb in prof:
    k = (name.replace("conv_igemm_kernel", "conv").replace("Cfg", ""), shape)
    v = agg[k]; v[0] += fl; v[1] += s.elapsed_time(e); v[2] += 1
if "--json" in sys.argv:          # algorithmic FLOPs / bytes per kernel family for tools/roofline_report.py
    import json
    fam = collections.defaultdict(lambda: dict(flops=0.0, bytes=0.0, ms=0.0, launches=0))
    for name, fl, s, e, shape, nb in prof:
        f = fam[name]
        f["flops"] += fl; f["bytes"] += nb; f["ms"] += s.elapsed_time(e); f["launches"] += 1
    json.dump(fam, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
tot = sum(v[1] for v in agg.values())
print(f"step {step_ms:.1f} ms, instrumented GEMM kernels {tot:.1f} ms")
byname = collections.defaultdict(lambda: [0.0, 0])
for (name, shape), v in agg.items():
    byname[name][0] += v[1]; byname[name][1] += v[2]
for name, (ms, n) in sorted(byname.items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:44s} {n:5d} launches {ms:7.1f} ms")
print(f"{'kernel':44s} {'(n,h,w,ca,cb,co,res,mask,bias)':38s} calls   ms    us/call  TFLOP/s")
for (name, shape), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:44s} {str(shape):38s} {v[2]:5d} {v[1]:7.1f} {v[1]/v[2]*1e3:8.1f} {v[0]/v[1]/1e9:8.1f}")
