#!/usr/bin/env python3
"""How much of a train step is host-side enqueue time?  (enqueue = time until optimize_parameters returns
without a device sync; step = with sync).  enqueue ~ step => the step is launch-bound at this batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
import argparse
opt = bench.options(argparse.Namespace(img_chn=26, dtype="fp32"))
from refid_amd.train import TwoImageEventRecurrentRestorationModel
model = TwoImageEventRecurrentRestorationModel(opt)
x, ev, gt = bench.synthetic_batch(B, 23, 256, 256, 26, 0, "cuda")
data = {"lq": x, "voxel": ev, "gt": gt}
for it in range(1, 4):
    model.feed_data(data); model.optimize_parameters(it)
torch.cuda.synchronize()
for it in range(4, 8):
    t0 = time.perf_counter()
    model.feed_data(data); model.optimize_parameters(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} enqueue {1e3*(t1-t0):7.1f} ms   step {1e3*(t2-t0):7.1f} ms")
