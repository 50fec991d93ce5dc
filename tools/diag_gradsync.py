#!/usr/bin/env python3
"""Where does a 1-rank RCCL group cost step time?  (REFID_FORCE_GRADSYNC=1 python bench.py was 4 % slower than the plain step
although RCCL launches no kernel at one rank.)  Times the B=8 train step with pieces of the distributed path stubbed out."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["REFID_FORCE_GRADSYNC"] = "1"
import torch
import torch.distributed as dist
import bench

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
args = bench.parse_args(["--no-cpu-baseline", "--no-roofline"])
from refid_amd.train import TwoImageEventRecurrentRestorationModel
model = TwoImageEventRecurrentRestorationModel(bench.options(args))
x, ev, gt = bench.synthetic_batch(args.batch, args.T, args.size, args.size, args.img_chn, 100, dev)
model.feed_data({"lq": x, "voxel": ev, "gt": gt})
real_ar = dist.all_reduce
gs = model.grad_sync


def run(tag, steps=4):
    it = run.it
    for _ in range(2):
        it += 1; model.update_learning_rate(it); model.optimize_parameters(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        it += 1; model.update_learning_rate(it); model.optimize_parameters(it)
    torch.cuda.synchronize()
    run.it = it
    print(f"{tag:50s} {(time.perf_counter() - t0) / steps * 1e3:8.2f} ms/step", flush=True)


run.it = 0
run("full dist path (early + late + loss all-reduce)")
dist.all_reduce = lambda t, *a, **k: real_ar(t, *a, **k) if t.numel() > 1 else None
run("no loss all-reduce")
dist.all_reduce = real_ar
model.grad_sync = None
run("no gradient all-reduces (loss all-reduce only)")
model.grad_sync = lambda phase: gs(phase) if phase == "late" else None
run("late phase only (the early slices are skipped)")
model.grad_sync = None; model.dist_on = False
run("dist off (group still initialised)")
model.dist_on = True; model.grad_sync = gs
run("full dist path again")
