#!/usr/bin/env python3
"""Peak HBM of a train step (torch's caching allocator: max allocated / reserved) for the BASELINE configurations:
  python tools/mem_footprint.py                    configs[1]: B=8, T=23, 256x256, fp32
  python tools/mem_footprint.py --config 3         configs[2]'s per-GPU share: B=4, T=25, bf16 -- with a 1-rank RCCL process group up
                                                   (its buffers) and, with --graph, the three-graph replay's private pool
  --batch / --T / --dtype / --graph override."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from refid_amd.train import TwoImageEventRecurrentRestorationModel

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--batch", type=int); ap.add_argument("--T", type=int); ap.add_argument("--dtype")
ap.add_argument("--graph", action="store_true"); ap.add_argument("--rccl", action="store_true")
a = ap.parse_args()
B, T, dtype = (8, 23, "fp32") if a.config == 2 else (4, 25, "bf16")
B, T, dtype = a.batch or B, a.T or T, a.dtype or dtype
if a.rccl or a.config == 3:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", REFID_FORCE_GRADSYNC="1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
m = TwoImageEventRecurrentRestorationModel(bench.options(argparse.Namespace(img_chn=26, dtype=dtype)))
if a.graph:
    m.set_graph_mode(True)
x, ev, gt = bench.synthetic_batch(B, T, 256, 256, 26, 0, "cuda")
m.feed_data({"lq": x, "voxel": ev, "gt": gt})
for it in (1, 2, 3, 4):
    m.update_learning_rate(it); m.optimize_parameters(it)
torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info()
print(f"B={B} T={T} {dtype}{' graph' if a.graph else ''}{' +1-rank RCCL group' if torch.distributed.is_initialized() else ''}: "
      f"max allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, max reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB, "
      f"device in use {(total - free) / 2**30:.1f} of {total / 2**30:.0f} GiB; weight-gradient group "
      f"{m.net_g.engine.recurrent_ops[0].w_group}")
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
