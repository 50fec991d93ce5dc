import sys, argparse, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from refid_amd.train import TwoImageEventRecurrentRestorationModel
m = TwoImageEventRecurrentRestorationModel(bench.options(argparse.Namespace(img_chn=26, dtype="fp32")))
x, ev, gt = bench.synthetic_batch(8, 23, 256, 256, 26, 0, "cuda")
m.feed_data({"lq": x, "voxel": ev, "gt": gt})
for it in (1, 2, 3):
    m.update_learning_rate(it); m.optimize_parameters(it)
torch.cuda.synchronize()
print("max allocated GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.max_memory_reserved() / 2**30)
