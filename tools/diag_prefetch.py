#!/usr/bin/env python3
"""Why did `bench.py --steps 20` run 6x slower with the prefetcher than with a resident batch (round 4)?  Runs 20 un-synchronised
steps in several variants of the hand-over and prints ms/step and the allocator's state."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from refid_amd.train import TwoImageEventRecurrentRestorationModel
from refid_amd.data import CUDAPrefetcher

variant = sys.argv[1] if len(sys.argv) > 1 else "prefetch"
args = bench.parse_args(["--no-cpu-baseline"])
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = TwoImageEventRecurrentRestorationModel(bench.options(args))


class HB:
    def __init__(s):
        s.b = []
        for k in range(2):
            x, ev, gt = bench.synthetic_batch(8, 23, 256, 256, 26, 100 + 1000 * k, dev)
            s.b.append({"lq": x.cpu().pin_memory(), "voxel": ev.cpu().pin_memory(), "gt": gt.cpu().pin_memory()})

    def __iter__(s):
        k = 0
        while True:
            yield dict(s.b[k % 2]); k += 1


hb = HB()
pre = CUDAPrefetcher(hb, device=dev, time_waits=True)
dev_batches = [{k: v.to(dev) for k, v in b.items()} for b in hb.b]
it = 0


def step():
    global it
    it += 1
    if variant == "prefetch":
        model.feed_data(pre.next())
    elif variant == "resident_alternate":
        model.feed_data(dev_batches[it % 2])
    elif variant == "same_stream_copy":            # H2D on the compute stream, no side stream, no record_stream
        model.feed_data({k: v.to(dev, non_blocking=True) for k, v in hb.b[it % 2].items()})
    elif variant == "prefetch_sync_every_4":
        if it % 4 == 0:
            torch.cuda.synchronize()
        model.feed_data(pre.next())
    model.update_learning_rate(it); model.optimize_parameters(it)


for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
st = torch.cuda.memory_stats()
print(f"{variant:24s} {1e3*(t2-t0)/20:8.1f} ms/step (host enqueue {1e3*(t1-t0)/20:7.1f} ms/step)  reserved {torch.cuda.memory_reserved()/2**30:6.1f} GiB  "
      f"alloc retries {st.get('num_alloc_retries', -1)}  hipMalloc calls {st.get('segment.all.allocated', -1)}  hipFree {st.get('segment.all.freed', -1)}", flush=True)
