#!/usr/bin/env python3
"""Does an initialised RCCL process group change a pinned host -> device copy?  (bench.py's prefetching step was 18 ms slower
with a 1-rank group.)  Times a 300 MB pinned copy on a side stream: host time of the call, device time of the copy."""
import os
import sys
import time

import torch

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
h = torch.empty(75 * 1024 * 1024, dtype=torch.float32).pin_memory()
side = torch.cuda.Stream()


def probe(tag):
    for k in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record()
            d = h.to(dev, non_blocking=True)
            e1.record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        if k:
            print(f"{tag:28s} host call {host * 1e3:7.3f} ms   device copy {e0.elapsed_time(e1):7.3f} ms  ({h.numel() * 4 / e0.elapsed_time(e1) / 1e6:.1f} GB/s)", flush=True)


probe("before init_process_group")
import torch.distributed as dist
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
probe("after init (device_id)")
t = torch.ones(1024, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
probe("after the first all_reduce")
dist.destroy_process_group()
