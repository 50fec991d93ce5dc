"""Data-parallel plumbing: one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI).

Reference: basicsr/utils/dist_util.py:11-30,66-78 (init_dist / get_dist_info; it hard-codes
``num_gpus = 4`` at :27 -- here the device is LOCAL_RANK) and base_model.py:66-72 (DDP wrap).

The HIP engine writes parameter gradients straight into one flat arena, bypassing autograd's
per-parameter hooks, so torch's DistributedDataParallel reducer would never fire.  ``GradSync``
is the replacement: a SUM all-reduce over slices of the flat gradient arena, issued
asynchronously in two phases so the first one overlaps the rest of BPTT:
  "early"  forward-sweep encoders, bottleneck, decoders, pred -- final once the forward-sweep
           BPTT is done (weights are shared over T, so nothing is final earlier);
  "late"   event head, backward-sweep encoders, image branch -- final at the end.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): 64 MB of fp32 gradients is ~0.1-0.7 ms
either way, so four large slices are used rather than many small buckets.
Averaging (1/world) is folded into the fused clip+AdamW kernel (grad_scale).
"""
import os

import torch
import torch.distributed as dist

EARLY_PREFIXES = ("encoders_forward.", "resblocks.", "decoders.", "pred.")


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    if launcher != "pytorch":
        raise ValueError(f"Invalid launcher type: {launcher}")
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if backend == "nccl":
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def bucket_slices(offsets, total):
    """Contiguous [start, end) runs of the arena for the early and late phases."""
    runs = {"early": [], "late": []}
    cur_phase, start = None, 0
    keys = list(offsets.keys())
    for idx, k in enumerate(keys):
        phase = "early" if k.startswith(EARLY_PREFIXES) else "late"
        off = offsets[k][0]
        if phase != cur_phase:
            if cur_phase is not None:
                runs[cur_phase].append((start, off))
            cur_phase, start = phase, off
    runs[cur_phase].append((start, total))
    return runs


def shard_batch(n_samples, rank, world):
    """Rank-strided shard of sample indices (EnlargedSampler: data_sampler.py:32-45)."""
    return list(range(rank, n_samples, world))


class GradSync:
    def __init__(self, flat_grad, offsets, group=None):
        self.flat = flat_grad
        self.group = group
        self.runs = bucket_slices(offsets, flat_grad.numel())
        self.pending = []

    def __call__(self, phase):
        for a, b in self.runs[phase]:
            self.pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))
        if phase == "late":
            for w in self.pending:
                w.wait()
            self.pending = []
