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
import time

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


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(local_rank):
    """NUMA node of the GPU this rank drives (sysfs: the PCI function's numa_node), or -1 when unknown."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:  # noqa: BLE001  (no GPU, no sysfs, older torch: the caller falls back to an even split)
        return -1


def pin_to_local_cores(local_rank, local_world, numa_nodes=None):
    """One process per GPU enqueues ~5000 kernel launches per step from Python; with 8 ranks on one host the enqueue
    threads must not wander over the sockets.  Pins this process to its share of the cores of its GPU's NUMA node
    (the ranks whose GPUs sit on the same node split that node's cores evenly; unknown topology: an even split of the
    cores the process may use).  Returns the sorted core list it pinned to ([] when the platform cannot pin)."""
    if not hasattr(os, "sched_setaffinity") or local_world < 1:
        return []
    allowed = sorted(os.sched_getaffinity(0))
    if numa_nodes is None:
        numa_nodes = [gpu_numa_node(r) for r in range(local_world)]
    node = numa_nodes[local_rank] if local_rank < len(numa_nodes) else -1
    cores, peers = allowed, list(range(local_world))
    if node >= 0:
        try:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                on_node = sorted(_cpulist(f.read()) & set(allowed))
            if on_node:
                cores, peers = on_node, [r for r in range(local_world) if numa_nodes[r] == node]
        except OSError:
            pass
    idx, n = peers.index(local_rank), len(peers)
    per = max(1, len(cores) // n)
    mine = cores[idx * per:(idx + 1) * per] if idx * per < len(cores) else cores[-per:]
    os.sched_setaffinity(0, mine)
    return mine


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
        self.host_s = {"early": 0.0, "late": 0.0}      # host time spent inside the calls (enqueue only: nothing here may block)
        self.calls = 0

    def __call__(self, phase):
        t0 = time.perf_counter()
        for a, b in self.runs[phase]:
            self.pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))
        if phase == "late":
            for w in self.pending:
                w.wait()
            self.pending = []
            self.calls += 1
        self.host_s[phase] += time.perf_counter() - t0
