"""Host -> device hand-over of a batch and event voxelisation on the GPU (the steps immediately before the hot path).

``CUDAPrefetcher`` mirrors the reference's class of that name (basicsr/data/prefetch_dataloader.py:84-125, selected by
``prefetch_mode: cuda`` + ``pin_memory: true`` in the dataset options, train.py:197-205): batch k+1 travels host -> HBM on a
side stream while step k computes; ``next()`` makes the compute stream wait for the copy and starts the following one.

Event voxelisation:

Mirrors ``events_to_voxel_grid(events, num_bins, width, height)`` of the reference
(basicsr/data/event_util.py:6-66; events = [N x 4] rows of [timestamp, x, y, polarity], sorted by
time) and the recurrent datasets' slicing of a (2m+n+1)- or (n+1)-bin grid into sliding two-bin
pairs (image_npy_dataset.py:226-232)."""
import ctypes as C

import torch

from ._lib import RefidHipError, check, lib


def events_to_voxel_grid(events, num_bins, width, height, return_format="CHW"):
    """events: (N,4) float64 CUDA/CPU tensor [t, x, y, p]; returns a (num_bins,H,W) float32 CUDA tensor."""
    if events.dim() != 2 or events.shape[1] != 4:
        raise AssertionError("events must be [N x 4]")
    assert num_bins > 0 and width > 0 and height > 0
    if not events.is_cuda:
        events = events.cuda()
    ev = events.to(torch.float64)
    ts = ev[:, 0].contiguous()
    xs = ev[:, 1].to(torch.int32).contiguous()          # .astype(int): truncation
    ys = ev[:, 2].to(torch.int32).contiguous()
    ps = ev[:, 3].to(torch.float32).contiguous()
    first, last = float(ev[0, 0]), float(ev[-1, 0])
    vox = torch.empty((num_bins, height, width), dtype=torch.float32, device=ev.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().refid_events_to_voxel(ts.data_ptr(), xs.data_ptr(), ys.data_ptr(), ps.data_ptr(), ev.shape[0],
                                      num_bins, width, height, first, last, vox.data_ptr(), st),
          "refid_events_to_voxel")
    if return_format == "CHW":
        return vox
    if return_format == "HWC":
        return vox.permute(1, 2, 0)
    raise RefidHipError(f"unknown return_format {return_format}")


def sliding_bin_pairs(voxel):
    """(bins,H,W) -> (bins-1, 2, H, W): adjacent-bin pairs fed to the network as `event`
    (image_npy_dataset.py:226-232)."""
    return torch.stack([voxel[:-1], voxel[1:]], dim=1)


class CUDAPrefetcher:
    """prefetch_dataloader.py:84-125.  ``loader`` is any re-iterable of dict batches whose tensors live in (preferably
    pinned) host memory; ``next()`` returns the batch on the device, or None at the end of an epoch; ``reset()`` starts
    the next epoch.  Additions over the reference: the returned tensors are tied to the consumer stream
    (``record_stream``: the caching allocator must not hand their memory to the NEXT copy while the step still reads
    them), and -- only with ``time_waits=True`` (bench.py) -- the time the compute stream actually had to wait for a copy
    is measured with events (``exposed_ms()``); a training loop keeps no per-iteration state, like the reference class."""

    def __init__(self, loader, opt=None, device=None, time_waits=False):
        self.ori_loader = loader
        self.loader = iter(loader)
        self.opt = opt
        if device is None:
            device = torch.device("cuda" if (opt or {}).get("num_gpu", 1) != 0 else "cpu")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RefidHipError("CUDAPrefetcher: a GPU device is required (the HIP path has no CPU fallback)")
        self.stream = torch.cuda.Stream(device=self.device)
        self.time_waits = bool(time_waits)
        self._waits = []                     # (event before the wait, event after it) on the consumer stream; time_waits only
        self._waited_ms = 0.0                # completed pairs are folded in here, so the list stays bounded
        self.preload()

    def preload(self):
        try:
            self.batch = next(self.loader)
        except StopIteration:
            self.batch = None
            return None
        with torch.cuda.stream(self.stream):
            self.batch = {k: (v.to(device=self.device, non_blocking=True) if torch.is_tensor(v) else v)
                          for k, v in self.batch.items()}

    def next(self):
        cur = torch.cuda.current_stream(self.device)
        if self.time_waits:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            cur.wait_stream(self.stream)
            e1.record(cur)
            self._waits.append((e0, e1))
            while len(self._waits) > 64 and self._waits[0][1].query():        # fold finished pairs: no unbounded event list
                a, b = self._waits.pop(0)
                self._waited_ms += a.elapsed_time(b)
        else:
            cur.wait_stream(self.stream)
        batch = self.batch
        if batch is not None:
            for v in batch.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        self.preload()
        return batch

    def reset(self):
        self.loader = iter(self.ori_loader)
        self.preload()

    def exposed_ms(self, clear=True):
        """Total time (ms) the consumer stream spent waiting for host -> device copies since the last call
        (synchronises the device).  Needs ``time_waits=True``."""
        if not self.time_waits:
            raise RefidHipError("CUDAPrefetcher.exposed_ms: construct with time_waits=True")
        torch.cuda.synchronize(self.device)
        t = self._waited_ms + sum(a.elapsed_time(b) for a, b in self._waits)
        if clear:
            self._waits = []
            self._waited_ms = 0.0
        return t
