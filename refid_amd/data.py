"""Event voxelisation on the GPU (the step immediately before the hot path).

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
