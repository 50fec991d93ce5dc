"""Tiled whole-frame inference (BASELINE config 5).

Tile geometry and overlap-averaging follow the reference's ``grids`` / ``grids_inverse``
(basicsr/models/twoImage_event_recurrent_model.py:190-270): num_row = (h-1)//crop+1, adaptive step
ceil((h-crop)/(num_row-1) - 1e-8), last tile clamped to h-crop, overlaps averaged with a count
map.  The reference version unpacks 4-D sizes and therefore crashes on this network's 5-D
`event` (and 5-D sharp `lq`); this one is 5-D aware.  Tiles can be sharded over ranks
(rank-strided), partial sums are all-reduced."""
import ctypes as C
import math

import torch

from ._lib import RefidHipError, check, lib


def tile_origins(size, crop):
    """1-D tile origins with the reference's adaptive step."""
    if crop >= size:
        return [0], min(crop, size)
    num = (size - 1) // crop + 1
    step = crop if num == 1 else math.ceil((size - crop) / (num - 1) - 1e-8)
    out, i, last = [], 0, False
    while i < size and not last:
        if i + crop >= size:
            i = size - crop
            last = True
        out.append(i)
        i += step
    return out, crop


def grid_indices(h, w, crop):
    rows, ch = tile_origins(h, crop)
    cols, cw = tile_origins(w, crop)
    return [dict(i=i, j=j) for i in rows for j in cols], ch, cw


def tiled_forward(net, x, event, crop, max_minibatch=1, rank=0, world=1):
    """x: (1,C,H,W) or (1,2,3,H,W); event: (1,T,2,H,W).  Returns (1,T,3,H,W) (overlap-averaged)."""
    if event.shape[0] != 1:
        raise AssertionError("tiled inference expects batch 1 (reference: assert b == 1)")
    if crop % 8:
        raise RuntimeError("crop_size must be a multiple of 8")
    H, W = event.shape[-2], event.shape[-1]
    idx, ch, cw = grid_indices(H, W, crop)
    T = event.shape[1]
    oc = net.out_chn
    acc = torch.zeros((T * oc, H, W), dtype=torch.float32, device=event.device)
    cnt = torch.zeros((H, W), dtype=torch.float32, device=event.device)
    mine = idx[rank::world]
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)     # noqa: E731
    with torch.no_grad():
        for k in range(0, len(mine), max_minibatch):
            part = mine[k:k + max_minibatch]
            xs = torch.cat([x[..., d["i"]:d["i"] + ch, d["j"]:d["j"] + cw] for d in part], dim=0).contiguous()
            es = torch.cat([event[..., d["i"]:d["i"] + ch, d["j"]:d["j"] + cw] for d in part], dim=0).contiguous()
            out = net(x=xs, event=es)                                        # (n,T,3,ch,cw)
            for n, d in enumerate(part):
                check(lib().refid_tile_add(out[n].data_ptr(), acc.data_ptr(), cnt.data_ptr(), T * oc, ch, cw, H, W,
                                           d["i"], d["j"], st()), "refid_tile_add")
    if world > 1:
        torch.distributed.all_reduce(acc)
        torch.distributed.all_reduce(cnt)
    check(lib().refid_tile_normalize(acc.data_ptr(), cnt.data_ptr(), T * oc, H, W, st()), "refid_tile_normalize")
    return acc.view(1, T, oc, H, W)
