"""CPU-only: the oracle's restatements of the path's neighbours against reference-generated vectors
(event voxelisation, the tensor2img / PSNR / SSIM validation tail) and hand-computed answers (tile geometry)."""
import os

import numpy as np
import pytest

from oracle import refid_oracle as O


@pytest.mark.parametrize("name", ["voxel_a", "voxel_b"])
def test_voxelisation_matches_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    bins, h, w = [int(v) for v in z["meta"]]
    ev = z["events"]
    vox = O.events_to_voxel_grid(ev, bins, w, h)
    np.testing.assert_array_equal(vox, z["voxel"])           # same numpy ops, same order: bit exact
    assert np.array_equal(ev, z["events"])                   # input not clobbered


def test_voxelisation_edge_cases():
    # single time stamp: deltaT == 0 -> 1.0 (event_util.py:33-34); polarity 0 counts as -1
    ev = np.array([[2.0, 1, 1, 0.0], [2.0, 2, 1, 1.0]])
    v = O.events_to_voxel_grid(ev, 3, 4, 3)
    assert v[0, 1, 1] == -1.0 and v[0, 1, 2] == 1.0 and v[1:].sum() == 0
    # last event lands exactly on the last bin with dt = 0
    ev = np.array([[0.0, 0, 0, 1.0], [1.0, 3, 2, 1.0]])
    v = O.events_to_voxel_grid(ev, 2, 4, 3)
    assert v[0, 0, 0] == 1.0 and v[1, 2, 3] == 1.0


def test_tile_geometry_known_answers():
    # BASELINE config 5: 1224 x 1632 with 512 tiles -> 3 x 4 tiles, adaptive steps 356 / 374
    idx = O.tile_grid(1224, 1632, 512)
    assert sorted({d["i"] for d in idx}) == [0, 356, 712]
    assert sorted({d["j"] for d in idx}) == [0, 374, 748, 1120]
    assert len(idx) == 12
    assert O.tile_grid(256, 256, 256) == [{"i": 0, "j": 0}]
    # the product's geometry is the same function of (h, w, crop)
    from refid_amd.tiling import grid_indices
    for h, w, c in [(1224, 1632, 512), (720, 1280, 256), (256, 256, 256), (264, 520, 256), (512, 512, 384)]:
        got, ch, cw = grid_indices(h, w, c)
        assert got == O.tile_grid(h, w, c) and ch == c and cw == c
    cover = np.zeros((264, 520), int)
    for d in O.tile_grid(264, 520, 256):
        cover[d["i"]:d["i"] + 256, d["j"]:d["j"] + 256] += 1
    assert cover.min() >= 1


def test_validation_tail_matches_reference(golden_dir):
    """tests/golden/metrics.npz holds what the reference's own tensor2img (utils/img_util.py:59-121), calculate_psnr
    (metrics/psnr_ssim.py:9-63) and calculate_ssim (:135-182,225-303) return for three frame pairs
    (oracle/make_golden.py::run_metrics): the restatement must agree."""
    import torch
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    for n in z["names"]:
        pred, gt = torch.from_numpy(z[f"{n}/pred"]), torch.from_numpy(z[f"{n}/gt"])
        for t, key in ((pred, "pred_u8_bgr"), (gt, "gt_u8_bgr")):
            ref = torch.from_numpy(z[f"{n}/{key}"].copy()).permute(2, 0, 1).flip(0)      # HWC BGR -> CHW RGB
            assert torch.equal(O.tensor2img_u8(t), ref)                                    # quantisation: bit exact
        assert abs(O.psnr_u8(O.tensor2img_u8(pred), O.tensor2img_u8(gt)) - float(z[f"{n}/psnr"])) < 1e-10
        assert abs(O.ssim3d_u8(pred, gt) - float(z[f"{n}/ssim"])) < 1e-6                   # fp32 conv3d, summation order
    gt = torch.from_numpy(z["c/gt"])
    assert float(z["same/psnr"]) == float("inf") and O.psnr_u8(O.tensor2img_u8(gt), O.tensor2img_u8(gt)) == float("inf")
    assert abs(O.ssim3d_u8(gt, gt) - float(z["same/ssim"])) < 1e-6


def test_ssim_known_answers():
    """Closed-form cases of the SSIM definition (next to the reference-generated vectors above)."""
    import torch
    a = torch.rand(3, 24, 24, generator=torch.Generator().manual_seed(0))
    assert abs(O.ssim3d_u8(a, a) - 1.0) < 1e-6                           # identical images
    c1 = torch.full((3, 24, 24), 100 / 255.0); c2 = torch.full((3, 24, 24), 140 / 255.0)
    C1 = (0.01 * 255) ** 2                                               # constant images: variance terms vanish
    expect = (2 * 100 * 140 + C1) / (100 ** 2 + 140 ** 2 + C1)
    assert abs(O.ssim3d_u8(c1, c2) - expect) < 2e-3      # fp32 E[x^2]-mu^2 cancellation at x~100
    assert O.ssim3d_u8(a, 1 - a) < 0.2                                    # anti-correlated
