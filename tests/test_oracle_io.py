"""CPU-only: the oracle's restatements of the path's neighbours against reference-generated vectors
(event voxelisation) and hand-computed answers (tile geometry)."""
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
