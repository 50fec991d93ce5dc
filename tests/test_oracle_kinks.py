"""CPU: the activation-kink bookkeeping of oracle/kink_tape.py, pinned on the reference-generated fixture.

The fp32 oracle (plain PyTorch fp32 = the reference's own arithmetic; tests/test_oracle_golden.py pins it to the reference's
outputs) is NOT within 5e-6 of the float64 train step on tests/golden/full26_train: it takes the other branch of one ReLU
whose float64 pre-activation is 4e-8 of its tensor's scale, and that one element moves a weight gradient by 5e-5 of its
largest entry.  With that sign decision injected into the float64 step the fp32 oracle is back inside ~5e-6 everywhere.
This is why tests/test_hip_network.py compares the HIP path with the float64 step *given the same sign decisions*."""
import os

import numpy as np
import torch

from oracle import kink_tape as K

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_the_fp32_reference_arithmetic_misses_the_naive_bar_by_one_relu_sign():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    meta = [int(v) for v in np.load(os.path.join(GOLD, "full26_train.npz"))["meta"]]
    loss64, g64, pred64, pre64 = K.run(*meta, dtype=torch.float64)
    loss32, g32, pred32, pre32 = K.run(*meta, dtype=torch.float32)
    # the recorded activations cover every ReLU / LeakyReLU of the path: 163 (conv, occurrence) keys, 19.1 M elements
    assert set(pre64) == set(pre32) and len(pre64) == 163
    assert sum(v.numel() for v in pre64.values()) == 19071296
    # forward values agree to fp32 round-off ...
    assert float((pred32.double() - pred64).abs().max()) < 2e-6
    # ... but the gradients do not meet a naive 5e-6 bar: one weight gradient is 5e-5 of its largest entry away
    naive, where = K.worst_deviation(g32, g64)
    assert naive > 2e-5 and "conv1.weight" in where, (naive, where)
    # because ONE element took the other branch, at a float64 pre-activation far below fp32's resolution of that sum
    force, report = K.flips(pre64, {k: v > 0 for k, v in pre32.items()})
    assert len(report) == 1 and report[0][0] < 1e-7, report
    assert report[0][1][0] == where[:-len(".weight")], (report, where)
    # given the same sign decision the float64 step explains the fp32 oracle to ~5e-6 of every tensor's largest entry
    _, g64f, _, _ = K.run(*meta, dtype=torch.float64, force=force)
    injected, where2 = K.worst_deviation(g32, g64f)
    assert injected < 6e-6 and injected < naive / 8, (injected, where2)
    # and forcing nothing changes nothing (the tape itself is transparent)
    assert float(abs(loss64 - K.run(*meta, dtype=torch.float64, force={})[0])) == 0.0


def test_forced_branches_compose_through_the_double_leaky_relu():
    """evr_level applies LeakyReLU(.2) twice to the first conv's output (rsm:81-82,284-285); a forced element must get the
    composed slope of the OTHER branch (1 <-> 0.04), as the fused epilogue's single mask does."""
    tape = K.KinkTape(force={("c", 0): torch.tensor([1, 2])})
    tape.last, tape.fresh = "c", True
    x = torch.tensor([1.0, 2.0, -3.0, -4.0], dtype=torch.float64, requires_grad=True)
    y = tape.act(tape.act(x, 0.2), 0.2)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.tensor([1.0, 0.04, 1.0, 0.04], dtype=torch.float64))
    assert torch.equal(tape.pre[("c", 0)], x.detach())
