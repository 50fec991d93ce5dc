"""TEST INFRASTRUCTURE (like everything under oracle/): activation-kink bookkeeping for the float64 oracle.

Why.  ReLU / LeakyReLU are not differentiable at 0.  A pre-activation whose exact (float64) value lies below the fp32
resolution of the sum that produced it has no decidable sign in ANY fp32 computation -- the reference's own PyTorch fp32
path included: on tests/golden/full26_train (19.1 M activation elements) 3 float64 pre-activations lie within 1e-7 of
their tensor's largest entry from zero (44 within 1e-6), the closest at 4.0e-8, and the fp32 oracle takes the other branch there,
which moves `encoders_backward.0 ... conv1.weight`'s gradient by 5.0e-5 of its largest entry (and everything upstream of
that mask by 1e-5 .. 4e-5) while every other tensor stays within 2e-6.  A comparison "every gradient within 5e-6 of the
float64 oracle" is therefore a lottery over which side of such a kink an arithmetic lands on, not a test of its accuracy.

What.  `run()` evaluates the oracle's train step while recording every ReLU / LeakyReLU input under the key
(name of the conv that produced it, occurrence of that conv).  `flips()` compares the recorded float64 signs with the signs
another computation took (the HIP path's stored activations, or the fp32 oracle's) and returns the elements that differ
together with their distance from zero relative to the tensor's scale.  `run(force=...)` re-evaluates the float64 step with
exactly those elements taken on the other branch: "the float64 answer given the same sign decisions", against which an
fp32-class computation must again be within the strict bar.  The gate built from this (tests/test_hip_network.py,
tests/test_oracle_kinks.py) is stricter than the old one where it matters -- a differing sign is accepted only if its
float64 pre-activation is below `tau` of the tensor's scale, and the number of such elements is bounded -- and independent
of luck.

Follows the oracle's structure (oracle/refid_oracle.py: conv_layer :227-232, image_encoder_block :240-245, trunk :280-290,
evr_level :293-311, residual_block :314-318), i.e. reference rsm:41-49, 81-84, 284-285, 488-503, 719-726, 755-758.
"""
from collections import defaultdict

import torch

from . import refid_oracle as O


class _FProxy:
    """torch.nn.functional with relu / leaky_relu routed through the tape (everything else untouched)."""

    def __init__(self, real, tape):
        self._real, self._tape = real, tape

    def __getattr__(self, name):
        return getattr(self._real, name)

    def relu(self, x, *a, **k):
        return self._tape.act(x, 0.0)

    def leaky_relu(self, x, negative_slope=0.01, *a, **k):
        return self._tape.act(x, float(negative_slope))


class KinkTape:
    def __init__(self, force=None):
        self.force = force or {}          # key -> LongTensor of flat element indices taken on the OTHER branch
        self.pre = {}                     # key -> detached pre-activation
        self.count = defaultdict(int)
        self.last, self.fresh, self.cur = None, False, None

    # -- hooks ---------------------------------------------------------------------------------
    def conv(self, real):
        def _conv(P, name, x, *a, **k):
            self.last, self.fresh = name, True
            return real(P, name, x, *a, **k)
        return _conv

    def act(self, x, slope):
        if self.fresh:                    # first activation after a conv: a new (name, occurrence)
            self.cur = (self.last, self.count[self.last])
            self.count[self.last] += 1
            self.fresh = False
            self.pre[self.cur] = x.detach().clone()
        # (a second activation on the same value -- evr_level's LeakyReLU after conv_layer's, rsm:81-82,284-285 -- shares the
        #  key: a forced element is forced in both, which gives the composed slope the fused epilogue uses)
        pos = x > 0
        idx = self.force.get(self.cur)
        if idx is not None and idx.numel():
            pos = pos.clone()
            flat = pos.view(-1)
            flat[idx] = ~flat[idx]
        return torch.where(pos, x, x * slope)

    def __enter__(self):
        self._F, self._conv = O.F, O._conv
        O.F = _FProxy(self._F, self)
        O._conv = self.conv(self._conv)
        return self

    def __exit__(self, *exc):
        O.F, O._conv = self._F, self._conv
        return False


def run(img_chn, base, B, T, H, W, seed, dtype=torch.float64, force=None):
    """One oracle train step on the fixture's closed-form parameters / inputs; returns (loss, grads, pred, pre-activations)."""
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed, dtype=dtype)
    x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash", dtype=dtype)
    with KinkTape(force) as tape:
        loss, _, grads, pred = O.train_step({k: v.clone() for k, v in P.items()}, O.TrainState(P), x, ev, gt)
    return loss, grads, pred, tape.pre


def flips(pre64, signs):
    """pre64: key -> float64 pre-activation (NCHW); signs: key -> bool tensor of the same shape (True = the positive branch was
    taken).  Returns (force, report): force = key -> flat indices whose branch differs; report = list of
    (|pre| / max |pre| of the tensor, key, flat index) for every such element, largest first."""
    force, report = {}, []
    for key, pos in signs.items():
        p = pre64[key]
        assert tuple(p.shape) == tuple(pos.shape), (key, tuple(p.shape), tuple(pos.shape))
        diff = ((p > 0) != pos).reshape(-1).nonzero().reshape(-1)
        if diff.numel():
            force[key] = diff
            scale = float(p.abs().max())
            for i in diff.tolist():
                report.append((abs(float(p.reshape(-1)[i])) / scale, key, i))
    report.sort(reverse=True)
    return force, report


def worst_deviation(grads, ref):
    """max over tensors of max |g - ref| / max |ref|, with the tensor's name (tensors whose reference is all zero must be zero)."""
    worst = (0.0, None)
    for k, r in ref.items():
        s = float(r.abs().max())
        g = grads[k].double().cpu() if not isinstance(grads[k], float) else grads[k]
        if s == 0.0:
            assert float(g.abs().max()) == 0.0, k
            continue
        worst = max(worst, (float((g - r).abs().max()) / s, k))
    return worst
