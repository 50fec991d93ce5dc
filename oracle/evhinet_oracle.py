"""CPU oracle for the second network of the repository, ``SingleMultiConnectEVHINet``
-- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/`` may import this module.

A functional plain-PyTorch restatement over the reference's state-dict keys of
``archs/single_multiconnect_evhinet_arch.py`` (cited as ``evh``) and ``archs/arch_util.py::FAC_bias``
(``au:421-426``): a single-stage HINet U-Net whose encoder features are modulated (``feat * w + b``)
by an event encoder, with the single-image output ``[conv2(x) + image]`` of the SAM head.

Parity status: PINNED -- ``oracle/make_golden.py`` runs the reference class itself (import stubs only)
on closed-form weights/inputs and commits outputs + gradients as ``tests/golden/evhinet_*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against them.

What the reference's forward really uses (evh:127-167, single version): ``conv_ev1``,
``down_path_ev[0..1]`` (``down_path_ev[2]`` is evaluated but its result never reaches the output, because
the deepest ``down_path_1`` block is called without an event filter, evh:150-151), ``conv_01``,
``down_path_1``, ``skip_conv_1``, ``up_path_1`` and ``sam12.conv2``; ``sam12.conv1/conv3`` are evaluated
and discarded (evh:160-166).  Every other registered module (stage 2, csff, cat12, last) is dead
weight that still has to exist in the state dict.  Dead branches are not evaluated here: results are
identical, and their parameters get zero gradients (the reference leaves ``.grad`` at None).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch
import torch.nn.functional as F

from .refid_oracle import hash_fill

Params = Dict[str, torch.Tensor]


def param_shapes(in_chn: int = 3, ev_chn: int = 6, wf: int = 64, depth: int = 3, fac_place: int = 2,
                 hin_position_left: int = 0, hin_position_right: int = 4) -> "OrderedDict[str, tuple]":
    """State-dict keys/shapes in registration order (evh:68-125, 200-224, 262-285, 318-322)."""
    S: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, co, ci, k, bias=True):
        S[name + ".weight"] = (co, ci, k, k)
        if bias:
            S[name + ".bias"] = (co,)

    def block(name, ci, co, downsample, csff, hin, ev=False):
        conv(name + ".identity", co, ci, 1)
        conv(name + ".conv_1", co, ci, 3)
        conv(name + ".conv_2", co, co, 3)
        if ev:
            conv(name + ".conv_before_merge", 2 * co, co, 1)
        if downsample and csff:
            for n in ("csff_enc", "csff_dec", "csff_enc_mask", "csff_dec_mask"):
                conv(f"{name}.{n}", co, co, 3)
        if hin:
            S[name + ".norm.weight"] = (co // 2,)
            S[name + ".norm.bias"] = (co // 2,)
        if downsample:
            conv(name + ".downsample", co, co, 4, bias=False)

    def chans(i):
        return wf if i == 0 else (2 ** (i - 1)) * wf, (2 ** i) * wf

    for stage, csff in (("down_path_1", False), ("down_path_2", True)):
        for i in range(depth):
            ci, co = chans(i)
            hin = hin_position_left <= i <= hin_position_right
            block(f"{stage}.{i}", ci, co, i + 1 < depth, csff and (i + 1 < depth), hin)
    # registration order in the reference: down_path_1, down_path_2, conv_01, conv_02, down_path_ev, conv_ev1, ...
    conv("conv_01", wf, in_chn, 3)
    conv("conv_02", wf, in_chn, 3)
    for i in range(min(depth, fac_place + 1)):
        ci, co = chans(i)
        block(f"down_path_ev.{i}", ci, co, i + 1 < depth, False, hin_position_left <= i <= hin_position_right, ev=True)
    conv("conv_ev1", wf, ev_chn, 3)
    for stage in ("up_path_1", "up_path_2"):
        for j, i in enumerate(reversed(range(depth - 1))):
            co = (2 ** i) * wf
            S[f"{stage}.{j}.up.weight"] = (2 * co, co, 2, 2)
            S[f"{stage}.{j}.up.bias"] = (co,)
            block(f"{stage}.{j}.conv_block", 2 * co, co, False, False, False)
    for stage in ("skip_conv_1", "skip_conv_2"):
        for j, i in enumerate(reversed(range(depth - 1))):
            conv(f"{stage}.{j}", (2 ** i) * wf, (2 ** i) * wf, 3)
    conv("sam12.conv1", wf, wf, 3)
    conv("sam12.conv2", 3, wf, 3)
    conv("sam12.conv3", wf, 3, 3)
    conv("cat12", wf, 2 * wf, 1)
    conv("last", in_chn, wf, 3)
    return S


def make_params(seed: int = 0, scale: float = 1.0, **kw) -> Params:
    """Closed-form weights (hash_fill), fan-in scaled; norm weights around 1."""
    P: Params = OrderedDict()
    for idx, (k, shp) in enumerate(param_shapes(**kw).items()):
        if k.endswith("norm.weight"):
            P[k] = 1.0 + 0.25 * hash_fill(shp, seed * 1000 + idx)
        elif len(shp) == 4:
            fan = shp[1] * shp[2] * shp[3]
            P[k] = hash_fill(shp, seed * 1000 + idx, scale * (3.0 / fan) ** 0.5)
        else:
            P[k] = 0.1 * hash_fill(shp, seed * 1000 + idx)
    return P


def make_inputs(B: int, H: int, W: int, in_chn: int = 3, ev_chn: int = 6, seed: int = 0):
    x = 0.5 + 0.5 * hash_fill((B, in_chn, H, W), 7000 + seed)
    ev = hash_fill((B, ev_chn, H, W), 7100 + seed)
    gt = 0.5 + 0.5 * hash_fill((B, in_chn, H, W), 7200 + seed)
    return x, ev, gt


def _conv(P, name, x, stride=1, padding=0):
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride, padding)


def hin_lrelu(P, name, x, slope):
    """Half instance norm + LeakyReLU (evh:233-236): InstanceNorm2d(affine, eps 1e-5, biased var) on the
    first half of the channels, identity on the second half."""
    if name + ".norm.weight" in P:
        a, b = torch.chunk(x, 2, dim=1)
        a = F.instance_norm(a, weight=P[name + ".norm.weight"], bias=P[name + ".norm.bias"], eps=1e-5)
        x = torch.cat([a, b], dim=1)
    return F.leaky_relu(x, slope)


def conv_block(P, name, x, slope=0.2):
    """UNetConvBlock / UNetEVConvBlock body up to ``out`` (evh:231-238, 294-301)."""
    o = hin_lrelu(P, name, _conv(P, name + ".conv_1", x, 1, 1), slope)
    o = F.leaky_relu(_conv(P, name + ".conv_2", o, 1, 1), slope)
    return o + _conv(P, name + ".identity", x)


def fac_bias(feat, filt):
    """arch_util.py:421-426."""
    w, b = torch.chunk(filt, 2, dim=1)
    return feat * w + b


def forward(P: Params, x: torch.Tensor, event: torch.Tensor, depth: int = 3, fac_place: int = 2,
            relu_slope: float = 0.2) -> torch.Tensor:
    """evh:127-167 with defaults fac_kernel_size=1, fac_before_downsample=True; returns out_1 (the list's
    only element)."""
    image = x
    ev = []
    e1 = _conv(P, "conv_ev1", event, 1, 1)
    for i in range(min(depth, fac_place + 1)):
        if i + 1 >= depth:
            break                                   # deepest event block: computed by the reference, never used
        name = f"down_path_ev.{i}"
        out = conv_block(P, name, e1, relu_slope)
        ev.append(_conv(P, name + ".conv_before_merge", out))           # merge before downsample (evh:308-309)
        e1 = _conv(P, name + ".downsample", out, 2, 1)
    x1 = _conv(P, "conv_01", image, 1, 1)
    encs = []
    for i in range(depth):
        name = f"down_path_1.{i}"
        out = conv_block(P, name, x1, relu_slope)
        if i + 1 < depth:
            if i <= fac_place:
                out = fac_bias(out, ev[i])                                # evh:246-247
            encs.append(out)
            x1 = _conv(P, name + ".downsample", out, 2, 1)
        else:
            x1 = out
    for j in range(depth - 1):
        name = f"up_path_1.{j}"
        up = F.conv_transpose2d(x1, P[name + ".up.weight"], P[name + ".up.bias"], stride=2)
        bridge = _conv(P, f"skip_conv_1.{j}", encs[-j - 1], 1, 1)
        x1 = conv_block(P, name + ".conv_block", torch.cat([up, bridge], 1), relu_slope)
    return _conv(P, "sam12.conv2", x1, 1, 1) + image                     # SAM: img (evh:44), sam_feature unused


def psnr_loss(pred, target, loss_weight=1.0):
    """losses/losses.py:95-120 (toY=False)."""
    import math
    return loss_weight * (10.0 / math.log(10.0)) * torch.log(((pred - target) ** 2).mean(dim=(1, 2, 3)) + 1e-8).mean()
