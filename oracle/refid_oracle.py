"""CPU oracle for the REFID hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32/fp64) restatement of the reference network
``FinalBidirectionAttenfusion`` and of the train step around it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product path (``refid_amd``) never does.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
three hot-path files from ``/root/reference`` (with import stubs only, no
source edits), runs them on closed-form inputs/weights and commits the outputs
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors.  The neighbours of the path are pinned the
same way: event voxelisation (``voxel_*.npz``) and the validation tail --
``tensor2img`` + ``calculate_psnr`` + ``calculate_ssim`` (``metrics.npz``,
``tests/test_oracle_io.py``).  The reference itself ships no tests or golden
vectors for this path.

Every function cites the reference lines it follows (paths relative to
``/root/reference/basicsr/models``):
  arch = archs/XXNet_final_attenfusion_arch.py
  rsm  = archs/recurrent_sub_modules.py
  fm   = archs/fusion_modules.py

The network is written functionally over a flat ``dict`` of parameters whose
keys and shapes are exactly the reference's state-dict,
so a released checkpoint's ``['params']`` can be fed in unchanged.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- #
# parameter inventory
# --------------------------------------------------------------------------- #
def param_shapes(img_chn: int, ev_chn: int = 2, out_chn: int = 3, num_encoders: int = 3,
                 base_num_channels: int = 32, num_residual_blocks: int = 2,
                 num_block: int = 1) -> "OrderedDict[str, tuple]":
    """State-dict keys/shapes in the reference's registration order.

    Order follows the ctor: arch:81-128 (base-class ctor registers nothing;
    ``head``, per-level ``encoders_backward``/``encoders_forward`` modules,
    ``head_img``, ``img_encoders``, then resblocks / decoders / pred).
    """
    assert num_block >= 1, "ConvResidualBlocks: at least one ResidualBlockNoBN per trunk (rsm:719-726, make_layer :760-773)"
    b = base_num_channels
    sh: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, co, ci, k, bias=True):
        sh[name + ".weight"] = (co, ci, k, k)
        if bias:
            sh[name + ".bias"] = (co,)

    conv("head.conv2d", b, ev_chn, 5)                                     # arch:98-99
    enc_in = [b * 2 ** i for i in range(num_encoders)]
    enc_out = [b * 2 ** (i + 1) for i in range(num_encoders)]

    def evr(prefix, ci, co, fuse, atten):                                   # rsm:251-268
        conv(prefix + ".conv.conv2d", co, ci, 3)
        if atten:                                                           # fm:238-288
            a = prefix + ".atten_fuse"
            c, dw, ffn = ci, ci, 2 * ci
            sh[a + ".beta"] = (1, c, 1, 1)        # own parameters precede child modules
            sh[a + ".gamma"] = (1, co, 1, 1)
            conv(a + ".conv1", dw, c, 1)
            sh[a + ".conv2.weight"] = (dw, 1, 3, 3); sh[a + ".conv2.bias"] = (dw,)
            conv(a + ".conv1_e", dw, c, 1)
            sh[a + ".conv2_e.weight"] = (dw, 1, 3, 3); sh[a + ".conv2_e.bias"] = (dw,)
            conv(a + ".conv3", c, 2 * dw, 1)
            conv(a + ".se_1.1", dw // 2, dw, 1); conv(a + ".se_1.3", dw, dw // 2, 1)
            conv(a + ".se_2.1", dw // 2, dw, 1); conv(a + ".se_2.3", dw, dw // 2, 1)
            conv(a + ".conv4", ffn, c, 1)
            conv(a + ".conv5", co, ffn, 1)
            conv(a + ".conv_y_side", co, c, 1)
            for n in ("norm1", "norm1_e", "norm2"):
                sh[f"{a}.{n}.weight"] = (c,); sh[f"{a}.{n}.bias"] = (c,)
        t = prefix + ".recurrent_block.forward_trunk.main"
        conv(t + ".0", co, 2 * co, 3)
        for k in range(num_block):                                          # make_layer(ResidualBlockNoBN, num_block)
            conv(t + f".2.{k}.conv1", co, co, 3)
            conv(t + f".2.{k}.conv2", co, co, 3)
        if fuse:
            conv(prefix + ".fuse_two_dir.conv2d", co, 2 * co, 1)
        sh[prefix + ".down.weight"] = (co, co, 4, 4)

    # nn.ModuleList registration: all of encoders_backward, then encoders_forward
    for i in range(num_encoders):
        evr(f"encoders_backward.{i}", enc_in[i], enc_out[i], False, i == 1)
    for i in range(num_encoders):
        evr(f"encoders_forward.{i}", enc_in[i], enc_out[i], True, i == 1)
    conv("head_img.conv2d", b, img_chn, 5)                                 # arch:119-120
    for i in range(num_encoders):                                          # rsm:27-38
        p = f"img_encoders.{i}"
        conv(p + ".identity", enc_out[i], enc_in[i], 1)
        conv(p + ".conv_1", enc_out[i], enc_in[i], 3)
        conv(p + ".conv_2", enc_out[i], enc_out[i], 3)
        sh[p + ".down.weight"] = (enc_out[i], enc_out[i], 4, 4)
    cmax = b * 2 ** num_encoders
    for i in range(num_residual_blocks):                                   # arch:61-64
        conv(f"resblocks.{i}.conv1", cmax, cmax, 3)
        conv(f"resblocks.{i}.conv2", cmax, cmax, 3)
    for j, ci in enumerate(reversed(enc_out)):                             # arch:66-73, rsm:375-384
        p = f"decoders.{j}"
        sh[p + ".transposed_conv2d.weight"] = (ci, ci // 2, 2, 2)
        sh[p + ".transposed_conv2d.bias"] = (ci // 2,)
        t = p + ".forward_trunk.main"
        conv(t + ".0", ci // 2, ci, 3)
        conv(t + ".2.0.conv1", ci // 2, ci // 2, 3)          # (the decoders' trunks always hold ONE block: rsm:375-384 does not
        conv(t + ".2.0.conv2", ci // 2, ci // 2, 3)          #  pass num_block on)
    conv("pred.conv2d", out_chn, b, 3)                                     # arch:75-77
    return sh


def hash_fill(shape, salt: int, scale: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    """Closed-form, framework-independent pseudo-random fill in [-scale, scale).

    value(i) = scale * (2 * frac(sin(i*12.9898 + salt*78.233) * 43758.5453) - 1),
    evaluated in float64 so that it regenerates identically anywhere.
    """
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.float64)
    v = torch.sin(i * 12.9898 + float(salt) * 78.233) * 43758.5453
    v = v - torch.floor(v)
    return ((2.0 * v - 1.0) * scale).reshape(shape).to(dtype)


def make_params(img_chn: int, base_num_channels: int = 32, mode: str = "hash", seed: int = 0,
                dtype=torch.float32, **kw) -> Params:
    """Deterministic parameters for tests/benchmarks.

    mode="hash": closed-form fill scaled like torch's default conv init
    (U(-1/sqrt(fan_in), 1/sqrt(fan_in))), LayerNorm weight ~1, beta/gamma
    NON-zero (they are zero-initialised in the reference, fm:287-288, which
    would silence the attention/FFN branches -- SURVEY.md section 7.1).
    mode="init": the reference's from-scratch initialisation (rsm:752-753,
    776-800; torch defaults elsewhere) drawn from torch.manual_seed(seed).
    """
    shapes = param_shapes(img_chn, base_num_channels=base_num_channels, **kw)
    P: Params = OrderedDict()
    g = torch.Generator().manual_seed(seed)
    for idx, (k, s) in enumerate(shapes.items()):
        # torch's default conv init: U(+-1/sqrt(fan_in)), fan_in = size(1)*kh*kw of the
        # weight (also for ConvTranspose2d), same bound for the bias
        ws = s if len(s) == 4 else shapes.get(k[:-4] + "weight", (1, 1, 1, 1))
        fan_in = ws[1] * ws[2] * ws[3] if len(ws) == 4 else 1
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        if mode == "hash":
            if ".norm" in k and k.endswith(".weight"):
                t = 1.0 + hash_fill(s, seed * 1000 + idx, 0.25)
            elif ".norm" in k:
                t = hash_fill(s, seed * 1000 + idx, 0.1)
            elif k.endswith((".beta", ".gamma")):
                t = hash_fill(s, seed * 1000 + idx, 0.5)
            else:
                t = hash_fill(s, seed * 1000 + idx, bound)
        elif mode == "init":
            if ".norm" in k:
                t = torch.ones(s) if k.endswith(".weight") else torch.zeros(s)
            elif k.endswith((".beta", ".gamma")):
                t = torch.zeros(s)
            elif ".main.2." in k and ".conv" in k:                         # rsm:752-753,776-800 (every ResidualBlockNoBN)
                if k.endswith(".weight"):
                    std = math.sqrt(2.0 / (s[1] * s[2] * s[3]))
                    t = torch.randn(s, generator=g) * std * 0.1
                else:
                    t = torch.zeros(s)
            else:
                t = (torch.rand(s, generator=g) * 2 - 1) * bound
        else:
            raise ValueError(mode)
        P[k] = t.to(dtype)
    return P


def make_inputs(B: int, T: int, H: int, W: int, img_chn: int, seed: int = 0, mode: str = "hash",
                dtype=torch.float32):
    """Synthetic (x, event, gt) with the value distributions of SURVEY.md 8(d).

    mode="hash": closed form (fixtures).  mode="rng": numpy PCG64(seed) --
    frames/gt U[0,1); voxels 85 % zeros, rest N(0,1) rounded to 1/8, clip +-4;
    for img_chn=26 the 20 deblur-voxel channels of x are voxel-like.
    """
    if mode == "hash":
        x = (hash_fill((B, img_chn, H, W), seed * 7 + 1) + 1.0) * 0.5
        ev = hash_fill((B, T, 2, H, W), seed * 7 + 2, 2.0)
        ev = torch.where(ev.abs() < 1.4, torch.zeros_like(ev), torch.round(ev * 8) / 8)
        gt = (hash_fill((B, T, 3, H, W), seed * 7 + 3) + 1.0) * 0.5
        if img_chn == 6:
            x = x.reshape(B, 2, 3, H, W)
        return x.to(dtype), ev.to(dtype), gt.to(dtype)
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))

    def vox(shape):
        v = np.round(rng.standard_normal(shape, dtype=np.float32) * 8) / 8
        v = np.clip(v, -4, 4)
        v[rng.random(shape, dtype=np.float32) < 0.85] = 0
        return v.astype(np.float32)

    x = rng.random((B, img_chn, H, W), dtype=np.float32)
    if img_chn == 26:                                   # image_npy_dataset.py:211-221
        x[:, 3:13] = vox((B, 10, H, W))
        x[:, 16:26] = vox((B, 10, H, W))
    ev = vox((B, T, 2, H, W))
    gt = rng.random((B, T, 3, H, W), dtype=np.float32)
    x = torch.from_numpy(x)
    if img_chn == 6:
        x = x.reshape(B, 2, 3, H, W)
    return x.to(dtype), torch.from_numpy(ev).to(dtype), torch.from_numpy(gt).to(dtype)


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def _conv(P: Params, name: str, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride, padding, 1, groups)


def conv_layer(P, name, x, padding, relu_slope: Optional[float]):
    """ConvLayer.forward, norm=None: conv (+ LeakyReLU).  rsm:52-84."""
    out = _conv(P, name + ".conv2d", x, 1, padding)
    if relu_slope is not None:
        out = F.leaky_relu(out, relu_slope)
    return out


def conv_down(P, name, x):
    """conv_down: 4x4, stride 2, pad 1, no bias.  rsm:12-14."""
    return F.conv2d(x, P[name + ".weight"], None, 2, 1)


def image_encoder_block(P, name, x):
    """ImageEncoderConvBlock.forward.  rsm:41-49."""
    out1 = F.leaky_relu(_conv(P, name + ".conv_1", x, 1, 1), 0.2)
    out2 = F.leaky_relu(_conv(P, name + ".conv_2", out1, 1, 1), 0.2)
    out = out2 + _conv(P, name + ".identity", x)
    return conv_down(P, name + ".down", out)


def layer_norm2d(x, weight, bias, eps=1e-6):
    """LayerNormFunction.forward (per-pixel LN over C, biased var).  fm:97-108."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + eps).sqrt()
    return weight.view(1, -1, 1, 1) * y + bias.view(1, -1, 1, 1)


def egaca(P, a, ev, img, taps: Optional[dict] = None):
    """CrossmodalAtten_imgeventalladd.forward (EGACA).  fm:290-333.

    se_1 is applied to x_e for BOTH products (fm:312-313); se_2 is dead.
    """
    dw = ev.shape[1]
    x = layer_norm2d(img, P[a + ".norm1.weight"], P[a + ".norm1.bias"])
    xe = layer_norm2d(ev, P[a + ".norm1_e.weight"], P[a + ".norm1_e.bias"])
    x = F.gelu(_conv(P, a + ".conv2", _conv(P, a + ".conv1", x), 1, 1, dw))
    xe = F.gelu(_conv(P, a + ".conv2_e", _conv(P, a + ".conv1_e", xe), 1, 1, dw))
    m = F.adaptive_avg_pool2d(xe, 1)
    s = torch.sigmoid(_conv(P, a + ".se_1.3", F.relu(_conv(P, a + ".se_1.1", m))))
    zf = _conv(P, a + ".conv3", torch.cat((x * s, xe * s), dim=1))
    y = ev + img + zf * P[a + ".beta"]
    f = _conv(P, a + ".conv5", F.gelu(_conv(P, a + ".conv4",
              layer_norm2d(y, P[a + ".norm2.weight"], P[a + ".norm2.bias"]))))
    out = _conv(P, a + ".conv_y_side", y) + f * P[a + ".gamma"]
    if taps is not None:
        taps["egaca_se"] = s.detach().clone()
        taps["egaca_y"] = y.detach().clone()
        taps["egaca_out"] = out.detach().clone()
    return out


def trunk(P, name, u, h):
    """ConvResidualBlocks on cat(u, h) with one ResidualBlockNoBN.

    rsm:719-726 (conv3x3 2C->C, LeakyReLU 0.1, blocks), rsm:755-758
    (identity + conv2(relu(conv1(x)))).  ``h=None`` -> zeros (rsm:666-668, 395-397).
    """
    if h is None:
        h = torch.zeros_like(u)
    v = F.leaky_relu(_conv(P, name + ".0", torch.cat([u, h], dim=1), 1, 1), 0.1)
    k = 0
    while f"{name}.2.{k}.conv1.weight" in P:                               # num_block ResidualBlockNoBN (rsm:760-773)
        v = v + _conv(P, f"{name}.2.{k}.conv2", F.relu(_conv(P, f"{name}.2.{k}.conv1", v, 1, 1)), 1, 1)
        k += 1
    return v


def evr_level(P, name, level: int, x, y, prev_state, bi_state, taps=None):
    """SimpleRecurrentThenDownAttenfusionmodifiedConvLayer.forward.  rsm:270-296.

    The first conv is a ConvLayer that already applies LeakyReLU(0.2) (rsm:81-82)
    and the layer applies LeakyReLU(0.2) again (rsm:284-285) -> slope 0.04.
    Level 1 (use_atten_fuse) replaces conv by EGACA (rsm:275-276).
    """
    if y is not None:
        if level == 1:
            x = egaca(P, name + ".atten_fuse", x, y, taps)
        else:
            x = F.leaky_relu(conv_layer(P, name + ".conv", x + y, 1, 0.2), 0.2)
    else:
        x = F.leaky_relu(conv_layer(P, name + ".conv", x, 1, 0.2), 0.2)
    state = trunk(P, name + ".recurrent_block.forward_trunk.main", x, prev_state)
    x = state
    if bi_state is not None:
        x = conv_layer(P, name + ".fuse_two_dir", torch.cat((x, bi_state), 1), 0, 0.2)
    return conv_down(P, name + ".down", x), state


def residual_block(P, name, x):
    """ResidualBlock.forward, norm=None: relu(conv2(relu(conv1(x))) + x).  rsm:488-503."""
    out = F.relu(_conv(P, name + ".conv1", x, 1, 1))
    out = _conv(P, name + ".conv2", out, 1, 1)
    return F.relu(out + x)


def decoder_level(P, name, x, prev_state):
    """TransposeRecurrentConvLayer.forward.  rsm:386-408."""
    out = F.conv_transpose2d(x, P[name + ".transposed_conv2d.weight"],
                             P[name + ".transposed_conv2d.bias"], stride=2)
    state = trunk(P, name + ".forward_trunk.main", out, prev_state)
    return state, state


# --------------------------------------------------------------------------- #
# the network (A0)
# --------------------------------------------------------------------------- #
def forward(P: Params, x: torch.Tensor, event: torch.Tensor, num_encoders: int = 3,
            num_residual_blocks: int = 2, taps: Optional[dict] = None) -> torch.Tensor:
    """FinalBidirectionAttenfusion.forward.  arch:130-218.

    Reproduces the list-aliasing behaviour of arch:167,180-181: every forward
    step fuses the FINAL backward-sweep state of each level.
    ``taps`` (optional dict) receives intermediate tensors for the fixtures.
    """
    if x.dim() == 5:                                                       # arch:140-141
        b_, t_, c_, h_, w_ = x.shape
        x = x.reshape(b_, t_ * c_, h_, w_)
    b, t, nb, h, w = event.shape                                           # arch:142-143
    if h % 8 or w % 8:
        # the reference fails with a shape error at the first skip-sum; SURVEY 8(b)
        raise RuntimeError(f"H and W must be multiples of 8, got {h}x{w}")
    ev = event.reshape(b * t, nb, h, w)
    head = conv_layer(P, "head_img", x, 2, 0.2)                             # arch:147-148
    e = conv_layer(P, "head", ev, 2, 0.2).reshape(b, t, -1, h, w)           # arch:149,158
    x_blocks: List[torch.Tensor] = []
    xi = head
    for i in range(num_encoders):                                           # arch:152-154
        xi = image_encoder_block(P, f"img_encoders.{i}", xi)
        x_blocks.append(xi)
    if taps is not None:
        taps["head"] = head.detach().clone()
        taps["e"] = e.detach().clone()
        for i, xb in enumerate(x_blocks):
            taps[f"x_block{i}"] = xb.detach().clone()

    bstate: List[Optional[torch.Tensor]] = [None] * num_encoders
    for fi in range(t - 1, -1, -1):                                         # arch:172-181
        cur = e[:, fi]
        for i in range(num_encoders):
            cur, st = evr_level(P, f"encoders_backward.{i}", i, cur,
                                None if i == 0 else x_blocks[i - 1], bstate[i], None)
            bstate[i] = st
    final_b = bstate                                # aliasing: arch:180-181 -> final states only
    if taps is not None:
        for i in range(num_encoders):
            taps[f"final_bstate{i}"] = final_b[i].detach().clone()

    fstate: List[Optional[torch.Tensor]] = [None] * num_encoders
    dstate: List[Optional[torch.Tensor]] = [None] * num_encoders
    outs = []
    for fi in range(t):                                                     # arch:185-216
        cur = e[:, fi]
        e_blocks = []
        tp = taps if (taps is not None and fi == t - 1) else None
        for i in range(num_encoders):
            cur, st = evr_level(P, f"encoders_forward.{i}", i, cur,
                                None if i == 0 else x_blocks[i - 1], fstate[i], final_b[i], tp)
            e_blocks.append(cur)
            fstate[i] = st
            if taps is not None and fi in (0, t - 1):
                taps[f"fwd_out{i}_t{fi}"] = cur.detach().clone()
                taps[f"fwd_state{i}_t{fi}"] = st.detach().clone()
        for i in range(num_residual_blocks):                                # arch:199-203
            cur = residual_block(P, f"resblocks.{i}", cur + x_blocks[-1] if i == 0 else cur)
        if taps is not None and fi == t - 1:
            taps["bottleneck_tlast"] = cur.detach().clone()
        for i in range(num_encoders):                                       # arch:210-212
            cur, st = decoder_level(P, f"decoders.{i}", cur + e_blocks[num_encoders - i - 1], dstate[i])
            dstate[i] = st
            if taps is not None and fi == t - 1:
                taps[f"dec_state{i}_tlast"] = st.detach().clone()
        outs.append(conv_layer(P, "pred", cur + head, 1, None))             # arch:215
    return torch.stack(outs, dim=1)                                         # arch:218


# --------------------------------------------------------------------------- #
# train step (S1-S3) and evaluation tail
# --------------------------------------------------------------------------- #
def charbonnier(pred, gt, eps: float = 1e-12):
    """CharbonnierLoss, weight 1, mean.  losses/losses.py:28-30,143-173."""
    return torch.sqrt((pred - gt) ** 2 + eps).mean()


def cosine_lr(base_lr: float, step_index: int, t_max: int, eta_min: float) -> float:
    """torch CosineAnnealingLR closed form (base_model.py:91-95); step_index = scheduler.last_epoch."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step_index / t_max)) / 2


class TrainState:
    """AdamW state for the oracle train step."""

    def __init__(self, P: Params):
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.step = 0


def train_step(P: Params, state: TrainState, x, event, gt, lr=2e-4, betas=(0.9, 0.99), eps=1e-8,
               weight_decay=1e-4, clip=0.01, **kw):
    """optimize_parameters: twoImage_event_recurrent_model.py:273-310.

    zero_grad -> forward -> Charbonnier -> (+0*sum p) -> backward ->
    clip_grad_norm_(0.01) -> AdamW (torch semantics, decoupled decay applied to
    every parameter, including the 13 that only ever see zero gradients).
    Updates ``P`` in place; returns (loss, total_grad_norm_before_clip, grads).
    """
    leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    pred = forward(leaves, x, event, **kw)
    loss = charbonnier(pred, gt)
    total = loss + 0 * sum(p.sum() for p in leaves.values())
    total.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(clip / (gnorm + 1e-6), max=1.0)                      # torch clip_grad_norm_
    state.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** state.step
    bc2 = 1 - b2 ** state.step
    with torch.no_grad():
        for k, p in P.items():
            g = grads[k] * coef
            p.mul_(1 - lr * weight_decay)
            state.m[k].mul_(b1).add_(g, alpha=1 - b1)
            state.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (state.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(state.m[k], denom, value=-lr / bc1)
    return loss.detach(), gnorm, grads, pred.detach()


def tensor2img_u8(frame: torch.Tensor) -> torch.Tensor:
    """tensor2img quantisation: clamp [0,1], x255, round -> uint8.  utils/img_util.py:90-117.

    (The RGB->BGR flip there is PSNR-invariant and omitted.)  frame: (3,H,W).  Pinned: tests/golden/metrics.npz."""
    return (frame.detach().float().clamp(0, 1) * 255.0).round().to(torch.uint8)


def psnr_u8(a: torch.Tensor, b: torch.Tensor) -> float:
    """calculate_psnr on uint8 images, crop_border 0, float64 MSE.  metrics/psnr_ssim.py:48-63.
    Pinned against the reference function's own output: tests/golden/metrics.npz."""
    mse = ((a.double() - b.double()) ** 2).mean().item()
    if mse == 0:
        return float("inf")
    return 20.0 * math.log10(255.0 / math.sqrt(mse))


def psnr_between(out_a: torch.Tensor, out_b: torch.Tensor) -> float:
    """PSNR (dB, peak 1) between two float outputs -- used for bf16/fp32 parity."""
    mse = ((out_a.double() - out_b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


# --------------------------------------------------------------------------- #
# callers either side of the path
# --------------------------------------------------------------------------- #
def events_to_voxel_grid(events, num_bins, width, height):
    """numpy restatement of basicsr/data/event_util.py:6-66 (np.int -> int; otherwise line by line).
    events: (N,4) float64 [t, x, y, p]; NOT modified in place (the reference overwrites column 0)."""
    import numpy as np
    events = np.array(events, dtype=np.float64, copy=True)
    voxel_grid = np.zeros((num_bins, height, width), np.float32).ravel()
    last_stamp, first_stamp = events[-1, 0], events[0, 0]
    deltaT = last_stamp - first_stamp
    if deltaT == 0:
        deltaT = 1.0
    ts = (num_bins - 1) * (events[:, 0] - first_stamp) / deltaT
    xs = events[:, 1].astype(int)
    ys = events[:, 2].astype(int)
    pols = events[:, 3].copy()
    pols[pols == 0] = -1
    tis = ts.astype(int)
    dts = ts - tis
    vals_left = pols * (1.0 - dts)
    vals_right = pols * dts
    valid = tis < num_bins
    np.add.at(voxel_grid, xs[valid] + ys[valid] * width + tis[valid] * width * height, vals_left[valid])
    valid = (tis + 1) < num_bins
    np.add.at(voxel_grid, xs[valid] + ys[valid] * width + (tis[valid] + 1) * width * height, vals_right[valid])
    return np.reshape(voxel_grid, (num_bins, height, width))


def tile_grid(h, w, crop):
    """Tile origins of grids(): twoImage_event_recurrent_model.py:190-240 (trans_num=1, no random crops)."""
    num_row = (h - 1) // crop + 1
    num_col = (w - 1) // crop + 1
    step_j = crop if num_col == 1 else math.ceil((w - crop) / (num_col - 1) - 1e-8)
    step_i = crop if num_row == 1 else math.ceil((h - crop) / (num_row - 1) - 1e-8)
    idxes = []
    i, last_i = 0, False
    while i < h and not last_i:
        j = 0
        if i + crop >= h:
            i = h - crop
            last_i = True
        last_j = False
        while j < w and not last_j:
            if j + crop >= w:
                j = w - crop
                last_j = True
            idxes.append({"i": i, "j": j})
            j = j + step_j
        i = i + step_i
    return idxes


def ssim3d_u8(a: torch.Tensor, b: torch.Tensor) -> float:
    """calculate_ssim -> _ssim_3d restated: metrics/psnr_ssim.py:135-182 (+ :283-290 max_value) on
    tensor2img-quantised frames.  a, b: (3,H,W) float in [0,1].  (cv2.getGaussianKernel(11,1.5) =
    normalised exp(-(i-5)^2/(2*1.5^2)); the reference runs the conv3d in fp32 on the GPU.)
    Pinned against the reference function's own output (run on the CPU): tests/golden/metrics.npz."""
    import numpy as np
    g = np.exp(-((np.arange(11) - 5) ** 2) / (2 * 1.5 ** 2)); g = g / g.sum()
    window = np.outer(g, g)
    kernel = torch.tensor(np.stack([window * k for k in g], axis=0)).float()
    conv3d = torch.nn.Conv3d(1, 1, (11, 11, 11), stride=1, padding=(5, 5, 5), bias=False, padding_mode="replicate")
    conv3d.weight.requires_grad = False
    conv3d.weight[0, 0] = kernel
    img1 = tensor2img_u8(a).permute(1, 2, 0).float()          # HWC, values 0..255
    img2 = tensor2img_u8(b).permute(1, 2, 0).float()
    max_value = 1 if float(img1.max()) <= 1 else 255
    C1, C2 = (0.01 * max_value) ** 2, (0.03 * max_value) ** 2
    f = lambda t: conv3d(t.unsqueeze(0).unsqueeze(0)).squeeze(0).squeeze(0)   # noqa: E731
    with torch.no_grad():
        mu1, mu2 = f(img1), f(img2)
        mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
        s1 = f(img1 ** 2) - mu1_sq
        s2 = f(img2 ** 2) - mu2_sq
        s12 = f(img1 * img2) - mu1_mu2
        m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return float(m.mean())
