"""GPU parity of the fused conv tile / wgrad C-ABI entry points against torch CPU fp64
(the same torch.nn.functional ops the oracle is made of).  Tolerance: fp32 round-off of
a K-long dot product, rtol 1e-4 / atol 1e-5 on O(1) data (north-star bar is 1e-3/1e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-5


def _ops():
    from refid_amd import ops
    return ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).double().cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def lrelu(x, s):
    return torch.where(x > 0, x, x * s)


def run_fwd(N, H, W, Ca, Cb, Co, k, stride, pad, bias=True, res=False, slope_pre=1.0, slope_post=1.0,
            mask=False, pad_ca=None):
    ops = _ops()
    Ci = Ca + Cb
    xa = rnd(N, Ca, H, W, seed=1)
    xb = rnd(N, Cb, H, W, seed=2) if Cb else None
    w = rnd(Co, Ci, k, k, seed=3, scale=1.0 / np.sqrt(Ci * k * k))
    b = rnd(Co, seed=4) if bias else None
    x = torch.cat([xa, xb], 1) if Cb else xa
    ref = F.conv2d(x, w, b, stride, pad)
    ref = lrelu(ref, slope_pre)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = rnd(N, Co, Ho, Wo, seed=5) if res else None
    if res:
        ref = ref + r
    ref = lrelu(ref, slope_post)
    m = rnd(N, Co, Ho, Wo, seed=6) if mask else None
    if mask:
        ref = ref * torch.where(m > 0, 1.0, 0.3)
    kc, bn = ops.conv_kc(k, k, stride), ops.conv_bn(k, k, stride, 0, Co)
    wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_FWD, bn, kc, k, k, Co, Ci)
    copad = -(-Co // bn) * bn
    a_dev = nhwc(xa)
    if pad_ca:      # channel-padded source (e.g. 2 -> 4, 26 -> 28): extra channels are zero
        buf = torch.zeros(N, H, W, pad_ca, device="cuda")
        buf[..., :Ca] = a_dev
        a_dev = buf
    Cop = -(-Co // 4) * 4
    outbuf = torch.full((N, Ho, Wo, Cop), 7.0, device="cuda")
    out = outbuf[..., :Co]
    ops.conv2d(a_dev, wp, out, kh=k, kw=k, stride=stride, pad=pad, cout=Co, cout_pad=copad,
               in_b=nhwc(xb) if Cb else None, bias=b.float().cuda() if bias else None,
               res=nhwc(r) if res else None, mask=nhwc(m) if mask else None,
               slope_pre=slope_pre, slope_post=slope_post, slope_mask=0.3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    if Cop != Co:
        assert float(outbuf[..., Co:].min()) == 7.0      # pad channels untouched


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co, k, s, p
    (2, 16, 32, 32, 0, 64, 3, 1, 1),          # EvR L0 first conv
    (1, 24, 40, 64, 64, 64, 3, 1, 1),         # trunk main.0 (two-source), ragged tiles
    (1, 8, 8, 128, 128, 128, 3, 1, 1),        # 128-wide tile, tiny image
    (1, 12, 20, 64, 0, 256, 3, 1, 1),         # two cout tiles
    (2, 16, 16, 32, 32, 32, 3, 1, 1),         # decoder level 2 (32-wide tile)
    (1, 16, 32, 32, 0, 3, 3, 1, 1),           # pred: 3 output channels
    (1, 16, 24, 2, 0, 32, 5, 1, 2),           # event head: Cin=2 (padded to 4)
    (1, 16, 24, 26, 0, 32, 5, 1, 2),          # image head: Cin=26 (padded to 28)
    (2, 8, 16, 64, 64, 64, 1, 1, 0),          # fuse_two_dir
    (1, 16, 16, 32, 0, 64, 1, 1, 0),          # identity 1x1
    (1, 8, 8, 128, 0, 128, 1, 1, 0),
    (1, 8, 8, 64, 0, 32, 1, 1, 0),
    (2, 16, 32, 64, 0, 64, 4, 2, 1),          # conv_down
    (1, 24, 40, 128, 0, 128, 4, 2, 1),
    (1, 16, 16, 32, 0, 64, 2, 2, 0),          # 2x2 stride 2 (convT input gradient)
    (1, 8, 16, 64, 0, 128, 2, 2, 0),
])
def test_conv_forward_geometries(cfg):
    N, H, W, Ca, Cb, Co, k, s, p = cfg
    pad_ca = (-(-Ca // 4) * 4) if Ca % 4 else None
    run_fwd(N, H, W, Ca, Cb, Co, k, s, p, pad_ca=pad_ca)


def test_conv_fused_epilogues():
    run_fwd(1, 16, 32, 64, 0, 64, 3, 1, 1, slope_pre=0.04)                        # double LeakyReLU(.2)
    run_fwd(1, 16, 32, 64, 0, 64, 3, 1, 1, res=True)                                # trunk residual
    run_fwd(1, 8, 32, 128, 0, 128, 3, 1, 1, res=True, slope_post=0.0)               # ResidualBlock
    run_fwd(1, 8, 32, 64, 64, 64, 3, 1, 1, slope_pre=0.1, bias=True)                # trunk main.0
    run_fwd(1, 8, 32, 64, 0, 64, 3, 1, 1, bias=False, res=True, mask=True)          # dgrad-style epilogue


def _pack_dgrad(ops, w, role, k, stride, mode, rows, kdim):
    kc = ops.conv_kc(k, k, stride, mode)
    bn = ops.conv_bn(k, k, stride, mode, rows)
    o, i = (w.shape[1], w.shape[0]) if role in (ops.ROLE_CONVT, ops.ROLE_CONVT_DGRAD) else (w.shape[0], w.shape[1])
    return ops.pack_conv_weights(w.float().cuda(), role, bn, kc, w.shape[2], w.shape[3], o, i), -(-rows // bn) * bn


@pytest.mark.parametrize("cfg", [(1, 16, 32, 32, 64, 3), (1, 8, 24, 128, 64, 3), (2, 8, 8, 64, 128, 1),
                                 (1, 8, 16, 256, 128, 3)])
def test_conv_dgrad_stride1(cfg):
    ops = _ops()
    N, H, W, Ci, Co, k = cfg
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=2, scale=0.1)
    g = rnd(N, Co, H, W, seed=3)
    F.conv2d(x, w, None, 1, k // 2).backward(g)
    wp, rp = _pack_dgrad(ops, w, ops.ROLE_DGRAD, k, 1, 0, Ci, Co)
    out = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(nhwc(g), wp, out, kh=k, kw=k, stride=1, pad=k // 2, cout=Ci, cout_pad=rp)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)


def test_conv_dgrad_split_rows():
    """dgrad of a two-source conv computed as two row ranges of the same packed weights."""
    ops = _ops()
    N, H, W, C = 1, 8, 32, 64
    x = rnd(N, 2 * C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, 2 * C, 3, 3, seed=2, scale=0.1)
    g = rnd(N, C, H, W, seed=3)
    F.conv2d(x, w, None, 1, 1).backward(g)
    wp, rp = _pack_dgrad(ops, w, ops.ROLE_DGRAD, 3, 1, 0, 2 * C, C)
    gd = nhwc(g)
    for base in (0, C):
        out = torch.empty(N, H, W, C, device="cuda")
        ops.conv2d(gd, wp, out, kh=3, kw=3, stride=1, pad=1, cout=C, cout_pad=rp, co_base=base)
        np.testing.assert_allclose(nchw(out).numpy(), x.grad[:, base:base + C].numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [(2, 16, 32, 64), (1, 24, 40, 128), (1, 8, 8, 256)])
def test_conv_down_dgrad(cfg):
    ops = _ops()
    N, H, W, C = cfg
    x = rnd(N, C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, C, 4, 4, seed=2, scale=0.1)
    y = F.conv2d(x, w, None, 2, 1)
    g = rnd(*y.shape, seed=3)
    y.backward(g)
    wp, rp = _pack_dgrad(ops, w, ops.ROLE_DOWN_DGRAD, 4, 2, 2, C, C)
    out = torch.empty(N, H, W, C, device="cuda")
    ops.conv2d(nhwc(g), wp, out, kh=4, kw=4, stride=2, pad=1, mode=2, cout=C, cout_pad=rp)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [(2, 8, 8, 256, 128), (1, 12, 20, 128, 64), (1, 16, 32, 64, 32)])
def test_conv_transpose_fwd_and_dgrad(cfg):
    ops = _ops()
    N, H, W, Ci, Co = cfg
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Ci, Co, 2, 2, seed=2, scale=0.1)
    b = rnd(Co, seed=3)
    y = F.conv_transpose2d(x, w, b, stride=2)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    wp, rp = _pack_dgrad(ops, w, ops.ROLE_CONVT, 1, 1, 1, 4 * Co, Ci)
    out = torch.empty(N, 2 * H, 2 * W, Co, device="cuda")
    ops.conv2d(nhwc(x.detach()), wp, out, kh=1, kw=1, stride=1, pad=0, mode=1, cout=4 * Co, cout_pad=rp,
               bias=b.float().cuda())
    np.testing.assert_allclose(nchw(out).numpy(), y.detach().numpy(), rtol=RTOL, atol=ATOL)
    # the same GEMM on the register-operand pointwise tile (algo 3, mode 1): pixel-shuffle store, bias per real channel,
    # + residual (sample / row / column wrap of the 32-pixel wave tiles: W = 8, 20, 32)
    wq = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_CONVT, 32, 8, 2, 2, Co, Ci)
    out3 = torch.full((N, 2 * H, 2 * W, Co), 7.0, device="cuda")
    ops.conv2d(nhwc(x.detach()), wq, out3, kh=1, kw=1, stride=1, pad=0, mode=1, cout=4 * Co, cout_pad=-(-4 * Co // 32) * 32,
               bias=b.float().cuda(), algo=3)
    np.testing.assert_allclose(nchw(out3).numpy(), y.detach().numpy(), rtol=RTOL, atol=ATOL)
    r = rnd(N, Co, 2 * H, 2 * W, seed=5)
    ops.conv2d(nhwc(x.detach()), wq, out3, kh=1, kw=1, stride=1, pad=0, mode=1, cout=4 * Co, cout_pad=-(-4 * Co // 32) * 32,
               bias=b.float().cuda(), res=nhwc(r), slope_post=0.2, algo=3)
    np.testing.assert_allclose(nchw(out3).numpy(), F.leaky_relu(y.detach() + r, 0.2).numpy(), rtol=RTOL, atol=ATOL)
    wd, rd = _pack_dgrad(ops, w, ops.ROLE_CONVT_DGRAD, 2, 2, 0, Ci, Co)
    dx = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(nhwc(g), wd, dx, kh=2, kw=2, stride=2, pad=0, cout=Ci, cout_pad=rd)
    np.testing.assert_allclose(nchw(dx).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    # the same input gradient as ONE patch GEMM (K = 4 Co) on the pointwise tile, with a mask and a second output
    wq = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_CONVT_DGRAD_PW, 32, 8, 2, 2, Co, Ci)
    dx3 = torch.full((N, H, W, Ci), 7.0, device="cuda")
    ops.conv2d(nhwc(g), wq, dx3, kh=2, kw=2, stride=2, pad=0, cout=Ci, cout_pad=-(-Ci // 32) * 32, algo=3)
    np.testing.assert_allclose(nchw(dx3).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    m, plus = rnd(N, Ci, H, W, seed=6), rnd(N, Ci, H, W, seed=7)
    o2 = torch.empty_like(dx3)
    ops.conv2d(nhwc(g), wq, dx3, kh=2, kw=2, stride=2, pad=0, cout=Ci, cout_pad=-(-Ci // 32) * 32, algo=3, mask=nhwc(m),
               slope_mask=0.1, add2=nhwc(plus), out2=o2)
    want = x.grad * torch.where(m > 0, 1.0, 0.1)
    np.testing.assert_allclose(nchw(dx3).numpy(), want.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(nchw(o2).numpy(), (want + plus).numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co, k, s, p
    (2, 16, 32, 64, 0, 64, 3, 1, 1),
    (1, 24, 40, 64, 64, 64, 3, 1, 1),
    (1, 8, 8, 256, 256, 256, 3, 1, 1),
    (1, 16, 32, 32, 32, 32, 3, 1, 1),         # narrow-output tile
    (1, 16, 32, 32, 0, 64, 3, 1, 1),          # narrow-input tile
    (2, 8, 32, 32, 0, 32, 3, 1, 1),           # both narrow
    (1, 8, 32, 32, 0, 3, 3, 1, 1),            # pred: 3 real output rows (g padded to 4) -- thin-output streaming kernel
    (2, 9, 17, 32, 0, 3, 3, 1, 1),            #   ... ragged rows / columns, two samples
    (1, 5, 3, 16, 0, 1, 3, 1, 1),             #   ... one output channel, 16 input channels
    (3, 12, 20, 32, 0, 4, 3, 1, 1),           #   ... four real output channels
    (1, 16, 24, 28, 0, 32, 5, 1, 2),
    (2, 8, 16, 128, 128, 128, 1, 1, 0),
    (1, 16, 16, 32, 0, 64, 1, 1, 0),
    (2, 9, 13, 64, 0, 128, 1, 1, 0),          # pointwise tile: odd pixel count, 4x1 waves
    (1, 8, 8, 128, 0, 64, 1, 1, 0),           # 2x2 waves, SN=2
    (1, 7, 5, 64, 64, 32, 1, 1, 0),           # 1x4 waves, two sources
    (1, 8, 8, 256, 256, 256, 1, 1, 0),        # several channel tiles
    (1, 8, 8, 16, 0, 16, 1, 1, 0),            # narrow: LDS-tiled fallback
    (2, 16, 32, 64, 0, 64, 4, 2, 1),
    (1, 24, 40, 128, 0, 128, 4, 2, 1),
    (1, 16, 16, 32, 0, 64, 2, 2, 0),
])
def test_conv_wgrad(cfg):
    ops = _ops()
    N, H, W, Ca, Cb, Co, k, s, p = cfg
    x = rnd(N, Ca + Cb, H, W, seed=1)
    w = rnd(Co, Ca + Cb, k, k, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, s, p)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    dw = torch.zeros(Co, Ca + Cb, k, k, device="cuda")
    db = torch.zeros(Co, device="cuda")
    xa = nhwc(x[:, :Ca])
    xb = nhwc(x[:, Ca:]) if Cb else None
    gd = nhwc(g)
    if Co % 4:              # channel-padded gradient (pred): pad channel carries garbage-free zeros
        buf = torch.zeros(*gd.shape[:3], -(-Co // 4) * 4, device="cuda")
        buf[..., :Co] = gd
        gd = buf
    for _ in range(2):      # accumulates: two calls -> 2x
        ops.conv2d_wgrad(gd, xa, dw, kh=k, kw=k, stride=s, pad=p, in_b=xb, db=db)
    scale = max(1.0, float(w.grad.abs().max()))
    np.testing.assert_allclose(dw.double().cpu().numpy() / 2, w.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
    np.testing.assert_allclose(db.double().cpu().numpy() / 2, b.grad.numpy(), rtol=RTOL, atol=ATOL * scale)


@pytest.mark.parametrize("cfg", [(2, 16, 24, 32, 3), (1, 9, 40, 32, 4)])
def test_thin_output_wgrad_phases_and_groups(cfg):
    """The streaming form of the thin-output 3x3 weight gradient (`pred`, arch:215) writes the plan's own slabs: persistent
    phases (overwrite / add / reduce) and grouped time steps give the one-shot gradients' sum; REFID_THINOUT_WGRAD=0 (the
    MFMA tile it replaces) is compared in a subprocess-free way through torch."""
    ops = _ops()
    N, H, W, Ci, Co = cfg
    w = rnd(Co, Ci, 3, 3, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    steps = []
    for t in range(3):
        x = rnd(N, Ci, H, W, seed=10 + t)
        g = rnd(N, Co, H, W, seed=20 + t)
        F.conv2d(x, w, b, 1, 1).backward(g)
        gd = nhwc(g)
        if Co % 4:
            buf = torch.zeros(*gd.shape[:3], 4, device="cuda"); buf[..., :Co] = gd; gd = buf
        steps.append((gd, nhwc(x), None))
    kw = dict(kh=3, kw=3, stride=1, pad=1)
    for grouping in ([[0], [1], [2]], [[0, 1, 2]], [[0], [1, 2]]):
        dw = torch.zeros(Co, Ci, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
        sl, first = None, True
        for grp in grouping:
            (g0, a0, _), more = steps[grp[0]], [steps[i] for i in grp[1:]]
            sl = ops.conv2d_wgrad(g0, a0, dw, db=db, phase=1 if first else 2, slabs=sl, more=more, **kw)
            first = False
        ops.conv2d_wgrad(steps[0][0], steps[0][1], dw, db=db, phase=3, slabs=sl, **kw)
        scale = max(1.0, float(w.grad.abs().max()))
        np.testing.assert_allclose(dw.double().cpu().numpy(), w.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
        np.testing.assert_allclose(db.double().cpu().numpy(), b.grad.numpy(), rtol=RTOL, atol=ATOL * scale)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co
    (2, 16, 24, 64, 0, 64), (1, 9, 13, 128, 0, 64), (2, 8, 16, 64, 0, 128), (1, 16, 16, 128, 0, 128), (1, 7, 5, 64, 64, 64),
    (1, 8, 8, 256, 256, 256), (1, 12, 20, 32, 0, 64), (3, 5, 7, 128, 128, 128), (1, 4, 4, 96, 0, 96)])
def test_pointwise_streaming_wgrad(cfg):
    """csrc/wgrad_pws.hip: the streaming 1x1 weight gradient (LDS-DMA ring, every (OW, WI) instantiation, two sources, ragged
    pixel counts, the first recurrent step's missing second source) against torch; persistent phases and grouped time steps
    give the one-shot gradients' sum."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    Ci = Ca + Cb
    w = rnd(Co, Ci, 1, 1, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    steps = []
    for t in range(3):
        x = rnd(N, Ci, H, W, seed=10 + t)
        g = rnd(N, Co, H, W, seed=20 + t)
        F.conv2d(x, w, b).backward(g)
        steps.append((nhwc(g), nhwc(x[:, :Ca]), nhwc(x[:, Ca:]) if Cb else None))
    kw = dict(kh=1, kw=1, stride=1, pad=0, i_total=Ci)
    scale = max(1.0, float(w.grad.abs().max()))
    for grouping in ([[0], [1], [2]], [[0, 1, 2]], [[0], [1, 2]]):
        dw = torch.zeros(Co, Ci, 1, 1, device="cuda"); db = torch.zeros(Co, device="cuda")
        sl, first = None, True
        for grp in grouping:
            (g0, a0, b0), more = steps[grp[0]], [steps[i] for i in grp[1:]]
            sl = ops.conv2d_wgrad(g0, a0, dw, in_b=b0, db=db, phase=1 if first else 2, slabs=sl, more=more, **kw)
            first = False
        ops.conv2d_wgrad(steps[0][0], steps[0][1], dw, in_b=steps[0][2], db=db, phase=3, slabs=sl, **kw)
        np.testing.assert_allclose(dw.double().cpu().numpy(), w.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
        np.testing.assert_allclose(db.double().cpu().numpy(), b.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
    if Cb:      # first recurrent step: the second source does not exist yet, the slab geometry already covers it
        dw1 = torch.zeros(Co, Ci, 1, 1, device="cuda")
        s1 = ops.conv2d_wgrad(steps[0][0], steps[0][1], dw1, phase=1, **kw)
        ops.conv2d_wgrad(steps[0][0], steps[0][1], dw1, phase=3, slabs=s1, **kw)
        x0 = rnd(N, Ci, H, W, seed=10)
        g0 = rnd(N, Co, H, W, seed=20)
        ref = torch.einsum("nohw,nihw->oi", g0, x0[:, :Ca])
        assert float((dw1[:, :Ca, 0, 0].double().cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
        assert float(dw1[:, Ca:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co
    (2, 32, 64, 64, 0, 64), (1, 40, 24, 32, 0, 64), (1, 16, 32, 64, 64, 128), (2, 12, 20, 32, 0, 32), (1, 64, 64, 128, 0, 128)])
def test_conv_down_wgrad_through_parity_phases(cfg):
    """refid_wgrad_desc.algo 7: the weight gradient of conv_down (4x4, stride 2, pad 1: recurrent_sub_modules.py:12-14) on the
    2x4-tile Winograd kernel through the four parity phases of the input, against torch; persistent phases and grouped time
    steps give the one-shot gradients' sum; and its distance from the float64 gradient stays in the fp32 class."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    Ci = Ca + Cb
    w = rnd(Co, Ci, 4, 4, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    steps = []
    for t in range(3):
        x = rnd(N, Ci, H, W, seed=10 + t)
        x = torch.where(x > -0.3, x, 0.1 * x) + 0.2
        g = rnd(N, Co, H // 2, W // 2, seed=20 + t, scale=1e-2)
        F.conv2d(x, w, b, 2, 1).backward(g)
        steps.append((nhwc(g), nhwc(x[:, :Ca]), nhwc(x[:, Ca:]) if Cb else None))
    kw = dict(kh=4, kw=4, stride=2, pad=1, i_total=Ci, algo=7)
    for grouping in ([[0], [1], [2]], [[0, 1, 2]], [[0], [1, 2]]):
        dw = torch.zeros(Co, Ci, 4, 4, device="cuda"); db = torch.zeros(Co, device="cuda")
        sl, first = None, True
        for grp in grouping:
            (g0, a0, b0), more = steps[grp[0]], [steps[i] for i in grp[1:]]
            sl = ops.conv2d_wgrad(g0, a0, dw, in_b=b0, db=db, phase=1 if first else 2, slabs=sl, more=more, **kw)
            first = False
        ops.conv2d_wgrad(steps[0][0], steps[0][1], dw, in_b=steps[0][2], db=db, phase=3, slabs=sl, **kw)
        err = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
        assert err < 2e-5, (grouping, err)
        assert float((db.double().cpu() - b.grad).abs().max() / b.grad.abs().max()) < 1e-5
    # the direct tile on the same operands: the two forms agree to the fp32 class
    dw0 = torch.zeros(Co, Ci, 4, 4, device="cuda")
    for (g0, a0, b0) in steps:
        ops.conv2d_wgrad(g0, a0, dw0, kh=4, kw=4, stride=2, pad=1, in_b=b0, i_total=Ci)
    assert float((dw0 - dw).abs().max() / dw0.abs().max()) < 3e-5


def test_conv_transpose_wgrad():
    ops = _ops()
    N, H, W, Ci, Co = 1, 8, 16, 128, 64
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Ci, Co, 2, 2, seed=2).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, stride=2)
    g = rnd(*y.shape, seed=3)
    y.backward(g)
    dw = torch.zeros(Ci, Co, 2, 2, device="cuda")
    # roles swapped: "g" := layer input (low res), "src" := output gradient (high res) -> IOHW
    ops.conv2d_wgrad(nhwc(x), nhwc(g), dw, kh=2, kw=2, stride=2, pad=0)
    np.testing.assert_allclose(dw.double().cpu().numpy(), w.grad.numpy(), rtol=RTOL, atol=ATOL * 4)


@pytest.mark.parametrize("cfg", [(2, 8, 32, 128, 64), (1, 6, 64, 64, 32), (3, 4, 32, 256, 128), (1, 5, 96, 64, 16)])
def test_conv_transpose_wgrad_as_streaming_patch_gemm(cfg):
    """refid_wgrad_desc.algo 8: the 2x2 stride-2 weight gradient over non-overlapping patches as ONE streaming 1x1 weight gradient
    (K = (dy, dx, co) over the even / odd rows of the output gradient, columns permuted into IOHW by the reduction): one-shot,
    and grouped time steps with persistent slabs + the queued reduction (phase 4 + flush); against autograd and the direct tile."""
    ops = _ops()
    N, H, W, Ci, Co = cfg
    T = 3
    w = rnd(Ci, Co, 2, 2, seed=2).requires_grad_(True)
    steps = []
    for t in range(T):
        x = rnd(N, Ci, H, W, seed=10 + t)
        y = F.conv_transpose2d(x, w, None, stride=2)
        g = rnd(*y.shape, seed=20 + t)
        y.backward(g)
        steps.append((nhwc(x), nhwc(g)))
    scale = float(w.grad.abs().max())
    dw = torch.zeros(Ci, Co, 2, 2, device="cuda")
    for xs, gs in steps:                                            # one-shot calls accumulate
        ops.conv2d_wgrad(xs, gs, dw, kh=2, kw=2, stride=2, pad=0, algo=8)
    assert float((dw.double().cpu() - w.grad).abs().max()) / scale < 2e-5
    dw0 = torch.zeros_like(dw)
    for xs, gs in steps:
        ops.conv2d_wgrad(xs, gs, dw0, kh=2, kw=2, stride=2, pad=0)
    assert float((dw0 - dw).abs().max()) / scale < 2e-5
    dw2 = torch.zeros_like(dw)                                      # grouped: steps 0+1 in one launch, step 2 added, queued reduction
    sl = ops.conv2d_wgrad(steps[0][0], steps[0][1], dw2, kh=2, kw=2, stride=2, pad=0, algo=8, phase=1,
                          more=[(steps[1][0], steps[1][1], None)])
    sl = ops.conv2d_wgrad(steps[2][0], steps[2][1], dw2, kh=2, kw=2, stride=2, pad=0, algo=8, phase=2, slabs=sl)
    ops.conv2d_wgrad(steps[0][0], steps[0][1], dw2, kh=2, kw=2, stride=2, pad=0, algo=8, phase=4, slabs=sl)
    ops.wgrad_finish_flush()
    assert float((dw2.double().cpu() - w.grad).abs().max()) / scale < 2e-5


def test_layout_and_elementwise():
    ops = _ops()
    x = rnd(2, 26, 8, 24, seed=1).float()
    d = ops.nchw_to_nhwc(x.cuda(), 28)
    assert d.shape == (2, 8, 24, 28)
    np.testing.assert_array_equal(d[..., :26].permute(0, 3, 1, 2).cpu().numpy(), x.numpy())
    assert float(d[..., 26:].abs().max()) == 0.0
    stack = torch.zeros(2, 3, 3, 8, 24, device="cuda")           # (B,T,3,H,W)
    src = torch.rand(2, 8, 24, 4, device="cuda")
    ops.nhwc_to_nchw(src[..., :3], 3, stack[:, 1], dst_batch_stride=3 * 3 * 8 * 24)
    np.testing.assert_array_equal(stack[:, 1].cpu().numpy(), src[..., :3].permute(0, 3, 1, 2).cpu().numpy())
    assert float(stack[:, 0].abs().max()) == 0.0 and float(stack[:, 2].abs().max()) == 0.0
    # whole (B,T,C,H,W) stacks in one launch, time-major on the NHWC side (sample t B + b)
    ev = torch.rand(3, 5, 2, 8, 24, device="cuda")
    tb = ops.nchw_to_nhwc_tb(ev, 4)
    assert tb.shape == (15, 8, 24, 4) and float(tb[..., 2:].abs().max()) == 0.0
    np.testing.assert_array_equal(tb[..., :2].reshape(5, 3, 8, 24, 2).permute(1, 0, 4, 2, 3).cpu().numpy(), ev.cpu().numpy())
    sl = ops.nchw_to_nhwc_tb(ev[:2], 4)                             # a batch slice: batch stride != T * block
    np.testing.assert_array_equal(sl.reshape(5, 2, 8, 24, 4).cpu().numpy(), tb.reshape(5, 3, 8, 24, 4)[:, :2].cpu().numpy())
    back = torch.zeros(3, 5, 2, 8, 24, device="cuda")
    ops.nhwc_to_nchw_tb(tb[..., :2], 2, back)
    np.testing.assert_array_equal(back.cpu().numpy(), ev.cpu().numpy())
    a, b = torch.rand(2, 8, 8, 64, device="cuda"), torch.rand(2, 8, 8, 64, device="cuda") - 0.5
    np.testing.assert_array_equal(ops.add(a, b).cpu().numpy(), (a + b).cpu().numpy())
    o = ops.act_bwd(a, b, 0.1)
    np.testing.assert_allclose(o.cpu().numpy(), (a * torch.where(b > 0, 1.0, 0.1)).cpu().numpy(), rtol=1e-6)
    o2 = ops.act_bwd(a, b, 0.1, out=o.clone(), accumulate=True)
    np.testing.assert_allclose(o2.cpu().numpy(), 2 * o.cpu().numpy(), rtol=1e-6)


def test_errors_are_loud():
    ops = _ops()
    from refid_amd._lib import RefidHipError
    x = torch.zeros(1, 8, 8, 64, device="cuda")
    with pytest.raises(RefidHipError):
        ops.conv2d(x, x, torch.zeros(1, 8, 8, 64, device="cuda"), kh=7, kw=7, cout=64, cout_pad=64)
    with pytest.raises(RefidHipError):
        ops.conv2d(x.cpu(), x, x, kh=3, kw=3, pad=1, cout=64, cout_pad=64)


# ---------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) tile (algo=1): same contract as the direct tile
# ---------------------------------------------------------------------------------------------
def run_wino(N, H, W, Ca, Cb, Co, bias=True, res=False, slope_pre=1.0, slope_post=1.0, mask=False, algo=1, terms=0):
    ops = _ops()
    Ci = Ca + Cb
    xa = rnd(N, Ca, H, W, seed=1)
    xb = rnd(N, Cb, H, W, seed=2) if Cb else None
    w = rnd(Co, Ci, 3, 3, seed=3, scale=1.0 / np.sqrt(Ci * 9))
    b = rnd(Co, seed=4) if bias else None
    x = torch.cat([xa, xb], 1) if Cb else xa
    ref = lrelu(F.conv2d(x, w, b, 1, 1), slope_pre)
    r = rnd(N, Co, H, W, seed=5) if res else None
    if res:
        ref = ref + r
    ref = lrelu(ref, slope_post)
    m = rnd(N, Co, H, W, seed=6) if mask else None
    if mask:
        ref = ref * torch.where(m > 0, 1.0, 0.3)
    if algo == 5:
        wp = ops.pack_conv_weights_wino6(w.float().cuda(), ops.ROLE_WINO_FWD, Co, Ci, f16=terms == 3)
    else:
        wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    copad = -(-Co // 64) * 64
    Cop = -(-Co // 4) * 4
    outbuf = torch.full((N, H, W, Cop), 7.0, device="cuda")
    out = outbuf[..., :Co]
    ops.conv2d(nhwc(xa), wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=copad, algo=algo, terms=terms,
               in_b=nhwc(xb) if Cb else None, bias=b.float().cuda() if bias else None,
               res=nhwc(r) if res else None, mask=nhwc(m) if mask else None,
               slope_pre=slope_pre, slope_post=slope_post, slope_mask=0.3)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    if Cop != Co:
        assert float(outbuf[..., Co:].min()) == 7.0


@pytest.mark.parametrize("cfg", [
    (2, 16, 32, 32, 0, 64), (1, 24, 40, 64, 64, 64), (1, 8, 8, 128, 128, 128), (1, 12, 20, 64, 0, 256),
    (2, 16, 16, 32, 32, 32), (1, 5, 3, 64, 0, 64), (1, 7, 33, 32, 0, 96), (1, 16, 32, 32, 0, 3),
])
def test_winograd_forward_geometries(cfg):
    run_wino(*cfg)


def test_winograd_fused_epilogues():
    run_wino(1, 16, 32, 64, 0, 64, slope_pre=0.04)
    run_wino(1, 16, 32, 64, 0, 64, res=True)
    run_wino(1, 8, 32, 128, 0, 128, res=True, slope_post=0.0)
    run_wino(1, 8, 32, 64, 64, 64, slope_pre=0.1)
    run_wino(1, 8, 32, 64, 0, 64, bias=False, res=True, mask=True)


# ---------------------------------------------------------------------------------------------
# Winograd x six bf16 products (algo=5, csrc/conv_wino6.hip): same contract, 8 output channels or more
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    (2, 16, 32, 32, 0, 64), (1, 24, 40, 64, 64, 64), (1, 8, 8, 128, 128, 128), (1, 12, 20, 64, 0, 256),
    (1, 5, 3, 64, 0, 64), (1, 7, 33, 32, 0, 96), (1, 9, 17, 36, 0, 64), (1, 8, 32, 16, 16, 64), (1, 6, 10, 8, 0, 40),
    (3, 4, 64, 48, 0, 128), (1, 8, 8, 256, 256, 256),
    # the 32-output-channel form (round 4: one column tile per wave, three workgroups per CU): decoder 2's trunk shapes,
    # ragged sizes, a partial channel tile (24, 12 of 32), two sources, a split-K grid (256 -> 32 at 8x8)
    (2, 16, 32, 32, 0, 32), (1, 24, 40, 32, 32, 32), (1, 7, 33, 64, 0, 24), (1, 5, 3, 36, 0, 12), (1, 8, 8, 256, 0, 32),
    # thin outputs (pred: 32 -> 3, written into a 4-channel buffer), one and four channels
    (1, 16, 32, 32, 0, 3), (2, 9, 17, 32, 0, 3), (1, 8, 32, 16, 0, 1), (1, 8, 32, 32, 0, 4),
])
@pytest.mark.parametrize("terms", [0, 3])
def test_wino6_forward_geometries(cfg, terms):
    """Ragged tiles, a partial last 16-channel chunk (36, 8, 48 channels), a partial channel tile (96, 40), two sources,
    and a small grid that takes the split-K form (512 -> 256 at 8x8); six bf16 products (terms 0) and three fp16 products on
    scaled two-plane operands (terms 3, round 6)."""
    run_wino(*cfg, algo=5, terms=terms)


@pytest.mark.parametrize("terms", [0, 3])
def test_wino6_fused_epilogues(terms):
    kw = dict(algo=5, terms=terms)
    run_wino(1, 16, 32, 64, 0, 64, slope_pre=0.04, **kw)
    run_wino(1, 16, 32, 64, 0, 64, res=True, **kw)
    run_wino(1, 8, 32, 128, 0, 128, res=True, slope_post=0.0, **kw)
    run_wino(1, 8, 32, 64, 64, 64, slope_pre=0.1, **kw)
    run_wino(1, 8, 32, 64, 0, 64, bias=False, res=True, mask=True, **kw)
    # the 32-output-channel form: decoder 2's trunk (main.0 two-source + LeakyReLU(.1), ReLU, + identity) and its
    # input gradients (residual + activation mask), a ragged size with a partial channel tile
    run_wino(1, 16, 32, 32, 32, 32, slope_pre=0.1, **kw)
    run_wino(1, 16, 32, 32, 0, 32, slope_pre=0.0, **kw)
    run_wino(1, 16, 32, 32, 0, 32, res=True, **kw)
    run_wino(1, 8, 32, 32, 0, 32, bias=False, res=True, mask=True, **kw)
    run_wino(1, 9, 37, 64, 0, 20, bias=False, res=True, mask=True, slope_post=0.0, **kw)


@pytest.mark.parametrize("terms", [0, 3])
@pytest.mark.parametrize("cfg", [(1, 16, 32, 64, 64), (1, 9, 24, 128, 64), (1, 8, 16, 256, 128), (1, 8, 8, 96, 40)])
def test_wino6_dgrad(cfg, terms):
    ops = _ops()
    N, H, W, Ci, Co = cfg
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    g = rnd(N, Co, H, W, seed=3)
    F.conv2d(x, w, None, 1, 1).backward(g)
    wd = ops.pack_conv_weights_wino6(w.float().cuda(), ops.ROLE_WINO_DGRAD, Co, Ci, f16=terms == 3)
    rp = -(-Ci // 64) * 64
    gd = nhwc(g)
    out = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(gd, wd, out, kh=3, kw=3, stride=1, pad=1, cout=Ci, cout_pad=rp, algo=5, terms=terms)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    if Ci >= 128:       # row-range issue (two-source convs): second half of the rows
        half = Ci // 2
        o2 = torch.empty(N, H, W, half, device="cuda")
        ops.conv2d(gd, wd, o2, kh=3, kw=3, stride=1, pad=1, cout=half, cout_pad=rp, co_base=half, algo=5, terms=terms)
        np.testing.assert_allclose(nchw(o2).numpy(), x.grad[:, half:].numpy(), rtol=RTOL, atol=ATOL)


def test_wino6_accuracy_class_and_tiny_gradients():
    """Six exact-split bf16 products per fp32 product keep the fp32 class: largest deviation from the float64
    convolution at K = 9 x 256 within 4e-6 of the output scale (the fp32 Winograd tile's class), and unchanged in RELATIVE
    terms for operands of magnitude 1e-8 (input gradients of a mean loss are that small; bf16 keeps fp32's exponent)."""
    ops = _ops()
    N, H, W, Ci, Co = 1, 32, 64, 256, 64
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=3.0 / np.sqrt(Ci * 9))
    ref = F.conv2d(x, w, None, 1, 1)
    scale = float(ref.abs().max())
    w6 = ops.pack_conv_weights_wino6(w.float().cuda(), ops.ROLE_WINO_FWD, Co, Ci)
    w1 = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    errs = {}
    for algo, wp in ((5, w6), (1, w1)):
        out = torch.empty(N, H, W, Co, device="cuda")
        ops.conv2d(nhwc(x), wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=64, algo=algo)
        errs[algo] = float((nchw(out) - ref).abs().max())
    assert errs[5] < 4e-6 * scale and errs[1] < 4e-6 * scale, errs
    assert errs[5] < 2.0 * errs[1], errs
    out = torch.empty(N, H, W, Co, device="cuda")
    ops.conv2d(nhwc(x * 1e-8), w6, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=64, algo=5)
    assert float((nchw(out) * 1e8 - ref).abs().max()) < 4e-6 * scale


def test_wino_f16_accuracy_class_and_dynamic_range():
    """Three fp16 products on two-plane operands (algo 5, mfma_terms 3; round 6).  The planes carry 22 bits and fp16's range is
    bridged by exact power-of-two scales (U per packing, V per tile, online along K), so:
    (a) O(1) data, K = 9 x 256: within 4e-6 of the output scale of the float64 convolution and within 2x the fp32 Winograd
        tile's own deviation -- the six-product form's gate, unchanged;
    (b) the same RELATIVE error for operands of magnitude 1e-8 and 1e+6, and for weights of magnitude 1e-6 / 1e+3 (no fixed
        scale could do that inside fp16's 2^-14 .. 2^16);
    (c) channels whose magnitude GROWS by 2^40 along K (every chunk 2^10 above the previous one: the accumulator rescale runs
        at every chunk) and shrinks again: the error stays relative to the result;
    (d) tiles of wildly different magnitude next to each other (per-tile scale), zero tiles and zero chunks."""
    ops = _ops()
    N, H, W, Ci, Co = 1, 32, 64, 256, 64
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=3.0 / np.sqrt(Ci * 9))

    def run(xx, ww, terms=3, algo=5):
        if algo == 5:
            wp = ops.pack_conv_weights_wino6(ww.float().cuda(), ops.ROLE_WINO_FWD, Co, Ci, f16=terms == 3)
        else:
            wp = ops.pack_conv_weights(ww.float().cuda(), ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
        out = torch.empty(N, H, W, Co, device="cuda")
        ops.conv2d(nhwc(xx), wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=64, algo=algo, terms=terms if algo == 5 else 0)
        return nchw(out)

    ref = F.conv2d(x, w, None, 1, 1)
    scale = float(ref.abs().max())
    e3 = float((run(x, w) - ref).abs().max())
    e6 = float((run(x, w, terms=0) - ref).abs().max())
    e1 = float((run(x, w, algo=1) - ref).abs().max())
    assert e3 < 4e-6 * scale and e3 < 2.0 * e1, (e3, e6, e1, scale)                       # (a)
    for sx, sw in ((1e-8, 1.0), (1e6, 1.0), (1.0, 1e-6), (1.0, 1e3), (1e-8, 1e3)):          # (b)
        xs, ws = (x * sx).float().double(), (w * sw).float().double()                       # (what the kernel is given)
        r = F.conv2d(xs, ws, None, 1, 1)
        assert float((run(xs, ws) - r).abs().max()) < 4e-6 * float(r.abs().max()), (sx, sw)
    # (c) chunk c of the input channels scaled by 2^(10 c) for c < 5, falling again afterwards
    ramp = torch.tensor([2.0 ** (10 * min(c, 9 - c)) if c < 10 else 1.0 for c in range(Ci // 16)], dtype=torch.float64)
    xc = (x * ramp.repeat_interleave(16).view(1, Ci, 1, 1)).float().double()
    r = F.conv2d(xc, w.float().double(), None, 1, 1)
    assert float((run(xc, w) - r).abs().max()) < 4e-6 * float(r.abs().max())
    # ... measured per output pixel too: a pixel whose window holds only the small channels' data is not drowned either
    xz = xc.clone()
    xz[:, 64:96] = 0.0                                       # the 2^40 chunks are zero chunks now
    r = F.conv2d(xz, w.float().double(), None, 1, 1)
    assert float((run(xz, w) - r).abs().max()) < 4e-6 * float(r.abs().max())
    # (d) per-tile scales: image columns scaled by 2^(-20 .. +20), a zero band, compared PER COLUMN BAND
    col = torch.tensor([2.0 ** (((c // 4) % 11 - 5) * 4) for c in range(W)], dtype=torch.float64)
    col[20:28] = 0.0
    xt = (x * col.view(1, 1, 1, W)).float().double()
    r = F.conv2d(xt, w.float().double(), None, 1, 1)
    got = run(xt, w)
    for c0 in range(0, W, 4):
        band = r[..., c0:c0 + 4].abs().max()
        # (a band's window reaches one pixel into its neighbours: its error is relative to the largest of the three)
        near = r[..., max(0, c0 - 4):c0 + 8].abs().max()
        assert float((got[..., c0:c0 + 4] - r[..., c0:c0 + 4]).abs().max()) <= 4e-6 * float(near) + 1e-30, (c0, float(band))


def test_wino6_rejects_bad_arguments():
    ops = _ops()
    from refid_amd._lib import RefidHipError
    w = torch.randn(64, 48, 3, 3, device="cuda")
    xa, xb = torch.randn(1, 8, 32, 24, device="cuda"), torch.randn(1, 8, 32, 24, device="cuda")
    with pytest.raises(RefidHipError, match="multiple of 16"):
        ops.conv2d(xa, ops.pack_conv_weights_wino6(w, ops.ROLE_WINO_FWD, 64, 48), torch.empty(1, 8, 32, 64, device="cuda"),
                   kh=3, kw=3, pad=1, cout=64, cout_pad=64, in_b=xb, algo=5)
    with pytest.raises(RefidHipError):
        ops.pack_conv_weights_wino6(w, ops.ROLE_FWD, 64, 48)


@pytest.mark.parametrize("cfg", [(1, 16, 32, 32, 64), (1, 9, 24, 128, 64), (1, 8, 16, 256, 128)])
def test_winograd_dgrad(cfg):
    ops = _ops()
    N, H, W, Ci, Co = cfg
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    g = rnd(N, Co, H, W, seed=3)
    F.conv2d(x, w, None, 1, 1).backward(g)
    wd = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_WINO_DGRAD, 64, 8, 3, 3, Co, Ci)
    rp = -(-Ci // 64) * 64
    gd = nhwc(g)
    out = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(gd, wd, out, kh=3, kw=3, stride=1, pad=1, cout=Ci, cout_pad=rp, algo=1)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    if Ci >= 64:        # row-range issue (two-source convs): second half of the rows
        half = Ci // 2
        o2 = torch.empty(N, H, W, half, device="cuda")
        ops.conv2d(gd, wd, o2, kh=3, kw=3, stride=1, pad=1, cout=half, cout_pad=rp, co_base=half, algo=1)
        np.testing.assert_allclose(nchw(o2).numpy(), x.grad[:, half:].numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [
    (2, 16, 32, 64, 0, 64), (1, 24, 40, 64, 64, 64), (1, 8, 8, 256, 256, 256), (1, 5, 3, 64, 0, 128),
    (1, 7, 33, 32, 0, 96), (1, 16, 32, 32, 32, 32), (1, 8, 32, 32, 0, 3),
])
@pytest.mark.parametrize("algo", [1, 5])
def test_winograd_wgrad(cfg, algo):
    """algo 1: fp32 MFMA, 2x2 tiles; algo 5: fp32 MFMA, 2x4 tiles (wgrad_wino24.hip)."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    x = rnd(N, Ca + Cb, H, W, seed=1)
    w = rnd(Co, Ca + Cb, 3, 3, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, 1, 1)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda")
    db = torch.zeros(Co, device="cuda")
    gd = nhwc(g)
    if Co % 4:
        buf = torch.zeros(*gd.shape[:3], -(-Co // 4) * 4, device="cuda")
        buf[..., :Co] = gd
        gd = buf
    xa = nhwc(x[:, :Ca])
    xb = nhwc(x[:, Ca:]) if Cb else None
    for _ in range(2):
        ops.conv2d_wgrad(gd, xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=algo)
    scale = max(1.0, float(w.grad.abs().max()))
    np.testing.assert_allclose(dw.double().cpu().numpy() / 2, w.grad.numpy(), rtol=RTOL, atol=2 * ATOL * scale)
    np.testing.assert_allclose(db.double().cpu().numpy() / 2, b.grad.numpy(), rtol=RTOL, atol=2 * ATOL * scale)


@pytest.mark.parametrize("cfg", [(2, 64, 64, 64, 0, 64), (1, 48, 80, 64, 64, 64), (1, 32, 32, 128, 128, 128), (4, 32, 32, 32, 0, 32)])
def test_wgrad_f4_accuracy_class(cfg):
    """Winograd weight gradient over 2x4 tiles (algo 5: F(3,2) x F(3,4)) against the float64 gradient on O(1) activations
    (offset, like the LeakyReLU outputs it reads) and small output gradients: its deviation is in the fp32 class -- below 2e-5
    of the tensor's largest entry and within 10x of the 2x2-tile form's (algo 1); the bias gradient is a plain sum."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    x = rnd(N, Ca + Cb, H, W, seed=1)
    x = torch.where(x > -0.3, x, 0.1 * x) + 0.2
    w = rnd(Co, Ca + Cb, 3, 3, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, 1, 1)
    g = rnd(*y.shape, seed=4, scale=1e-3)
    y.backward(g)
    xa = nhwc(x[:, :Ca]); xb = nhwc(x[:, Ca:]) if Cb else None
    err = {}
    for algo in (1, 5):
        dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
        ops.conv2d_wgrad(nhwc(g), xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=algo)
        err[algo] = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
        assert float((db.double().cpu() - b.grad).abs().max() / b.grad.abs().max()) < 1e-5
    assert err[5] < 2e-5 and err[5] < 10 * max(err[1], 1e-6), err


def test_convop_sub_batched_input_gradient_keeps_the_gelu_mask(monkeypatch):
    """engine.ConvOp.dgrad issues a batch whose tensors reach 2 GiB in sub-batches: the GELU' mask (conv5's input gradient,
    fusion_modules.py:327-329 backward) must ride in every sub-call -- sub-batched == whole batch, bit for bit; and a conv whose
    input-gradient tile cannot apply GELU' refuses the request instead of applying the leaky-step mask."""
    from collections import OrderedDict
    from refid_amd import engine
    from refid_amd.engine import ConvOp, ParamArena
    from refid_amd._lib import RefidHipError
    co, ci, N, H, W = 128, 128, 4, 8, 16
    A = ParamArena(OrderedDict([("c.weight", (co, ci, 1, 1)), ("c.bias", (co,)), ("d.weight", (64, 64, 3, 3)), ("d.bias", (64,))]),
                   torch.device("cuda"))
    A.p("c.weight").copy_(rnd(co, ci, 1, 1, seed=1, scale=0.1).float())
    A.p("d.weight").copy_(rnd(64, 64, 3, 3, seed=2, scale=0.1).float())
    op = ConvOp(A, "c")
    op.repack()
    g = nhwc(rnd(N, co, H, W, seed=3))
    c4 = nhwc(rnd(N, ci, H, W, seed=4))
    whole = op.dgrad(g, mask=c4, gelu_mask=True)
    from refid_amd import ops
    plain = op.dgrad(g)
    want = ops.gelu_bwd(plain, c4)
    assert float((whole - want).abs().max()) <= 1e-6 * float(want.abs().max())
    monkeypatch.setattr(engine, "_LIM4", g[0].numel() * 2 - 1)              # two samples no longer fit "2 GiB"
    monkeypatch.setattr(engine, "_batch_step", lambda n, *t: 1)
    parts = op.dgrad(g, mask=c4, gelu_mask=True)
    assert torch.equal(parts, whole)
    op3 = ConvOp(A, "d")
    op3.repack()
    with pytest.raises(RefidHipError, match="gelu_mask"):
        op3.dgrad(nhwc(rnd(1, 64, 8, 16, seed=5)), mask=nhwc(rnd(1, 64, 8, 16, seed=6)), gelu_mask=True)


@pytest.mark.parametrize("ca,cb,group", [(48, 48, 1), (48, 48, 3), (40, 24, 1), (64, 64, 3)])
def test_convop_two_source_weight_gradient_any_split(ca, cb, group):
    """engine.ConvOp: a two-source 3x3 conv whose first source is not a multiple of 32 channels (the Winograd
    weight-gradient tile selects the source per 32-channel tile) falls back to the direct tile; grouped time steps and
    immediate launches give the same gradient as torch."""
    from collections import OrderedDict
    from refid_amd.engine import ConvOp, ParamArena
    co, ci, N, H, W, T = 64, ca + cb, 1, 16, 32, 3
    A = ParamArena(OrderedDict([("c.weight", (co, ci, 3, 3)), ("c.bias", (co,))]), torch.device("cuda"))
    w = rnd(co, ci, 3, 3, seed=1, scale=0.1).requires_grad_(True)
    bias = rnd(co, seed=2).requires_grad_(True)
    A.p("c.weight").copy_(w.detach().float()); A.p("c.bias").copy_(bias.detach().float())
    op = ConvOp(A, "c")
    op.w_group = group
    for t in range(T):
        x = rnd(N, ci, H, W, seed=10 + t)
        g = rnd(N, co, H, W, seed=20 + t)
        F.conv2d(x, w, bias, 1, 1).backward(g)
        op._wgrad(nhwc(g), nhwc(x[:, :ca]), nhwc(x[:, ca:]))
    op._finish_wgrad()
    torch.cuda.synchronize()
    np.testing.assert_allclose(A.g("c.weight").cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(A.g("c.bias").cpu().numpy(), bias.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("algo,cfg", [(0, (1, 16, 32, 64, 64, 64, 3)), (1, (1, 16, 32, 64, 64, 64, 3)),
                                      (5, (1, 16, 32, 64, 64, 64, 3)), (5, (2, 20, 36, 32, 32, 96, 3)),
                                      (0, (2, 8, 16, 128, 0, 128, 1)), (0, (1, 16, 32, 64, 0, 64, 4))])
def test_wgrad_slab_phases(algo, cfg):
    """Persistent slabs: overwrite (1), add (2, here with the second source missing as at the first
    recurrent step), reduce (3) == sum of the one-shot gradients."""
    ops = _ops()
    N, H, W, Ca, Cb, Co, k = cfg
    s, p = (2, 1) if k == 4 else (1, k // 2)
    Ho = (H + 2 * p - k) // s + 1
    x = rnd(N, Ca + Cb, H, W, seed=1)
    g1, g2 = rnd(N, Co, Ho, Ho * W // H, seed=2), rnd(N, Co, Ho, Ho * W // H, seed=3)
    xa = nhwc(x[:, :Ca]); xb = nhwc(x[:, Ca:]) if Cb else None
    ref = torch.zeros(Co, Ca + Cb, k, k, device="cuda"); rb = torch.zeros(Co, device="cuda")
    ops.conv2d_wgrad(nhwc(g1), xa, ref, kh=k, kw=k, stride=s, pad=p, in_b=xb, db=rb, algo=algo)
    ops.conv2d_wgrad(nhwc(g2), xa, ref, kh=k, kw=k, stride=s, pad=p, db=rb, algo=algo, i_total=Ca + Cb)
    dw = torch.zeros_like(ref); db = torch.zeros_like(rb)
    sl = ops.conv2d_wgrad(nhwc(g1), xa, dw, kh=k, kw=k, stride=s, pad=p, in_b=xb, db=db, algo=algo, phase=1,
                          i_total=Ca + Cb)
    ops.conv2d_wgrad(nhwc(g2), xa, dw, kh=k, kw=k, stride=s, pad=p, db=db, algo=algo, phase=2, slabs=sl,
                     i_total=Ca + Cb)
    assert float(dw.abs().max()) == 0.0                       # nothing reduced yet
    ops.conv2d_wgrad(nhwc(g2), xa, dw, kh=k, kw=k, stride=s, pad=p, db=db, algo=algo, phase=3, slabs=sl,
                     i_total=Ca + Cb)
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(dw.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), rb.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)


# ---------------------------------------------------------------------------------------------
# bf16-operand direct tile (algo=2): operands rounded to bf16 (RNE), products/accumulation in fp32
# ---------------------------------------------------------------------------------------------
def _bf(t):
    return t.float().bfloat16().double()


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co, k, s, p, mode
    (2, 16, 32, 32, 0, 64, 3, 1, 1, 0), (1, 24, 40, 64, 64, 64, 3, 1, 1, 0), (1, 8, 8, 128, 128, 128, 3, 1, 1, 0),
    (1, 12, 20, 64, 0, 256, 3, 1, 1, 0), (2, 16, 16, 32, 32, 32, 3, 1, 1, 0), (1, 16, 32, 32, 0, 3, 3, 1, 1, 0),
    (1, 16, 24, 4, 0, 32, 5, 1, 2, 0), (1, 16, 24, 28, 0, 32, 5, 1, 2, 0), (2, 8, 16, 64, 64, 64, 1, 1, 0, 0),
    (1, 16, 16, 32, 0, 64, 1, 1, 0, 0), (2, 16, 32, 64, 0, 64, 4, 2, 1, 0), (1, 16, 16, 32, 0, 64, 2, 2, 0, 0),
])
def test_bf16_tile_forward(cfg):
    ops = _ops()
    N, H, W, Ca, Cb, Co, k, s, p, mode = cfg
    Ci = Ca + Cb
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=1.0 / np.sqrt(Ci * k * k))
    b = rnd(Co, seed=3)
    r = None
    ref = lrelu(F.conv2d(_bf(x), _bf(w), b, s, p), 0.1)
    kc, bn = 2 * ops.conv_kc(k, k, s), ops.conv_bn(k, k, s, 0, Co)
    wp = ops.pack_conv_weights_bf16(w.float().cuda(), ops.ROLE_FWD, bn, kc, k, k, Co, Ci)
    Cop = -(-Co // 4) * 4
    outbuf = torch.zeros(N, ref.shape[2], ref.shape[3], Cop, device="cuda")
    out = outbuf[..., :Co]
    ops.conv2d(nhwc(x[:, :Ca]), wp, out, kh=k, kw=k, stride=s, pad=p, cout=Co, cout_pad=-(-Co // bn) * bn,
               in_b=nhwc(x[:, Ca:]) if Cb else None, bias=b.float().cuda(), slope_pre=0.1, algo=2)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


def test_bf16_tile_dgrad_and_special_modes():
    ops = _ops()
    # stride-1 dgrad
    N, H, W, Ci, Co = 1, 8, 24, 128, 64
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    g = rnd(N, Co, H, W, seed=3)
    F.conv2d(x, _bf(w), None, 1, 1).backward(_bf(g))
    wd = ops.pack_conv_weights_bf16(w.float().cuda(), ops.ROLE_DGRAD, 128, 16, 3, 3, Co, Ci)
    out = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(nhwc(g), wd, out, kh=3, kw=3, stride=1, pad=1, cout=Ci, cout_pad=128, algo=2)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    # conv_down input gradient (mode 2) and ConvTranspose2d forward (mode 1)
    C = 64
    x = rnd(1, C, 16, 32, seed=4).requires_grad_(True)
    w4 = rnd(C, C, 4, 4, seed=5, scale=0.1)
    y = F.conv2d(x, _bf(w4), None, 2, 1)
    g = rnd(*y.shape, seed=6)
    y.backward(_bf(g))
    kc = 2 * ops.conv_kc(4, 4, 2, 2)
    wd = ops.pack_conv_weights_bf16(w4.float().cuda(), ops.ROLE_DOWN_DGRAD, 64, kc, 4, 4, C, C)
    out = torch.empty(1, 16, 32, C, device="cuda")
    ops.conv2d(nhwc(g), wd, out, kh=4, kw=4, stride=2, pad=1, mode=2, cout=C, cout_pad=64, algo=2)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=RTOL, atol=ATOL)
    wt = rnd(128, 64, 2, 2, seed=7, scale=0.1)
    xi = rnd(1, 128, 8, 16, seed=8)
    bt = rnd(64, seed=9)
    yt = F.conv_transpose2d(_bf(xi), _bf(wt), bt, stride=2)
    kc = 2 * ops.conv_kc(1, 1, 1, 1)
    wp = ops.pack_conv_weights_bf16(wt.float().cuda(), ops.ROLE_CONVT, 128, kc, 2, 2, 64, 128)
    o = torch.empty(1, 16, 32, 64, device="cuda")
    ops.conv2d(nhwc(xi), wp, o, kh=1, kw=1, stride=1, pad=0, mode=1, cout=256, cout_pad=256, bias=bt.float().cuda(), algo=2)
    np.testing.assert_allclose(nchw(o).numpy(), yt.numpy(), rtol=RTOL, atol=ATOL)


# ---------------------------------------------------------------------------------------------
# register-operand pointwise tile (algo=3)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co
    (2, 8, 16, 64, 64, 64), (1, 16, 16, 32, 0, 64), (1, 8, 8, 128, 0, 128), (1, 8, 8, 64, 0, 32),
    (1, 5, 7, 256, 256, 256), (1, 9, 11, 64, 0, 128), (1, 6, 10, 128, 0, 64), (1, 4, 4, 16, 0, 16),
])
def test_pointwise_tile(cfg):
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    Ci = Ca + Cb
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 1, 1, seed=2, scale=1.0 / np.sqrt(Ci))
    b = rnd(Co, seed=3)
    r = rnd(N, Co, H, W, seed=4)
    m = rnd(N, Co, H, W, seed=5)
    ref = lrelu(lrelu(F.conv2d(x, w, b), 0.2) + r, 0.5) * torch.where(m > 0, 1.0, 0.3)
    wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_FWD, 32, 8, 1, 1, Co, Ci)
    out = torch.empty(N, H, W, Co, device="cuda")
    ops.conv2d(nhwc(x[:, :Ca]), wp, out, kh=1, kw=1, cout=Co, cout_pad=-(-Co // 32) * 32, algo=3,
               in_b=nhwc(x[:, Ca:]) if Cb else None, bias=b.float().cuda(), res=nhwc(r), mask=nhwc(m),
               slope_pre=0.2, slope_post=0.5, slope_mask=0.3)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    # input gradient = the same tile on the transposed packing, issued as a row range
    if Cb:
        xg = x.clone().requires_grad_(True)
        g = rnd(N, Co, H, W, seed=6)
        F.conv2d(xg, w, None).backward(g)
        wd = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_DGRAD, 32, 8, 1, 1, Co, Ci)
        o2 = torch.empty(N, H, W, Cb, device="cuda")
        ops.conv2d(nhwc(g), wd, o2, kh=1, kw=1, cout=Cb, cout_pad=-(-Ci // 32) * 32, co_base=Ca, algo=3)
        np.testing.assert_allclose(nchw(o2).numpy(), xg.grad[:, Ca:].numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Ca, Cb, Co
    (2, 8, 16, 64, 64, 64), (1, 16, 16, 32, 0, 64), (1, 8, 8, 128, 0, 128), (1, 5, 7, 256, 256, 256), (1, 9, 11, 64, 0, 128),
    (1, 6, 10, 128, 0, 64), (1, 7, 9, 16, 0, 48), (1, 4, 33, 48, 16, 80),
])
def test_pointwise_tile_six_products(cfg):
    """refid_conv2d algo 3 with mfma_terms = 6: the pointwise tile's products as six bf16 MFMAs on exactly split operands
    (weights: refid_pack_conv_weights_split's 1x1 layout).  Same contract and the same distance from the float64 result as
    the fp32 MFMA form -- also for operands of magnitude 1e-8 (bf16 has fp32's exponent range)."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    Ci = Ca + Cb
    for mag in (1.0, 1e-8):
        x = rnd(N, Ci, H, W, seed=1) * mag
        w = rnd(Co, Ci, 1, 1, seed=2, scale=1.0 / np.sqrt(Ci))
        b = rnd(Co, seed=3) * mag
        r = rnd(N, Co, H, W, seed=4) * mag
        m = rnd(N, Co, H, W, seed=5)
        ref = lrelu(lrelu(F.conv2d(x, w, b), 0.2) + r, 0.5) * torch.where(m > 0, 1.0, 0.3)
        kw_ = dict(kh=1, kw=1, cout=Co, cout_pad=-(-Co // 32) * 32, algo=3, in_b=nhwc(x[:, Ca:]) if Cb else None,
                   bias=b.float().cuda(), res=nhwc(r), mask=nhwc(m), slope_pre=0.2, slope_post=0.5, slope_mask=0.3)
        w6 = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_FWD, 32, 1, 1, Co, Ci, planes=3)
        o6 = torch.empty(N, H, W, Co, device="cuda")
        ops.conv2d(nhwc(x[:, :Ca]), w6, o6, terms=6, **kw_)
        wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_FWD, 32, 8, 1, 1, Co, Ci)
        o32 = torch.empty(N, H, W, Co, device="cuda")
        ops.conv2d(nhwc(x[:, :Ca]), wp, o32, **kw_)
        e6 = float((nchw(o6).double() - ref).abs().max()) / mag
        e32 = float((nchw(o32).double() - ref).abs().max()) / mag
        assert e6 <= 2.0 * e32 + 2e-7, (mag, e6, e32)
        assert e6 <= 5e-6, (mag, e6)
    # input gradient = the same tile on the transposed packing, issued as a row range
    if Cb > 32:
        xg = x.clone().requires_grad_(True)
        g = rnd(N, Co, H, W, seed=6)
        F.conv2d(xg, w, None).backward(g)
        wd = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_DGRAD, 32, 1, 1, Co, Ci, planes=3)
        o2 = torch.empty(N, H, W, Cb, device="cuda")
        ops.conv2d(nhwc(g), wd, o2, kh=1, kw=1, cout=Cb, cout_pad=-(-Ci // 32) * 32, co_base=Ca, algo=3, terms=6)
        assert float((nchw(o2).double() - xg.grad[:, Ca:]).abs().max()) <= 5e-6 * max(1.0, float(xg.grad.abs().max()))


def test_pointwise_tile_six_products_rejects_odd_chunks():
    ops = _ops()
    from refid_amd._lib import RefidHipError
    x = torch.zeros(1, 4, 4, 24, device="cuda")
    w6 = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(1, 4, 4, 64, device="cuda")
    with pytest.raises(RefidHipError, match="multiples of 16"):
        ops.conv2d(x, w6, out, kh=1, kw=1, cout=64, cout_pad=64, algo=3, terms=6)


@pytest.mark.parametrize("cfg", [(3, 16, 40, 2, 32, 5), (2, 9, 33, 2, 8, 5), (1, 8, 32, 3, 16, 3), (5, 12, 64, 4, 32, 5)])
def test_thin_input_wgrad(cfg):
    """Event-head style weight gradient: <= 4 (zero padded) input channels, K x K taps packed 8 per MFMA column tile;
    one-shot and persistent-slab phases."""
    ops = _ops()
    N, H, W, Ci, Co, k = cfg
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, 1, k // 2)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    x4 = torch.zeros(N, H, W, 4, device="cuda")
    x4[..., :Ci] = nhwc(x)
    gd = nhwc(g)
    dw = torch.zeros(Co, Ci, k, k, device="cuda"); db = torch.zeros(Co, device="cuda")
    for _ in range(2):
        ops.conv2d_wgrad(gd, x4, dw, kh=k, kw=k, stride=1, pad=k // 2, db=db, i_total=Ci)
    scale = max(1.0, float(w.grad.abs().max()))
    np.testing.assert_allclose(dw.double().cpu().numpy() / 2, w.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
    np.testing.assert_allclose(db.double().cpu().numpy() / 2, b.grad.numpy(), rtol=RTOL, atol=ATOL * scale)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    sl = ops.conv2d_wgrad(gd, x4, dw2, kh=k, kw=k, stride=1, pad=k // 2, db=db2, i_total=Ci, phase=1)
    ops.conv2d_wgrad(gd, x4, dw2, kh=k, kw=k, stride=1, pad=k // 2, db=db2, i_total=Ci, phase=2, slabs=sl)
    ops.conv2d_wgrad(gd, x4, dw2, kh=k, kw=k, stride=1, pad=k // 2, db=db2, i_total=Ci, phase=3, slabs=sl)
    np.testing.assert_allclose(dw2.cpu().numpy(), dw.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(db2.cpu().numpy(), db.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)


def test_winograd_splitk_two_streams_have_private_workspaces():
    """the C ABI owns no scratch -- refid_conv2d takes the caller's workspace
    (refid_conv_workspace_bytes), so split-K Winograd launches issued concurrently from two streams cannot share
    partial sums.  Results == the serial launches, bit for bit; no workspace = no split, same values within rounding."""
    import ctypes as C
    from refid_amd import _lib
    ops = _ops()
    N, H, W, Ci, Co = 1, 16, 32, 256, 128            # 1x2 pixel tiles x 2 channel tiles, 32 chunks -> splits 4-way
    probs = []
    for s in range(2):
        x = nhwc(rnd(N, Ci, H, W, seed=10 + s))
        w = rnd(Co, Ci, 3, 3, seed=20 + s, scale=0.05).float().cuda()
        wp = ops.pack_conv_weights(w, ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
        probs.append((x, wp))
    d = _lib.ConvDesc()
    d.n, d.h, d.w, d.ho, d.wo, d.c_a, d.cout, d.cout_pad, d.algo, d.split_k = N, H, W, H, W, Ci, Co, 128, 1, 1
    assert _lib.lib().refid_conv_workspace_bytes(C.byref(d)) == 4 * N * H * W * Co * 4
    d.split_k = 0
    assert _lib.lib().refid_conv_workspace_bytes(C.byref(d)) == 0

    def run(x, wp, reps):
        out = torch.empty(N, H, W, Co, device="cuda")
        for _ in range(reps):
            ops.conv2d(x, wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=128, algo=1)
        return out

    serial = [run(x, wp, 1).clone() for x, wp in probs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for _ in range(3):                                 # interleave many launches so the two streams really overlap
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = run(*probs[i], reps=20)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(outs[i], serial[i]), f"stream {i}: concurrent split-K result differs from the serial one"
    keys = [k for k in ops._ws_cache if k[2] == "conv"]
    assert len({k[1] for k in keys}) >= 3              # default stream + the two side streams: one buffer each
    old, ops.WINO_SPLIT = ops.WINO_SPLIT, 0
    try:
        nosplit = run(*probs[0], reps=1)
    finally:
        ops.WINO_SPLIT = old
    np.testing.assert_allclose(nosplit.cpu().numpy(), serial[0].cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", ["down_fwd", "down_dgrad"])
def test_direct_tile_splitk_small_grids(kind):
    """The 4x4 / stride-2 tiles (conv_down forward, its input gradient) are long-K, few-tile launches at small batch
    (B=1: 32 workgroups for 256 channels at 64x64 -> 32x32): split-K over the input-channel chunks into the caller's
    workspace + the shared finishing pass.  Split == unsplit to rounding, == torch."""
    import ctypes as C
    from refid_amd import _lib
    ops = _ops()
    N, H, W, Cc = 1, 64, 64, 256
    w = rnd(Cc, Cc, 4, 4, seed=2, scale=1.0 / np.sqrt(Cc * 16))
    if kind == "down_fwd":
        x = rnd(N, Cc, H, W, seed=1)
        ref = F.conv2d(x, w, None, 2, 1)
        wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_FWD, 64, 8, 4, 4, Cc, Cc)
        src = nhwc(x)
        kw = dict(kh=4, kw=4, stride=2, pad=1, cout=Cc, cout_pad=256)
        oshape = (N, H // 2, W // 2, Cc)
        extra = {}
    else:
        g = rnd(N, Cc, H // 2, W // 2, seed=1)
        ref = F.conv_transpose2d(g, w, None, 2, 1)
        kc, bn = ops.conv_kc(4, 4, 2, 2), ops.conv_bn(4, 4, 2, 2, 128)
        wp = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_DOWN_DGRAD, bn, kc, 4, 4, Cc, Cc)
        src = nhwc(g)
        kw = dict(kh=4, kw=4, stride=2, pad=1, mode=2, cout=Cc, cout_pad=-(-Cc // bn) * bn)
        oshape = (N, H, W, Cc)
        m = rnd(N, Cc, H, W, seed=5)
        ref = ref * torch.where(m > 0, 1.0, 0.2)
        extra = dict(mask=nhwc(m), slope_mask=0.2)
    outs = []
    for split in (ops.WINO_SPLIT or 2, 0):
        old, ops.WINO_SPLIT = ops.WINO_SPLIT, split
        try:
            out = torch.empty(*oshape, device="cuda")
            ops.conv2d(src, wp, out, **kw, **extra)
            outs.append(out)
        finally:
            ops.WINO_SPLIT = old
    d = _lib.ConvDesc()
    d.n, d.h, d.w = src.shape[0], src.shape[1], src.shape[2]
    d.ho, d.wo = (oshape[1], oshape[2]) if kind == "down_fwd" else (d.h, d.w)
    d.c_a, d.cout, d.cout_pad, d.kh, d.kw, d.stride, d.pad = Cc, Cc, kw["cout_pad"], 4, 4, 2, 1
    d.mode, d.algo, d.split_k = kw.get("mode", 0), 0, 2
    assert _lib.lib().refid_conv_workspace_bytes(C.byref(d)) > 0          # this shape IS split
    np.testing.assert_allclose(outs[0].cpu().numpy(), outs[1].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nchw(outs[0]).numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("cfg", [(2, 16, 32, 64, 0, 64), (1, 24, 40, 64, 64, 64), (1, 9, 35, 128, 0, 64), (1, 8, 8, 256, 256, 256),
                                 (1, 5, 3, 64, 0, 128)])
def test_bf16_weight_gradient_tile(cfg):
    """refid_conv2d_wgrad algo 2 (csrc/wgrad_bf16.hip): bf16 matrix-core operands, fp32 accumulation.  Against a float64
    reference on the bf16-ROUNDED operands the result is exact to fp32 summation error; against the unrounded fp32
    reference it carries the operands' 2^-9 rounding (the config-3 criterion is PSNR / loss-level)."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    x = rnd(N, Ca + Cb, H, W, seed=1)
    g = rnd(N, Co, H, W, seed=4)
    w = rnd(Co, Ca + Cb, 3, 3, seed=2).requires_grad_(True)
    b = rnd(Co, seed=3).requires_grad_(True)
    F.conv2d(x, w, b, 1, 1).backward(g)
    xr, gr = x.float().bfloat16().double(), g.float().bfloat16().double()
    w2 = rnd(Co, Ca + Cb, 3, 3, seed=2).requires_grad_(True)
    F.conv2d(xr, w2, None, 1, 1).backward(gr)
    dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
    xa = nhwc(x[:, :Ca]); xb = nhwc(x[:, Ca:]) if Cb else None
    # two calls accumulate (weights shared over the T steps); persistent-slab phases as for the other tiles
    ops.conv2d_wgrad(nhwc(g), xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=2)
    sl = ops.conv2d_wgrad(nhwc(g), xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=2, phase=1, i_total=Ca + Cb)
    ops.conv2d_wgrad(nhwc(g), xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=2, phase=2, slabs=sl, i_total=Ca + Cb)
    ops.conv2d_wgrad(nhwc(g), xa, dw, kh=3, kw=3, stride=1, pad=1, in_b=xb, db=db, algo=2, phase=3, slabs=sl, i_total=Ca + Cb)
    got = dw.double().cpu() / 3                      # one-shot + (overwrite, add) reduced once = 3 contributions
    scale = float(w2.grad.abs().max())
    assert float((got - w2.grad).abs().max()) <= 2e-5 * scale, "bf16 tile vs float64 on the rounded operands"
    assert float((got - w.grad).abs().max()) <= 2e-2 * float(w.grad.abs().max())
    # bias gradient sums the fp32 (unrounded) gradient
    np.testing.assert_allclose(db.double().cpu().numpy() / 3, b.grad.numpy(), rtol=1e-4, atol=1e-4 * float(b.grad.abs().max()))
    # first recurrent step: the second source does not exist yet, the slab geometry already covers it (i_total)
    if Cb:
        dw1 = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda")
        s1 = ops.conv2d_wgrad(nhwc(g), xa, dw1, kh=3, kw=3, stride=1, pad=1, algo=2, phase=1, i_total=Ca + Cb)
        ops.conv2d_wgrad(nhwc(g), xa, dw1, kh=3, kw=3, stride=1, pad=1, algo=2, phase=3, slabs=s1, i_total=Ca + Cb)
        assert float((dw1[:, :Ca].double().cpu() - w2.grad[:, :Ca]).abs().max()) <= 2e-5 * scale
        assert float(dw1[:, Ca:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [(2, 16, 32, 64, 0, 64), (1, 24, 40, 64, 64, 64), (1, 9, 33, 64, 64, 128)])
@pytest.mark.parametrize("algo", [1, 5])
def test_winograd_wgrad_grouped_time_steps(cfg, algo):
    """refid_wgrad_desc.groups: the Winograd weight gradients of several time steps of one conv in ONE launch (the weights
    are shared over T) == the same calls one by one -- persistent-slab phases included."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    steps = []
    for t in range(4):
        x = rnd(N, Ca + Cb, H, W, seed=10 + t)
        g = rnd(N, Co, H, W, seed=20 + t)
        steps.append((nhwc(g), nhwc(x[:, :Ca]), nhwc(x[:, Ca:]) if Cb else None))
    kw = dict(kh=3, kw=3, stride=1, pad=1, algo=algo, i_total=Ca + Cb)

    def run(grouping):
        dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
        sl, first = None, True
        for grp in grouping:
            (g, a, b), more = steps[grp[0]], [steps[i] for i in grp[1:]]
            sl = ops.conv2d_wgrad(g, a, dw, in_b=b, db=db, phase=1 if first else 2, slabs=sl, more=more, **kw)
            first = False
        g, a, b = steps[0]
        ops.conv2d_wgrad(g, a, dw, in_b=b, db=db, phase=3, slabs=sl, **kw)
        return dw, db

    dw1, db1 = run([[0], [1], [2], [3]])
    for grouping in ([[0, 1, 2, 3]], [[0, 1, 2], [3]], [[0], [1, 2, 3]]):
        dw2, db2 = run(grouping)
        scale = float(dw1.abs().max())
        assert float((dw2 - dw1).abs().max()) <= 1e-5 * scale, grouping
        np.testing.assert_allclose(db2.cpu().numpy(), db1.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(db1.abs().max()))
    # one-shot (phase 0) grouping
    dw3 = torch.zeros_like(dw1); db3 = torch.zeros_like(db1)
    ops.conv2d_wgrad(*steps[0][:2], dw3, in_b=steps[0][2], db=db3, phase=0, more=steps[1:], **kw)
    assert float((dw3 - dw1).abs().max()) <= 1e-5 * float(dw1.abs().max())


@pytest.mark.parametrize("algo,k,stride,cfg", [(0, 3, 1, (2, 16, 32, 32, 0, 32)), (0, 4, 2, (1, 32, 64, 64, 0, 64)),
                                               (2, 3, 1, (1, 16, 32, 64, 64, 64)), (0, 3, 1, (1, 24, 40, 32, 32, 32))])
def test_direct_and_bf16_wgrad_grouped_time_steps(algo, k, stride, cfg):
    """The same grouping for the direct fp32 tiles (3x3 narrow, 4x4 stride 2) and the bf16 tile: 8 time steps in one or
    three launches == one launch per step."""
    ops = _ops()
    N, H, W, Ca, Cb, Co = cfg
    pad = 1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    steps = []
    for t in range(8):
        x = rnd(N, Ca + Cb, H, W, seed=10 + t)
        g = rnd(N, Co, Ho, Wo, seed=30 + t)
        steps.append((nhwc(g), nhwc(x[:, :Ca]), nhwc(x[:, Ca:]) if Cb else None))
    kw = dict(kh=k, kw=k, stride=stride, pad=pad, algo=algo, i_total=Ca + Cb)

    def run(grouping):
        dw = torch.zeros(Co, Ca + Cb, k, k, device="cuda"); db = torch.zeros(Co, device="cuda")
        sl, first = None, True
        for grp in grouping:
            (g, a, b), more = steps[grp[0]], [steps[i] for i in grp[1:]]
            sl = ops.conv2d_wgrad(g, a, dw, in_b=b, db=db, phase=1 if first else 2, slabs=sl, more=more, **kw)
            first = False
        g, a, b = steps[0]
        ops.conv2d_wgrad(g, a, dw, in_b=b, db=db, phase=3, slabs=sl, **kw)
        return dw, db

    dw1, db1 = run([[i] for i in range(8)])
    for grouping in ([list(range(8))], [[0, 1, 2], [3, 4, 5, 6], [7]]):
        dw2, db2 = run(grouping)
        assert float((dw2 - dw1).abs().max()) <= 1e-5 * float(dw1.abs().max()), grouping
        np.testing.assert_allclose(db2.cpu().numpy(), db1.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(db1.abs().max()))


# ---------------------------------------------------------------------------------------------
# split-bf16 direct 3x3 tile (algo=4): fp32 operands as sums of bf16 numbers on the bf16 matrix cores
# ---------------------------------------------------------------------------------------------
def _split_planes(t, planes):
    """What the tile multiplies with: the sum of the first `planes` bf16 numbers of every fp32 operand."""
    t = t.float()
    acc = torch.zeros_like(t)
    r = t.clone()
    for _ in range(planes):
        h = r.bfloat16().float()
        acc += h
        r = r - h
    return acc.double()


def run_split(N, H, W, Ca, Cb, Co, terms, bias=True, res=False, slope_pre=1.0, slope_post=1.0, mask=False, tol=None):
    ops = _ops()
    planes = {6: 3, 3: 2, 1: 1}[terms]
    Ci = Ca + Cb
    xa = rnd(N, Ca, H, W, seed=1)
    xb = rnd(N, Cb, H, W, seed=2) if Cb else None
    w = rnd(Co, Ci, 3, 3, seed=3, scale=1.0 / np.sqrt(Ci * 9))
    b = rnd(Co, seed=4) if bias else None
    x = torch.cat([xa, xb], 1) if Cb else xa
    # x6: the fp32 operands themselves; x1: what a bf16 conv multiplies (the error of the dropped cross terms is
    # covered by the tolerance for x3)
    xr, wr = (x, w) if terms != 1 else (_split_planes(x, 1), _split_planes(w, 1))
    ref = lrelu(F.conv2d(xr, wr, b, 1, 1), slope_pre)
    r = rnd(N, Co, H, W, seed=5) if res else None
    if res:
        ref = ref + r
    ref = lrelu(ref, slope_post)
    m = rnd(N, Co, H, W, seed=6) if mask else None
    if mask:
        ref = ref * torch.where(m > 0, 1.0, 0.3)
    bn = ops.conv_bn(3, 3, 1, 0, Co)
    wp = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_FWD, bn, 3, 3, Co, Ci, planes=planes)
    Cop = -(-Co // 4) * 4
    outbuf = torch.full((N, H, W, Cop), 7.0, device="cuda")
    out = outbuf[..., :Co]
    ops.conv2d(nhwc(xa), wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=-(-Co // bn) * bn, algo=4, terms=terms,
               in_b=nhwc(xb) if Cb else None, bias=b.float().cuda() if bias else None,
               res=nhwc(r) if res else None, mask=nhwc(m) if mask else None,
               slope_pre=slope_pre, slope_post=slope_post, slope_mask=0.3)
    rtol, atol = tol or (RTOL, ATOL)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=rtol, atol=atol)
    if Cop != Co:
        assert float(outbuf[..., Co:].min()) == 7.0


@pytest.mark.parametrize("terms", [6, 3, 1])
@pytest.mark.parametrize("cfg", [
    (2, 16, 32, 32, 0, 64), (1, 24, 40, 64, 64, 64), (1, 8, 8, 128, 128, 128), (1, 12, 20, 64, 0, 256),
    (2, 16, 16, 32, 32, 32), (1, 5, 3, 64, 0, 64), (1, 7, 33, 32, 0, 96), (1, 16, 32, 32, 0, 3), (1, 9, 70, 8, 0, 16),
    (3, 64, 64, 64, 0, 64),
])
def test_split_tile_forward_geometries(cfg, terms):
    # six products: the fp32 tolerance of this file; three: 2^-16 per product on O(1) sums
    run_split(*cfg, terms, tol={6: (RTOL, ATOL), 3: (2e-4, 6e-5), 1: (RTOL, ATOL)}[terms])


@pytest.mark.parametrize("terms", [6, 1])
def test_split_tile_fused_epilogues(terms):
    run_split(1, 16, 32, 64, 0, 64, terms, slope_pre=0.04)
    run_split(1, 16, 32, 64, 0, 64, terms, res=True)
    run_split(1, 8, 32, 128, 0, 128, terms, res=True, slope_post=0.0)
    run_split(1, 8, 32, 64, 64, 64, terms, slope_pre=0.1)
    run_split(1, 8, 32, 64, 0, 64, terms, bias=False, res=True, mask=True)
    run_split(1, 40, 32, 32, 0, 32, terms, bias=False, res=True, mask=True)


@pytest.mark.parametrize("terms", [6, 3])
@pytest.mark.parametrize("cfg", [(1, 16, 32, 32, 64), (1, 9, 24, 128, 64), (1, 8, 16, 256, 128)])
def test_split_tile_dgrad(cfg, terms):
    ops = _ops()
    N, H, W, Ci, Co = cfg
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    g = rnd(N, Co, H, W, seed=3)
    F.conv2d(x, w, None, 1, 1).backward(g)
    bn = ops.conv_bn(3, 3, 1, 0, min(Ci, 128))
    wd = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_DGRAD, bn, 3, 3, Co, Ci, planes={6: 3, 3: 2}[terms])
    rp = -(-Ci // bn) * bn
    gd = nhwc(g)
    rtol, atol = (RTOL, ATOL) if terms == 6 else (2e-4, 2e-4)
    out = torch.empty(N, H, W, Ci, device="cuda")
    ops.conv2d(gd, wd, out, kh=3, kw=3, stride=1, pad=1, cout=Ci, cout_pad=rp, algo=4, terms=terms)
    np.testing.assert_allclose(nchw(out).numpy(), x.grad.numpy(), rtol=rtol, atol=atol)
    if Ci >= 64:        # row-range issue (two-source convs): second half of the rows
        half = Ci // 2
        o2 = torch.empty(N, H, W, half, device="cuda")
        ops.conv2d(gd, wd, o2, kh=3, kw=3, stride=1, pad=1, cout=half, cout_pad=rp, co_base=half, algo=4, terms=terms)
        np.testing.assert_allclose(nchw(o2).numpy(), x.grad[:, half:].numpy(), rtol=rtol, atol=atol)


def test_split_tile_accuracy_classes():
    """Largest deviation from the fp64 convolution, K = 9 x 256: six products sit in the fp32 class (with the Winograd
    tile, both a few 1e-6 on O(1) sums), three products at ~3e-5, one product (plain bf16 operands) at ~1e-2."""
    ops = _ops()
    N, H, W, Ci, Co = 1, 32, 64, 256, 64
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=3.0 / np.sqrt(Ci * 9))
    ref = F.conv2d(x, w, None, 1, 1)
    xd = nhwc(x)
    errs = {}
    for terms in (6, 3, 1):
        wp = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_FWD, 64, 3, 3, Co, Ci, planes={6: 3, 3: 2, 1: 1}[terms])
        out = torch.empty(N, H, W, Co, device="cuda")
        ops.conv2d(xd, wp, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=64, algo=4, terms=terms)
        errs[terms] = float((nchw(out) - ref).abs().max())
    ww = ops.pack_conv_weights(w.float().cuda(), ops.ROLE_WINO_FWD, 64, 8, 3, 3, Co, Ci)
    out = torch.empty(N, H, W, Co, device="cuda")
    ops.conv2d(xd, ww, out, kh=3, kw=3, stride=1, pad=1, cout=Co, cout_pad=64, algo=1)
    errs["wino"] = float((nchw(out) - ref).abs().max())
    scale = float(ref.abs().max())
    assert errs[6] < 4e-6 * scale and errs["wino"] < 4e-6 * scale, errs
    assert errs[6] < errs[3] < errs[1], errs
    assert errs[3] < 4e-5 * scale and errs[1] > 1e-3 * scale, errs


def test_split_tile_rejects_bad_arguments():
    ops = _ops()
    from refid_amd._lib import RefidHipError
    w = torch.randn(64, 36, 3, 3, device="cuda")
    x = torch.randn(1, 8, 32, 36, device="cuda")
    out = torch.empty(1, 8, 32, 64, device="cuda")
    wp = ops.pack_conv_weights_split(w, ops.ROLE_FWD, 64, 3, 3, 64, 36, planes=3)
    with pytest.raises(RefidHipError, match="multiples of 8"):
        ops.conv2d(x, wp, out, kh=3, kw=3, stride=1, pad=1, cout=64, cout_pad=64, algo=4)
    x = torch.randn(1, 8, 32, 32, device="cuda")
    with pytest.raises(RefidHipError, match="1, 3 or 6"):
        ops.conv2d(x, wp, out, kh=3, kw=3, stride=1, pad=1, cout=64, cout_pad=64, algo=4, terms=19)     # (the fp16 form: stride-2 modes only)
    with pytest.raises(RefidHipError, match="1, 3 or 6"):
        ops.conv2d(x, wp, out, kh=3, kw=3, stride=1, pad=1, cout=64, cout_pad=64, algo=4, terms=4)
    with pytest.raises(RefidHipError):
        ops.pack_conv_weights_split(torch.randn(64, 32, 5, 5, device="cuda"), ops.ROLE_FWD, 64, 5, 5, 64, 32, planes=3)


@pytest.mark.parametrize("terms", [6, 3, 1, 19])
@pytest.mark.parametrize("cfg", [(2, 16, 64, 64, 64), (1, 24, 40, 128, 128), (1, 8, 8, 256, 256), (1, 18, 66, 32, 48), (3, 64, 64, 64, 64),
                                 (1, 10, 6, 16, 16)])
def test_split_tile_conv_down_forward(cfg, terms):
    """4x4 / stride 2 / pad 1 (`conv_down`, recurrent_sub_modules.py:12-14) on the split tile: a 2x2 conv over 2x2 input
    blocks that start at odd coordinates; bias / LeakyReLU / residual epilogue as everywhere."""
    ops = _ops()
    N, H, W, Ci, Co = cfg
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 4, 4, seed=2, scale=1.0 / np.sqrt(Ci * 16))
    b = rnd(Co, seed=3)
    xr, wr = (x, w) if terms != 1 else (_split_planes(x, 1), _split_planes(w, 1))
    y = lrelu(F.conv2d(xr, wr, b, 2, 1), 0.2)
    r = rnd(*y.shape, seed=4)
    ref = y + r
    bn = ops.conv_bn(4, 4, 2, 0, Co)
    wp = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_FWD, bn, 4, 4, Co, Ci, planes={6: 3, 3: 2, 1: 1, 19: 2}[terms],
                                     f16=terms == 19)
    out = torch.full((N, H // 2, W // 2, Co), 7.0, device="cuda")
    ops.conv2d(nhwc(x), wp, out, kh=4, kw=4, stride=2, pad=1, cout=Co, cout_pad=-(-Co // bn) * bn, bias=b.float().cuda(),
               res=nhwc(r), slope_pre=0.2, algo=4, terms=terms)
    rtol, atol = (RTOL, ATOL) if terms != 3 else (2e-4, 6e-5)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("terms", [6, 3, 1, 19])
@pytest.mark.parametrize("cfg", [(2, 16, 32, 64), (1, 24, 40, 128), (1, 8, 8, 256), (1, 36, 68, 32), (2, 64, 64, 64),
                                 (1, 16, 32, 24), (1, 8, 16, 40)])       # 24 / 40 channels: K % 16 == 8, half-empty last stage
def test_split_tile_conv_down_dgrad(cfg, terms):
    """Input gradient of conv_down: four output-parity classes of 2x2-tap convs over the output gradient (+ the fused
    residual / derivative-mask epilogue BPTT uses)."""
    ops = _ops()
    N, H, W, C = cfg
    x = rnd(N, C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, C, 4, 4, seed=2, scale=0.1)
    g = rnd(N, C, H // 2, W // 2, seed=3)
    wr, gr = (w, g) if terms != 1 else (_split_planes(w, 1), _split_planes(g, 1))
    F.conv2d(x, wr, None, 2, 1).backward(gr)
    r = rnd(N, C, H, W, seed=4)
    m = rnd(N, C, H, W, seed=5)
    ref = (x.grad + r) * torch.where(m > 0, 1.0, 0.3)
    bn = ops.conv_bn(4, 4, 2, 2, C)
    wp = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_DOWN_DGRAD, bn, 4, 4, C, C, planes={6: 3, 3: 2, 1: 1, 19: 2}[terms],
                                     f16=terms == 19)
    out = torch.empty(N, H, W, C, device="cuda")
    ops.conv2d(nhwc(g), wp, out, kh=4, kw=4, stride=2, pad=1, mode=2, cout=C, cout_pad=-(-C // bn) * bn, res=nhwc(r), mask=nhwc(m),
               slope_mask=0.3, algo=4, terms=terms)
    rtol, atol = (RTOL, ATOL) if terms != 3 else (2e-4, 2e-4)
    np.testing.assert_allclose(nchw(out).numpy(), ref.numpy(), rtol=rtol, atol=atol)


def test_split_tile_f16_accuracy_class_and_dynamic_range():
    """conv_down on three fp16 products (mfma_terms 19): two fp16 planes carry 22 bits; fp16's range is bridged by exact
    power-of-two scales -- the weights per packing, the activations per workgroup and online along K.  (a) O(1) data, K = 16 x
    256: within the six-bf16-product form's distance from float64 (both ~1e-6 of scale), forward and input gradient; (b) the
    same relative error at magnitudes 1e-8 / 1e+6 and for weights of 1e-6 / 1e+3; (c) channels that GROW by 2^40 along K (a
    rescale of the accumulators at every stage) and fall again; (d) image regions of different magnitude in different
    workgroups (the scale is per workgroup: 8 x 32 output pixels) keep the error relative to their own scale."""
    ops = _ops()
    N, H, W, C = 1, 64, 128, 256
    x = rnd(N, C, H, W, seed=1)
    w = rnd(C, C, 4, 4, seed=2, scale=3.0 / np.sqrt(C * 16))
    bn = ops.conv_bn(4, 4, 2, 0, C)
    bnd = ops.conv_bn(4, 4, 2, 2, C)

    def fwd(xx, ww, terms):
        wp = ops.pack_conv_weights_split(ww.float().cuda(), ops.ROLE_FWD, bn, 4, 4, C, C, planes=3 if terms == 6 else 2, f16=terms == 19)
        out = torch.empty(N, H // 2, W // 2, C, device="cuda")
        ops.conv2d(nhwc(xx), wp, out, kh=4, kw=4, stride=2, pad=1, cout=C, cout_pad=-(-C // bn) * bn, algo=4, terms=terms)
        return nchw(out)

    ref = F.conv2d(x, w, None, 2, 1)
    scale = float(ref.abs().max())
    e19 = float((fwd(x, w, 19) - ref).abs().max())
    e6 = float((fwd(x, w, 6) - ref).abs().max())
    assert e19 < 4e-6 * scale and e19 < 2.0 * e6 + 1e-7 * scale, (e19, e6, scale)                     # (a)
    g = rnd(N, C, H // 2, W // 2, seed=3)
    xg = x.clone().requires_grad_(True)
    F.conv2d(xg, w, None, 2, 1).backward(g)
    wd = ops.pack_conv_weights_split(w.float().cuda(), ops.ROLE_DOWN_DGRAD, bnd, 4, 4, C, C, planes=2, f16=True)
    od = torch.empty(N, H, W, C, device="cuda")
    ops.conv2d(nhwc(g), wd, od, kh=4, kw=4, stride=2, pad=1, mode=2, cout=C, cout_pad=-(-C // bnd) * bnd, algo=4, terms=19)
    assert float((nchw(od) - xg.grad).abs().max()) < 4e-6 * float(xg.grad.abs().max())
    for sx, sw in ((1e-8, 1.0), (1e6, 1.0), (1.0, 1e-6), (1.0, 1e3)):                                   # (b)
        xs, ws = (x * sx).float().double(), (w * sw).float().double()
        r = F.conv2d(xs, ws, None, 2, 1)
        assert float((fwd(xs, ws, 19) - r).abs().max()) < 4e-6 * float(r.abs().max()), (sx, sw)
    ramp = torch.tensor([2.0 ** (10 * min(c, 9 - c)) if c < 10 else 1.0 for c in range(C // 16)], dtype=torch.float64)   # (c)
    xc = (x * ramp.repeat_interleave(16).view(1, C, 1, 1)).float().double()
    r = F.conv2d(xc, w.float().double(), None, 2, 1)
    assert float((fwd(xc, w, 19) - r).abs().max()) < 4e-6 * float(r.abs().max())
    row = torch.tensor([2.0 ** (((y // 16) - 2) * 10) for y in range(H)], dtype=torch.float64)          # (d): bands of 16 rows
    xt = (x * row.view(1, 1, H, 1)).float().double()
    r = F.conv2d(xt, w.float().double(), None, 2, 1)
    got = fwd(xt, w, 19)
    for y0 in range(0, H // 2, 8):                             # a workgroup's 8 output rows = 16 input rows = one band (+ a halo row)
        near = r[:, :, max(0, y0 - 8):y0 + 16].abs().max()
        assert float((got[:, :, y0:y0 + 8] - r[:, :, y0:y0 + 8]).abs().max()) <= 4e-6 * float(near), y0


@pytest.mark.parametrize("case", [
    # (kind, co, ci, k, N, H, W, two_source, what)
    ("conv", 64, 64, 3, 2, 20, 36, False, "wino6 64-channel tile, ragged"),
    ("conv", 32, 64, 3, 1, 16, 32, True, "wino6 32-channel tile, two sources"),
    ("conv", 256, 256, 3, 1, 8, 8, False, "wino6 split-K: the finishing pass writes both outputs"),
    ("conv", 16, 16, 3, 1, 12, 20, False, "fp32 Winograd tile"),
    ("conv", 3, 32, 3, 1, 16, 16, False, "direct 3x3 tile, 3 output channels (scalar epilogue path)"),
    ("down", 64, 64, 4, 8, 32, 64, False, "conv_down on the split tile"),
    ("down", 64, 64, 4, 1, 16, 16, False, "conv_down on the fp32 tile (small grid, split-K)"),
    ("convT", 32, 64, 2, 1, 8, 16, False, "ConvTranspose: forward (pixel shuffle store) and its 2x2/s2 input gradient"),
    ("conv", 64, 128, 1, 1, 16, 16, True, "pointwise tile: no second output in the kernel -> add kernel fallback"),
])
def test_second_output_is_out_plus_add(case):
    """refid_conv_desc.out2 = out + add2 (ConvOp.fwd / dgrad `plus=`): the skip sums of arch:16-17,199-203,211 and their BPTT
    counterparts leave with the producing tile.  For every tile family: `out` is bit-identical to the call without a second
    output, and out2 == out + plus exactly (one fp32 add per element, as the add kernel does)."""
    from refid_amd import engine as E, ops
    kind, co, ci, k, N, H, W, two, _ = case
    name = "t"
    shapes = {"t.weight": (ci, co, 2, 2) if kind == "convT" else (co, ci, k, k)}
    if kind != "down":
        shapes["t.bias"] = (co,)
    arena = E.ParamArena(shapes, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(3)
    arena.flat_p.copy_(torch.randn(arena.total, device="cuda", generator=g) * 0.1)
    op = E.ConvOp(arena, name, kind=kind)
    op.repack()
    ca = ci // 2 if two else ci
    a = torch.randn(N, H, W, ca, device="cuda", generator=g)
    b = torch.randn(N, H, W, ci - ca, device="cuda", generator=g) if two else None
    ref = op.fwd(a, b, slope_pre=0.1)
    plus = torch.randn(ref.shape, device="cuda", generator=g)[..., :ref.shape[3]]
    if ref.shape[3] % 4:                                          # channel-padded output: views of padded buffers
        pad = torch.randn(*ref.shape[:3], E._pad4(ref.shape[3]), device="cuda", generator=g)
        plus = pad[..., :ref.shape[3]]
    if plus.is_contiguous() or ref.shape[3] % 4 == 0:
        out, o2 = op.fwd(a, b, slope_pre=0.1, plus=plus)
        assert torch.equal(out, ref)
        assert torch.equal(o2, ref + plus)
    # input gradient with residual + mask + second output
    gout = torch.randn(ref.shape, device="cuda", generator=g).contiguous() if ref.shape[3] % 4 == 0 else None
    if gout is not None and (kind != "conv" or not two):
        gin_ref = op.dgrad(gout)
        r = torch.randn_like(gin_ref)
        m = torch.randn_like(gin_ref)
        p2 = torch.randn_like(gin_ref)
        want = op.dgrad(gout, res=r, mask=m, slope_mask=0.2)
        got, got2 = op.dgrad(gout, res=r, mask=m, slope_mask=0.2, plus=p2)
        assert torch.equal(got, want)
        assert torch.equal(got2, want + p2)


def test_winograd_wgrad_eight_wave_form_subprocess():
    """REFID_WGRAD_WINO_IW=2 (an experiment that measured 0-4 % slower: two input-channel tiles per 8-wave workgroup sharing one
    staged gradient tile) must give the 4-wave form's weight gradient bit for bit -- a wave's arithmetic is unchanged -- and the
    bias gradient within rounding (its partials are summed in a different fixed order).  The switch is read once per process."""
    import subprocess, sys, os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from refid_amd import ops
torch.manual_seed(3)
out = {}
for (N, H, W, Ca, Cb, Co) in ((2, 12, 40, 64, 0, 64), (1, 8, 32, 64, 64, 128), (1, 9, 33, 32, 32, 64)):
    a = torch.randn(N, H, W, Ca, device="cuda"); b = torch.randn(N, H, W, Cb, device="cuda") if Cb else None
    g = torch.randn(N, H, W, Co, device="cuda")
    dw = torch.zeros(Co, Ca + Cb, 3, 3, device="cuda"); db = torch.zeros(Co, device="cuda")
    ops.conv2d_wgrad(g, a, dw, kh=3, kw=3, pad=1, in_b=b, db=db, algo=1)
    out[(N, H, W, Ca, Cb, Co)] = (dw.cpu(), db.cpu())
torch.save(out, sys.argv[1])
''' % ROOT
    res = {}
    for iw in ("1", "2"):
        path = f"/tmp/refid_wgrad_iw{iw}_{os.getpid()}.pt"
        env = dict(os.environ, REFID_WGRAD_WINO_IW=iw)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[iw] = torch.load(path)
        os.remove(path)
    for k in res["1"]:
        assert torch.equal(res["1"][k][0], res["2"][k][0]), k
        np.testing.assert_allclose(res["2"][k][1].numpy(), res["1"][k][1].numpy(), rtol=1e-5, atol=1e-5)
