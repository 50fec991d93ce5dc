"""GPU parity of the non-GEMM kernels (EGACA pieces, LayerNorm2d, loss, optimizer) against
torch CPU fp64 formulas of the reference (fusion_modules.py:97-134, 290-333; losses.py;
torch.optim.AdamW / clip_grad_norm_)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


def _ops():
    from refid_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def dev(t):
    return t.float().cuda().contiguous()


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a.double().cpu().numpy(), b.double().cpu().numpy(), rtol=rtol, atol=atol)


def ln_ref(x, w, b, eps=1e-6):          # NHWC fp64
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return w * ((x - mu) / (var + eps).sqrt()) + b


@pytest.mark.parametrize("C", [16, 64, 128])
def test_layernorm2d_fwd_bwd(C):
    ops = _ops()
    x = rnd(2, 6, 10, C, seed=1).requires_grad_(True)
    w = (1 + rnd(C, seed=2) * 0.3).requires_grad_(True)
    b = rnd(C, seed=3).requires_grad_(True)
    g = rnd(2, 6, 10, C, seed=4)
    y = ln_ref(x, w, b)
    y.backward(g)
    out = ops.layernorm2d_fwd(dev(x.detach()), dev(w.detach()), dev(b.detach()))
    close(out, y.detach())
    gx = torch.full((2, 6, 10, C), 0.5, device="cuda")
    dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ops.layernorm2d_bwd(dev(g), dev(x.detach()), dev(w.detach()), gx, dw, db, accumulate=True)
    close(gx, x.grad + 0.5)
    close(dw, w.grad, atol=1e-4)
    close(db, b.grad, atol=1e-4)


@pytest.mark.parametrize("C,H,W", [(16, 12, 20), (64, 16, 16)])
def test_dwconv_gelu_fwd_bwd(C, H, W):
    ops = _ops()
    N = 2
    x = rnd(N, C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, 1, 3, 3, seed=2, scale=0.5).requires_grad_(True)
    b = rnd(C, seed=3).requires_grad_(True)
    pre = F.conv2d(x, w, b, 1, 1, 1, C)
    act = F.gelu(pre)
    xd = dev(x.detach().permute(0, 2, 3, 1))
    pre_d, act_d, pool = ops.dwconv3x3_gelu_fwd(xd, dev(w.detach()), dev(b.detach()), want_pool=True)
    close(pre_d, pre.detach().permute(0, 2, 3, 1))
    close(act_d, act.detach().permute(0, 2, 3, 1))
    close(pool.sum(1), act.detach().sum((2, 3)), atol=1e-3)
    _, _, pool2 = ops.dwconv3x3_gelu_fwd(xd, dev(w.detach()), dev(b.detach()), want_pool=True)
    assert torch.equal(pool, pool2)                      # deterministic (no atomics on the forward path)
    gd = rnd(N, C, H, W, seed=4)
    pre.backward(gd)
    dw = torch.zeros(C, 1, 3, 3, device="cuda"); db = torch.zeros(C, device="cuda")
    gin = ops.dwconv3x3_bwd(dev(gd.permute(0, 2, 3, 1)), xd, dev(w.detach()), dw, db)
    close(gin, x.grad.permute(0, 2, 3, 1))
    close(dw, w.grad, atol=1e-3)
    close(db, b.grad, atol=1e-3)


def test_gelu_and_colsum():
    ops = _ops()
    x = rnd(2, 8, 8, 64, seed=1, scale=3).requires_grad_(True)
    g = rnd(2, 8, 8, 64, seed=2)
    y = F.gelu(x)
    y.backward(g)
    close(ops.gelu_fwd(dev(x.detach())), y.detach())
    close(ops.gelu_bwd(dev(g), dev(x.detach())), x.grad)
    db = torch.ones(64, device="cuda")
    ops.colsum(dev(g), db)
    close(db, g.sum((0, 1, 2)) + 1, atol=1e-4)
    db8 = torch.zeros(8, device="cuda")
    ops.colsum(dev(g)[..., :8], db8)
    close(db8, g[..., :8].sum((0, 1, 2)), atol=1e-4)


def test_se_scale_and_backward_pieces():
    ops = _ops()
    N, H, W, C = 2, 8, 12, 64
    xi = rnd(N, H, W, C, seed=1).requires_grad_(True)
    dwe = rnd(N, H, W, C, seed=2).requires_grad_(True)
    W1 = rnd(C // 2, C, seed=3, scale=0.3).requires_grad_(True); b1 = rnd(C // 2, seed=4).requires_grad_(True)
    W2 = rnd(C, C // 2, seed=5, scale=0.3).requires_grad_(True); b2 = rnd(C, seed=6).requires_grad_(True)
    xe = F.gelu(dwe)
    m = xe.mean((1, 2))
    z1 = F.relu(m @ W1.t() + b1)
    s = torch.sigmoid(z1 @ W2.t() + b2)
    xs = torch.cat([xi * s[:, None, None, :], xe * s[:, None, None, :]], -1)
    gxs = rnd(N, H, W, 2 * C, seed=7)
    xs.backward(gxs)
    # forward pieces
    pool = dev(xe.detach().sum((1, 2)))
    m_d, z1_d, s_d = ops.se_fwd(pool, 1.0 / (H * W), dev(W1.detach()), dev(b1.detach()), dev(W2.detach()), dev(b2.detach()))
    close(m_d, m.detach()); close(z1_d, z1.detach()); close(s_d, s.detach())
    xs_d = ops.scale_cat(dev(xi.detach()), dev(xe.detach()), s_d)
    close(xs_d, xs.detach())
    # backward pieces
    gs = ops.egaca_gs_reduce(dev(gxs), dev(xi.detach()), dev(xe.detach()))
    gs_ref = (gxs[..., :C] * xi.detach() + gxs[..., C:] * xe.detach()).sum((1, 2))
    close(gs, gs_ref, atol=1e-4)
    dW1 = torch.zeros_like(dev(W1.detach())); db1 = torch.zeros(C // 2, device="cuda")
    dW2 = torch.zeros_like(dev(W2.detach())); db2 = torch.zeros(C, device="cuda")
    gm = ops.se_bwd(gs, s_d, z1_d, m_d, dev(W1.detach()), dev(W2.detach()), dW1, db1, dW2, db2)
    close(dW1, W1.grad, atol=1e-4); close(db1, b1.grad, atol=1e-4)
    close(dW2, W2.grad, atol=1e-4); close(db2, b2.grad, atol=1e-4)
    gxi = torch.full((N, H, W, C), 0.25, device="cuda")
    gdwe = ops.egaca_bwd_elem(dev(gxs), s_d, gm, dev(dwe.detach()), gxi, accumulate_xi=True)
    close(gdwe, dwe.grad)
    close(gxi, xi.grad + 0.25)


def test_fold_back_and_scaled_pack():
    ops = _ops()
    Co, Ci = 64, 128
    Wt = rnd(Co, Ci, 1, 1, seed=1, scale=0.2).requires_grad_(True)
    b = rnd(Co, seed=2).requires_grad_(True)
    beta = rnd(Co, seed=3).requires_grad_(True)
    x = rnd(1, Ci, 8, 32, seed=4)
    r = rnd(1, Co, 8, 32, seed=5)
    y = r + F.conv2d(x, Wt, b) * beta.view(1, -1, 1, 1)          # fm:319 form
    g = rnd(1, Co, 8, 32, seed=6)
    y.backward(g)
    kc, bn = ops.conv_kc(1, 1, 1, 0), ops.conv_bn(1, 1, 1, 0, Co)
    wp = ops.pack_conv_weights(dev(Wt.detach()), ops.ROLE_FWD, bn, kc, 1, 1, Co, Ci, oscale=dev(beta.detach()))
    beff = ops.mul_vec(dev(b.detach()), dev(beta.detach()))
    out = torch.empty(1, 8, 32, Co, device="cuda")
    ops.conv2d(dev(x.permute(0, 2, 3, 1)), wp, out, kh=1, kw=1, cout=Co, cout_pad=Co, bias=beff,
               res=dev(r.permute(0, 2, 3, 1)))
    close(out, y.detach().permute(0, 2, 3, 1))
    # gradients of the folded conv, then fold back
    gwf = torch.zeros(Co, Ci, 1, 1, device="cuda"); gbf = torch.zeros(Co, device="cuda")
    ops.conv2d_wgrad(dev(g.permute(0, 2, 3, 1)), dev(x.permute(0, 2, 3, 1)), gwf, kh=1, kw=1, db=gbf)
    # the accumulated gradients already hold something (gradient accumulation): it must be added to,
    # never rescaled
    gw = torch.full((Co, Ci, 1, 1), 0.5, device="cuda"); gb = torch.full((Co,), -0.25, device="cuda")
    dbeta = torch.full((Co,), 2.0, device="cuda")
    ops.fold_back(dev(Wt.detach()), dev(b.detach()), dev(beta.detach()), gwf, gbf, gw, gb, dbeta)
    close(gw, Wt.grad + 0.5, atol=1e-4); close(gb, b.grad - 0.25, atol=1e-4); close(dbeta, beta.grad + 2.0, atol=1e-4)


def test_charbonnier_norm_adamw():
    ops = _ops()
    n = 4 * 1000
    pred = rnd(n, seed=1).requires_grad_(True)
    gt = rnd(n, seed=2)
    loss = torch.sqrt((pred - gt) ** 2 + 1e-12).mean()
    loss.backward()
    grad = torch.empty(n, device="cuda")
    ls = ops.charbonnier(dev(pred.detach()), dev(gt), grad)
    assert abs(ls.item() / n - loss.item()) < 1e-6
    close(grad, pred.grad, rtol=1e-4, atol=1e-9)
    # clip + AdamW, 3 steps, against torch
    p0 = rnd(n, seed=3)
    p_t = torch.nn.Parameter(p0.float().clone())
    opt = torch.optim.AdamW([p_t], lr=2e-4, weight_decay=1e-4, betas=(0.9, 0.99))
    p = dev(p0); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        g = rnd(n, seed=10 + step, scale=0.05).float()
        p_t.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([p_t], 0.01)
        opt.step()
        gd = g.cuda()
        sq = ops.grad_sqnorm(gd)
        assert abs(math.sqrt(sq[0].item()) - float(tn)) < 1e-5 * float(tn)
        assert torch.equal(ops.grad_sqnorm(gd)[0], sq[0])          # deterministic
        ops.clip_adamw(p, gd, m, v, sq, max_norm=0.01, lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, step=step)
        close(p, p_t.detach(), rtol=1e-5, atol=1e-7)


def _egaca_ref(P, a, ev, img):
    """fusion_modules.py:290-333 in float64 (NCHW), from a flat parameter dict with prefix a."""
    g = lambda k: P[f"{a}.{k}"].double()                                       # noqa: E731

    def ln(x, n):
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        return g(n + ".weight").view(1, -1, 1, 1) * ((x - mu) / (var + 1e-6).sqrt()) + g(n + ".bias").view(1, -1, 1, 1)

    c = ev.shape[1]
    xi = F.gelu(F.conv2d(F.conv2d(ln(img, "norm1"), g("conv1.weight"), g("conv1.bias")), g("conv2.weight"), g("conv2.bias"),
                         padding=1, groups=c))
    xe = F.gelu(F.conv2d(F.conv2d(ln(ev, "norm1_e"), g("conv1_e.weight"), g("conv1_e.bias")), g("conv2_e.weight"),
                         g("conv2_e.bias"), padding=1, groups=c))
    m = xe.mean((2, 3), keepdim=True)
    s = torch.sigmoid(F.conv2d(F.relu(F.conv2d(m, g("se_1.1.weight"), g("se_1.1.bias"))), g("se_1.3.weight"), g("se_1.3.bias")))
    zf = F.conv2d(torch.cat([xi * s, xe * s], 1), g("conv3.weight"), g("conv3.bias"))
    y = ev + img + zf * g("beta")
    ffn = F.conv2d(F.gelu(F.conv2d(ln(y, "norm2"), g("conv4.weight"), g("conv4.bias"))), g("conv5.weight"), g("conv5.bias"))
    return F.conv2d(y, g("conv_y_side.weight"), g("conv_y_side.bias")) + ffn * g("gamma")


@pytest.mark.parametrize("shape", [(2, 16, 16), (1, 24, 32), (3, 8, 16)])
def test_egaca_fused_forward_six_launches(shape):
    """The fused EGACA forward (LayerNorm prologues, squeeze-excite + operand scaling inside conv3, second residual, GELU
    second output: refid_pw_extras) at the shipped width (64 -> 128 channels): output vs a float64 restatement of
    fusion_modules.py:290-333, every stashed tensor vs the one-kernel-per-op path (which the backward pass was written
    against), and the launch count."""
    from oracle import refid_oracle as O
    from refid_amd import engine as E, ops
    from refid_amd.archs import define_network
    n, h, w = shape
    P = O.make_params(26, mode="hash", seed=3)
    a = "encoders_forward.1.atten_fuse"
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3,
                              base_num_channels=32, num_block=1, num_residual_blocks=2))
    net.load_state_dict(P)
    net = net.cuda()
    eng = net.engine
    eng.repack()
    A = eng.enc_f[1].att
    ev = rnd(n, 64, h, w, seed=1)
    img = rnd(n, 64, h, w, seed=2)
    ref = _egaca_ref(P, a, ev, img)
    evd, imgd = dev(ev.permute(0, 2, 3, 1)), dev(img.permute(0, 2, 3, 1))
    outs, stash, counts = {}, {}, {}
    for fused in (True, False):
        old, E.EGACA_FUSED = E.EGACA_FUSED, fused
        calls = []
        names = ("conv2d", "layernorm2d_fwd", "dwconv3x3_gelu_fwd", "se_fwd", "scale_cat", "add", "gelu_fwd")
        saved = {k: getattr(ops, k) for k in names}
        try:
            for k in names:
                setattr(ops, k, (lambda f, k: (lambda *aa, **kw: (calls.append(k), f(*aa, **kw))[1]))(saved[k], k))
            ip = eng._egaca_img_path(A, imgd)
            calls.clear()
            st = {}
            outs[fused] = eng._egaca_fwd(A, evd, imgd, ip, st)
            stash[fused] = st["eg"]
            counts[fused] = list(calls)
            plain = eng._egaca_fwd(A, evd, imgd, ip, None)             # inference: no side outputs
            assert torch.equal(plain, outs[fused])
        finally:
            for k in names:
                setattr(ops, k, saved[k])
            E.EGACA_FUSED = old
    # (round 4: conv_y_side rides in conv5's launch as a second operand source -> 5 launches; REFID_LINEAR_SPLIT=0: 6)
    assert len(counts[True]) == (5 if A.wp_cat is not None else 6) and len(counts[False]) == 12, (counts[True], counts[False])
    close(outs[True].permute(0, 3, 1, 2), ref, rtol=1e-3, atol=1e-4)
    close(outs[False].permute(0, 3, 1, 2), ref, rtol=1e-3, atol=1e-4)
    for k in stash[False]:
        close(stash[True][k], stash[False][k], rtol=1e-4, atol=2e-5)
