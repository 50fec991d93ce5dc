"""GPU parity of the HIP ``SingleMultiConnectEVHINet`` against the reference-generated golden
vectors (tests/golden/evhinet_*.npz, made by oracle/make_golden.py from the reference class itself) and of its two
non-GEMM kernels against plain torch.  Bar: rtol 1e-3 / atol 1e-4 (fp32)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import evhinet_oracle as E

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-3, 1e-4


def _net(wf, P):
    from refid_amd.archs import define_network
    net = define_network(dict(type="SingleMultiConnectEVHINet", wf=wf))
    net.load_state_dict(P, strict=True)
    return net.cuda()


@pytest.mark.parametrize("name", ["evhinet_tiny_train", "evhinet_odd_train", "evhinet_full_fwd"])
def test_forward_and_gradients_match_the_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    wf, B, H, W, seed = [int(v) for v in z["meta"]]
    P = E.make_params(seed=seed, wf=wf)
    x, ev, gt = E.make_inputs(B, H, W, seed=seed)
    net = _net(wf, P)
    train = "loss" in z.files
    sub = 2 if name == "evhinet_full_fwd" else 1
    if not train:
        with torch.no_grad():
            out = net(x=x.cuda(), event=ev.cuda())
        assert isinstance(out, list) and len(out) == 1
        np.testing.assert_allclose(out[0].cpu().numpy()[..., ::sub, ::sub], z["out"], rtol=RTOL, atol=ATOL)
        return
    out = net(x=x.cuda(), event=ev.cuda())
    assert isinstance(out, list) and len(out) == 1 and out[0].requires_grad
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), z["out"], rtol=RTOL, atol=ATOL)
    loss = E.psnr_loss(out[0], gt.cuda())                      # torch ops on the GPU tensor; the network's part is HIP
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    dead = set(str(k) for k in z["no_grad_keys"])
    worst = []
    for k, p in net.named_parameters():
        ref = z["grad/" + k]
        g = p.grad.cpu().numpy()
        if k in dead:
            assert float(np.abs(g).max()) == 0.0, f"{k}: dead parameter must have an exactly-zero gradient"
            continue
        scale = max(float(np.abs(ref).max()), 1e-8)
        worst.append((float(np.abs(g - ref).max()) / scale, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:5]


def test_state_dict_round_trip_and_errors():
    from refid_amd.archs import define_network
    net = define_network(dict(type="SingleMultiConnectEVHINet", wf=8)).cuda()
    assert list(net.state_dict().keys()) == list(E.param_shapes(wf=8).keys())
    with pytest.raises(RuntimeError):
        net(x=torch.zeros(1, 3, 30, 32, device="cuda"), event=torch.zeros(1, 6, 30, 32, device="cuda"))
    with pytest.raises(NotImplementedError):
        define_network(dict(type="SingleMultiConnectEVHINet", fac_before_downsample=False))


@pytest.mark.parametrize("cfg", [(2, 8, 12, 8, 4), (1, 16, 16, 64, 32), (3, 5, 7, 16, 8), (1, 64, 48, 128, 64), (1, 8, 8, 16, 0)])
def test_hin_lrelu_kernels(cfg):
    from refid_amd import ops
    N, H, W, C, ch = cfg
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(ch, generator=g)).requires_grad_(True) if ch else None
    beta = (0.2 * torch.randn(ch, generator=g)).requires_grad_(True) if ch else None
    if ch:
        y = torch.cat([F.instance_norm(x[:, :ch], weight=gamma, bias=beta, eps=1e-5), x[:, ch:]], 1)
    else:
        y = x
    ref = F.leaky_relu(y, 0.2)
    go = torch.randn(N, C, H, W, generator=g)
    ref.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    out, stats = ops.hin_lrelu_fwd(xd, gamma.detach().cuda() if ch else None, beta.detach().cuda() if ch else None, 0.2)
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    dg = torch.zeros(ch, device="cuda") if ch else None
    db = torch.zeros(ch, device="cuda") if ch else None
    gx = ops.hin_lrelu_bwd(go.permute(0, 2, 3, 1).contiguous().cuda(), out, xd, gamma.detach().cuda() if ch else None,
                           stats, dg, db, 0.2)
    np.testing.assert_allclose(gx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy(), rtol=1e-3, atol=2e-5)
    if ch:
        np.testing.assert_allclose(dg.cpu().numpy(), gamma.grad.numpy(), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(db.cpu().numpy(), beta.grad.numpy(), rtol=1e-3, atol=1e-3)
        out2, _ = ops.hin_lrelu_fwd(xd, gamma.detach().cuda(), beta.detach().cuda(), 0.2)
        assert torch.equal(out, out2)                          # deterministic reductions


def test_fac_bias_kernels():
    from refid_amd import ops
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(2, 16, 9, 11, generator=g).requires_grad_(True)
    filt = torch.randn(2, 32, 9, 11, generator=g).requires_grad_(True)
    w, b = torch.chunk(filt, 2, dim=1)
    ref = feat * w + b
    go = torch.randn(2, 16, 9, 11, generator=g)
    ref.backward(go)
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()      # noqa: E731
    out = ops.fac_fwd(nh(feat), nh(filt))
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    gf, gfi = ops.fac_bwd(nh(go), nh(feat), nh(filt))
    np.testing.assert_allclose(gf.permute(0, 3, 1, 2).cpu().numpy(), feat.grad.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gfi.permute(0, 3, 1, 2).cpu().numpy(), filt.grad.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("loss_type", ["PSNRLoss", "CharbonnierLoss"])
def test_train_steps_match_torch_adamw_on_the_oracle(loss_type):
    """ImageEventRestorationModel.optimize_parameters (HIP forward/backward, fused clip + AdamW) == the oracle network under
    torch's clip_grad_norm_(0.01) + AdamW (image_event_restoration_model.py:272-322), two steps."""
    from refid_amd.train import ImageEventRestorationModel
    opt = {"name": "t", "is_train": True, "num_gpu": 1,
           "network_g": dict(type="SingleMultiConnectEVHINet", wf=8), "path": {"pretrain_network_g": None},
           "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                     "scheduler": dict(type="TrueCosineAnnealingLR", T_max=50, eta_min=1e-7),
                     "pixel_opt": dict(type=loss_type, loss_weight=0.5, reduction="mean")},
           "val": {"max_minibatch": 1}}
    model = ImageEventRestorationModel(opt)
    P0 = E.make_params(seed=4, wf=8)
    model.net_g.load_state_dict(P0)
    x, ev, gt = E.make_inputs(2, 32, 32, seed=4)
    # torch reference on the CPU oracle
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    optim = torch.optim.AdamW(list(P.values()), lr=2e-4, weight_decay=1e-4, betas=(0.9, 0.99))
    losses = []
    for it in (1, 2):
        optim.zero_grad()
        out = E.forward(P, x, ev)
        loss = E.psnr_loss(out, gt, 0.5) if loss_type == "PSNRLoss" else 0.5 * torch.sqrt((out - gt) ** 2 + 1e-12).mean()
        (loss + 0 * sum(p.sum() for p in P.values())).backward()          # :315, gives the dead parameters zero grads
        torch.nn.utils.clip_grad_norm_(list(P.values()), 0.01)
        for g in optim.param_groups:
            g["lr"] = 2e-4 if it == 1 else 1e-7 + (2e-4 - 1e-7) * (1 + np.cos(np.pi * 1 / 50)) / 2
        optim.step()
        losses.append(float(loss))
        model.update_learning_rate(it)
        model.feed_data({"lq": x, "voxel": ev, "gt": gt})
        model.optimize_parameters(it)
        assert abs(model.get_current_log()["l_pix"] - losses[-1]) < 2e-4 * max(1.0, abs(losses[-1]))
    sd = model.net_g.state_dict()
    for k in sd:
        a, b = sd[k].double().cpu(), P[k].detach().double()
        disp = (b - P0[k].double()).abs().max().item()
        assert (a - b).abs().max().item() <= 0.03 * disp + 1e-9, k
    model.test()
    assert model.output.shape == (2, 3, 32, 32)
