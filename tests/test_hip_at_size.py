"""GPU: the BASELINE.json configurations AT THEIR FULL SIZES (VERDICT r1 "what's weak" #1-2, configs_untested).

The oracle cannot run a config-2 train step in seconds, so full-size runs are checked through size-independent
properties (per-sample linearity of the gradient, accumulation, stream-overlap / split-K invariance) and through
oracle runs on bounded crops of the SAME workload (one sample, few time steps, full resolution and width).
  configs[1]  B=8, 256x256, T=23, 26 ch, fp32 train step       test_config2_*
  configs[2]  B=4/GPU, 256x256, T=25 (11+3), bf16 compute        test_config3_*
  configs[4]  1224x1632, T=15, 512x512 tiles, rank-sharded       test_config5_*
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import refid_oracle as O

pytestmark = pytest.mark.gpu


def _params(img_chn, seed):
    """Released-checkpoint-like weights: the reference's initialisation with beta/gamma ~ N(0, 0.1^2)."""
    torch.manual_seed(seed)
    P = O.make_params(img_chn, mode="init", seed=seed)
    for k in P:
        if k.endswith((".beta", ".gamma")):
            P[k] = torch.randn_like(P[k]) * 0.1
    return P


def _net(img_chn, P, dtype="fp32"):
    from refid_amd.archs import define_network
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                              base_num_channels=32, num_block=1, num_residual_blocks=2, compute_dtype=dtype))
    net.load_state_dict(P, strict=True)
    return net.cuda()


def _charbonnier_grad(pred, gt, count):
    d = pred - gt
    return d / torch.sqrt(d * d + 1e-12) / count


def _per_tensor_err(eng, a, b):
    """max over parameters of max|a-b| / max|b| (per-tensor max-normalised error), and the parameter's name."""
    worst, name = 0.0, None
    for k, (o, n) in eng.arena.offsets.items():
        ref = b[o:o + n]
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(a[o:o + n].abs().max()) == 0.0, k
            continue
        e = float((a[o:o + n] - ref).abs().max()) / scale
        if e > worst:
            worst, name = e, k
    return worst, name


def test_config2_train_step_gradient_properties():
    """B=8, 256x256, T=23 BPTT with the production settings (side-stream weight gradients, persistent slabs, split-K):
    (1) stream overlap on == off; (2) Winograd split-K 'sample' == never, within 1e-5 of each tensor's scale;
    (3) the gradient is linear in the samples: backward of a 2-sample batch == the two samples run one after the
    other into the same arena (which also exercises gradient accumulation at size)."""
    from refid_amd import engine as E, ops
    P = _params(26, 11)
    net = _net(26, P)
    eng = net.engine
    g = torch.Generator().manual_seed(5)
    B, T, H = 8, 23, 256
    x, ev, gt = O.make_inputs(B, T, H, H, 26, seed=2, mode="rng")
    x, ev, gt = x.cuda(), ev.cuda(), gt.cuda()
    count = B * T * 3 * H * H

    def run(xs, es, gs, zero=True):
        if zero:
            eng.zero_grad()
        pred = eng.forward(xs, es, save=True)
        eng.backward(_charbonnier_grad(pred, gs, count))
        torch.cuda.synchronize()
        return pred

    pred = run(x, ev, gt)
    assert bool(torch.isfinite(pred).all())
    base = eng.arena.flat_g.clone()
    assert bool(torch.isfinite(base).all()) and float(base.abs().max()) > 0
    # (1) the weight gradients on the side stream (the default is the main stream since round 5) give the same gradients
    old = E.OVERLAP_WGRAD
    E.OVERLAP_WGRAD = not old
    try:
        run(x, ev, gt)
    finally:
        E.OVERLAP_WGRAD = old
    e1, k1 = _per_tensor_err(eng, eng.arena.flat_g, base)
    assert e1 <= 1e-5, (e1, k1)                      # only the atomically accumulated LN / depthwise sums may differ
    # (2) no split-K in the Winograd tile
    olds, ops.WINO_SPLIT = ops.WINO_SPLIT, 0
    try:
        p0 = run(x, ev, gt)
    finally:
        ops.WINO_SPLIT = olds
    e2, k2 = _per_tensor_err(eng, eng.arena.flat_g, base)
    assert e2 <= 1e-5, (e2, k2)
    assert float((p0 - pred).abs().max()) <= 1e-5
    # (3) linearity over samples + accumulation, at T=23.  Under the default split-K policy (by total grid size) a
    # 2-sample launch and two 1-sample launches split K differently: equal to summation-order rounding; under the
    # per-sample policy the partial sums are the same ones
    for policy, tol in ((ops.WINO_SPLIT, 2e-4), (1, 2e-5)):
        olds, ops.WINO_SPLIT = ops.WINO_SPLIT, policy
        try:
            run(x[2:4], ev[2:4], gt[2:4])
            two = eng.arena.flat_g.clone()
            run(x[2:3], ev[2:3], gt[2:3])
            run(x[3:4], ev[3:4], gt[3:4], zero=False)
        finally:
            ops.WINO_SPLIT = olds
        e3, k3 = _per_tensor_err(eng, eng.arena.flat_g, two)
        assert e3 <= tol, (policy, e3, k3)


_CROP_REF = {}


def _crop_reference():
    if not _CROP_REF:
        P = _params(26, 12)
        x, ev, gt = O.make_inputs(1, 3, 256, 256, 26, seed=3, mode="rng")
        Pc = {k: v.clone() for k, v in P.items()}
        _CROP_REF["v"] = (P, x, ev, gt) + tuple(O.train_step(Pc, O.TrainState(Pc), x, ev, gt))
    return _CROP_REF["v"]


@pytest.mark.parametrize("policy", ["auto", "bench_tiles"])
def test_config2_oracle_crop_gradients(policy):
    """One sample of the config-2 workload at full resolution and width, first 3 time steps: output and ALL 183
    parameter gradients against the oracle (per-tensor max-normalised).
    policy "auto": tile choice by the total grid of this 1-sample launch (split-K forms, conv_down on the fp32 MFMA tile);
    policy "bench_tiles": the per-sample policy (`REFID_SPLITK=sample`: decided as if 8 samples were in the batch), i.e.
    exactly the kernels the B=8 bench runs -- conv_down forward / input gradient on the six-product split tile, the
    Winograd x six-product tile without split-K at the upper levels."""
    from refid_amd import ops
    P, x, ev, gt, loss_ref, gnorm_ref, grads_ref, pred_ref = _crop_reference()
    net = _net(26, P)
    olds, ops.WINO_SPLIT = ops.WINO_SPLIT, (1 if policy == "bench_tiles" else ops.WINO_SPLIT)
    try:
        pred = net(x=x.cuda(), event=ev.cuda())
        loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.WINO_SPLIT = olds
    np.testing.assert_allclose(pred.detach().cpu().numpy(), pred_ref.numpy(), rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - float(loss_ref)) < 1e-5 * float(loss_ref)
    worst = []
    for k, p in net.named_parameters():
        r = grads_ref[k].double()
        scale = float(r.abs().max())
        if scale == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst.append((float((p.grad.double().cpu() - r).abs().max()) / scale, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 1e-3, worst[:5]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())))
    assert abs(gn - float(gnorm_ref)) < 1e-4 * float(gnorm_ref), (gn, float(gnorm_ref))


def test_default_constructor_full_width_gradients():
    """The reference's OWN constructor defaults (arch:90-92: num_encoders=4, num_block=3; every YAML overrides them with 3 / 1) at
    full width (base 32: 64 .. 512 channels, 1024 -> 512 trunk convs at the fourth level), one sample at 128 x 128, two time steps:
    output and all 282 parameter gradients against the oracle (per-tensor max-normalised), exactly as the config-2 crop test."""
    from refid_amd.archs import define_network
    torch.manual_seed(11)
    P = O.make_params(26, mode="init", seed=11, num_block=3, num_encoders=4)
    for k in P:
        if k.endswith((".beta", ".gamma")):
            P[k] = torch.randn_like(P[k]) * 0.1
    x, ev, gt = O.make_inputs(1, 2, 128, 128, 26, seed=12, mode="rng")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Pc = {k: v.clone() for k, v in P.items()}
    loss_ref, gnorm_ref, grads_ref, pred_ref = O.train_step(Pc, O.TrainState(Pc), x, ev, gt, num_encoders=4)
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2))       # every other keyword: the ctor default
    assert net.num_encoders == 4 and net.num_block == 3 and len(net.state_dict()) == len(P) == 282
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    pred = net(x=x.cuda(), event=ev.cuda())
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), pred_ref.numpy(), rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - float(loss_ref)) < 1e-5 * float(loss_ref)
    worst = []
    for k, p in net.named_parameters():
        r = grads_ref[k].double()
        scale = float(r.abs().max())
        if scale == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst.append((float((p.grad.double().cpu() - r).abs().max()) / scale, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:5]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())))
    assert abs(gn - float(gnorm_ref)) < 1e-4 * float(gnorm_ref), (gn, float(gnorm_ref))
    with pytest.raises(RuntimeError, match="multiples of 16"):
        net(x=x[..., :120, :].cuda(), event=ev[..., :120, :].cuda())


def test_config3_bf16_workload():
    """configs[2] per-GPU workload: B=4, T=25 (11+3 -> 2*11+3), 256x256, bf16 compute, one fused train step; quality
    parity of the bf16 forward against the oracle's fp32 output on one full-length sample (PSNR criterion, SURVEY 8d:
    the reference has no bf16 path of its own)."""
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    B, T, H = 4, 25, 256
    P = _params(26, 13)
    opt = {"name": "c3", "is_train": True, "num_gpu": 1,
           "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3,
                             base_num_channels=32, num_block=1, num_residual_blocks=2, compute_dtype="bf16"),
           "path": {"pretrain_network_g": None},
           "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                     "scheduler": dict(type="TrueCosineAnnealingLR", T_max=200000, eta_min=1e-7),
                     "pixel_opt": dict(type="CharbonnierLoss", loss_weight=1, reduction="mean")}, "val": {}}
    model = TwoImageEventRecurrentRestorationModel(opt)
    model.net_g.load_state_dict(P)
    x, ev, gt = O.make_inputs(B, T, H, H, 26, seed=4, mode="rng")
    with torch.no_grad():
        ref = O.forward(P, x[1:2], ev[1:2])                           # fp32 oracle, one full-length sample
        got = model.net_g(x=x[1:2].cuda(), event=ev[1:2].cuda()).cpu()
    assert got.shape == (1, T, 3, H, H)
    assert O.psnr_between(got, ref) > 50.0
    p16, p32 = O.psnr_between(got.clamp(0, 1), gt[1:2]), O.psnr_between(ref.clamp(0, 1), gt[1:2])
    assert abs(p16 - p32) < 0.01, (p16, p32)
    # the train step of the workload itself, against the fp32 HIP path on the same batch
    model.feed_data({"lq": x, "voxel": ev, "gt": gt})
    model.update_learning_rate(1)
    model.optimize_parameters(1)
    l16, g16 = model.get_current_log()["l_pix"], model.grad_norm()
    opt32 = dict(opt, network_g=dict(opt["network_g"], compute_dtype="fp32"))
    m32 = TwoImageEventRecurrentRestorationModel(opt32)
    m32.net_g.load_state_dict(P)
    m32.feed_data({"lq": x, "voxel": ev, "gt": gt})
    m32.update_learning_rate(1)
    m32.optimize_parameters(1)
    l32, g32 = m32.get_current_log()["l_pix"], m32.grad_norm()
    assert np.isfinite(l16) and abs(l16 - l32) < 2e-3 * l32, (l16, l32)
    assert abs(g16 - g32) < 0.05 * g32, (g16, g32)


# ---- config 5 -----------------------------------------------------------------------------------------
C5 = dict(H=1224, W=1632, T=15, crop=512)


def _c5_inputs():
    return O.make_inputs(1, C5["T"], C5["H"], C5["W"], 6, seed=6, mode="rng")   # 5-D sharp-VFI input (1,2,3,H,W)


def _c5_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refid_amd.tiling import tiled_forward
    torch.cuda.set_device(0)
    net = _net(6, _params(6, 14)).eval()
    x, ev, _ = _c5_inputs()
    out = tiled_forward(net, x.cuda(), ev.cuda(), crop=C5["crop"], max_minibatch=2, rank=rank, world=world)
    if rank == 0:
        torch.save(out.cpu(), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_config5_full_resolution_tiled_inference(tmp_path):
    """configs[4]: 1224x1632, T=15, the reference's 512x512 tile grid (twoImage_event_recurrent_model.py:190-270 made 5-D
    aware).  (1) the clamped last tile vs the oracle at full tile size; (2) the overlap-averaged
    assembly: single rank == two ranks sharing the tile list (gloo, the GPU is shared); (3) the validation tail
    (metrics.py PSNR / 3-D SSIM) on the assembled frames vs the oracle's restatement, on two 512x512 regions."""
    from refid_amd.metrics import calculate_psnr_frames, calculate_ssim_frames
    from refid_amd.tiling import grid_indices, tiled_forward
    H, W, T, crop = C5["H"], C5["W"], C5["T"], C5["crop"]
    P = _params(6, 14)
    net = _net(6, P).eval()
    x, ev, gt = _c5_inputs()
    idx, ch, cw = grid_indices(H, W, crop)
    assert (ch, cw) == (512, 512) and len(idx) == 3 * 4 and idx[-1] == dict(i=H - 512, j=W - 512)
    assert [d for d in O.tile_grid(H, W, crop)] == idx                 # same grid as the oracle's restatement of grids()
    xc, ec = x.cuda(), ev.cuda()
    # (1)
    with torch.no_grad():
        for d in (idx[-1],):                  # (the clamped last tile; one 512 x 512, T = 15 oracle forward costs 20-40 s of CPU)
            i, j = d["i"], d["j"]
            ref = O.forward(P, x[..., i:i + 512, j:j + 512], ev[..., i:i + 512, j:j + 512])
            got = net(x=xc[..., i:i + 512, j:j + 512].contiguous(), event=ec[..., i:i + 512, j:j + 512].contiguous())
            np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-4)
    # (2)
    one = tiled_forward(net, xc, ec, crop=crop, max_minibatch=2)
    assert one.shape == (1, T, 3, H, W) and bool(torch.isfinite(one).all())
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = os.path.join(tmp_path, "two_rank.pt")
    mp.spawn(_c5_worker, args=(2, port, path), nprocs=2, join=True)
    two = torch.load(path)
    assert float((two - one.cpu()).abs().max()) <= 1e-6
    # (3) metrics on two regions of the assembled output (oracle SSIM is a CPU 3-D filter: keep it bounded)
    for (i, j) in ((0, 0), (H - 512, W - 512)):
        a = one[0, :2, :, i:i + 512, j:j + 512].contiguous()
        b = gt[0, :2, :, i:i + 512, j:j + 512].contiguous()
        ps = calculate_psnr_frames(a, b.cuda())
        ss = calculate_ssim_frames(a, b.cuda())
        for t in range(2):
            assert abs(ps[t] - O.psnr_u8(O.tensor2img_u8(a[t].cpu()), O.tensor2img_u8(b[t]))) < 1e-9
            assert abs(ss[t] - O.ssim3d_u8(a[t].cpu(), b[t])) < 2e-5
    # whole-frame forward (no tiling) runs at this size too; tiling is an approximation of it by design
    with torch.no_grad():
        whole = net(x=xc, event=ec)
    assert whole.shape == one.shape
    psnr_tw = np.mean(calculate_psnr_frames(one, whole.clamp(0, 1)))
    assert psnr_tw > 20.0, psnr_tw
