"""End-to-end GPU parity of the HIP path against the CPU oracle and the reference-generated
golden vectors.  Bar (BASELINE.json north_star): rtol 1e-3 / atol 1e-4 in fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import refid_oracle as O
from oracle import kink_tape as K

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-3, 1e-4


def build(img_chn, base, P):
    from refid_amd.archs import define_network
    nb = 1 + max(int(k.split(".main.2.")[1].split(".")[0]) for k in P if ".main.2." in k)      # num_block of this state dict
    ne = 1 + max(int(k.split(".")[1]) for k in P if k.startswith("encoders_forward."))          # ... and its num_encoders
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=ne,
                              base_num_channels=base, num_block=nb, num_residual_blocks=2))
    net.load_state_dict(P, strict=True)
    return net.cuda()


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    img_chn, base, B, T, H, W, seed = [int(v) for v in z["meta"][:7]]
    nb = int(z["meta"][7]) if len(z["meta"]) > 7 else 1                   # num_block (tiny26_nb2_train: 2)
    ne = int(z["meta"][8]) if len(z["meta"]) > 8 else 3                   # num_encoders (tiny26_default_ctor_train: 4, tiny6_ne2_train: 2)
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed, num_block=nb, num_encoders=ne)
    x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash")
    return z, P, x, ev, gt, img_chn, base


@pytest.mark.parametrize("name", ["tiny26_train", "tiny6_train", "odd26_fwd", "full26_train", "tiny26_nb2_train",
                                  "tiny26_default_ctor_train", "tiny6_ne2_train"])
def test_forward_matches_reference_golden(golden_dir, name):
    z, P, x, ev, gt, img_chn, base = load(golden_dir, name)
    net = build(img_chn, base, P)
    with torch.no_grad():
        out = net(x=x.cuda(), event=ev.cuda())
    assert out.shape == z["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), z["out"], rtol=RTOL, atol=ATOL)


def test_forward_config1(golden_dir):
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "config1_fwd")
    net = build(img_chn, base, P)
    with torch.no_grad():
        out = net(x=x.cuda(), event=ev.cuda())
    assert tuple(out.shape) == (1, 4, 3, 128, 128)
    np.testing.assert_allclose(out[..., ::4, ::4].cpu().numpy(), z["out_sub"], rtol=RTOL, atol=ATOL)


def test_outputs_and_gradients_are_bit_deterministic():
    """No floating-point atomics on the path: parameter-gradient reductions (LayerNorm / depthwise / squeeze-excite /
    bias sums / weight-gradient slabs) run in a fixed order, so two backward passes over the same inputs give the same
    BITS for the output and all 183 gradients (weight-gradient kernels on the side stream included)."""
    P = O.make_params(26, base_num_channels=32, mode="hash", seed=5)
    net = build(26, 32, P)
    x, ev, gt = O.make_inputs(2, 3, 64, 64, 26, seed=5, mode="hash")
    x, ev, gt = x.cuda(), ev.cuda(), gt.cuda()
    runs = []
    for _ in range(3):
        net.zero_grad(set_to_none=False)
        pred = net(x=x, event=ev)
        torch.sqrt((pred - gt) ** 2 + 1e-12).mean().backward()
        torch.cuda.synchronize()
        runs.append((pred.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    for pred, grads in runs[1:]:
        assert torch.equal(pred, runs[0][0])
        diff = [k for k in grads if not torch.equal(grads[k], runs[0][1][k])]
        assert not diff, diff[:8]


def test_batched_slab_reductions_equal_the_one_by_one_launches():
    """refid_wgrad_desc.phase 4 + refid_wgrad_finish_flush: the element-wise stages of every op's slab reduction issued as one
    launch per kernel family give the BITS of the per-op launches (same per-element order of additions) -- all 183
    gradients, two backward passes each way (a second pass shows nothing stayed queued)."""
    from refid_amd import engine as E
    P = O.make_params(26, base_num_channels=32, mode="hash", seed=7)
    net = build(26, 32, P)
    x, ev, gt = O.make_inputs(2, 3, 64, 64, 26, seed=7, mode="hash")
    x, ev, gt = x.cuda(), ev.cuda(), gt.cuda()
    old, old_rows = E.FINISH_BATCH, E.ROWS_DEFER
    runs = []
    try:
        for flag in (True, False, True, False):
            E.FINISH_BATCH = flag
            E.ROWS_DEFER = flag                                     # (the queued LayerNorm / depthwise / bias sums: same bits too)
            net.zero_grad(set_to_none=False)
            pred = net(x=x, event=ev)
            torch.sqrt((pred - gt) ** 2 + 1e-12).mean().backward()
            torch.cuda.synchronize()
            runs.append({k: p.grad.clone() for k, p in net.named_parameters()})
    finally:
        E.FINISH_BATCH, E.ROWS_DEFER = old, old_rows
    assert any(float(g.abs().max()) > 0 for g in runs[0].values())
    for grads in runs[1:]:
        diff = [k for k in grads if not torch.equal(grads[k], runs[0][k])]
        assert not diff, diff[:8]


def _grad_check(net, P, grads_ref, rtol=2e-3):
    worst = []
    for k, p in net.named_parameters():
        g = p.grad.double().cpu()
        r = grads_ref[k].double()
        scale = max(float(r.abs().max()), 1e-7)
        err = float((g - r).abs().max()) / scale
        worst.append((err, k))
        if float(r.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, f"{k}: expected an exactly-zero gradient"
    worst.sort(reverse=True)
    assert worst[0][0] < rtol, f"largest relative gradient errors: {worst[:5]}"


@pytest.mark.parametrize("name", ["tiny26_train", "tiny6_train", "tiny26_nb2_train", "tiny26_default_ctor_train", "tiny6_ne2_train"])
def test_backward_matches_oracle_and_golden(golden_dir, name):
    z, P, x, ev, gt, img_chn, base = load(golden_dir, name)
    net = build(img_chn, base, P)
    Pc = {k: v.clone() for k, v in P.items()}
    st = O.TrainState(Pc)
    loss_ref, gnorm_ref, grads_ref, _ = O.train_step(Pc, st, x, ev, gt, num_encoders=net.num_encoders)
    pred = net(x=x.cuda(), event=ev.cuda())
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()        # torch only as the test's loss
    loss = loss + 0 * sum(p.sum() for p in net.parameters())          # the reference's DDP trick
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-4)
    _grad_check(net, P, grads_ref)
    gn = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    np.testing.assert_allclose(gn, z["grad_norms_all"], rtol=2e-3, atol=1e-7)
    assert int((z["grad_norms_all"] == 0).sum()) == 13 and int((gn == 0).sum()) == 13
    for k in z.files:
        if k.startswith("grad/"):
            ref = z[k]
            got = dict(net.named_parameters())[k[5:]].grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=5e-3, atol=2e-3 * max(np.abs(ref).max(), 1e-7), err_msg=k)


def test_full_width_gradients(golden_dir):
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "full26_train")
    net = build(img_chn, base, P)
    pred = net(x=x.cuda(), event=ev.cuda())
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-4)
    gn = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    np.testing.assert_allclose(gn, z["grad_norms_all"], rtol=3e-3, atol=1e-7)
    total = float(np.sqrt((gn.astype(np.float64) ** 2).sum()))
    np.testing.assert_allclose(total, float(z["grad_norm"]), rtol=1e-3)


def test_h_not_multiple_of_8_raises():
    P = O.make_params(26, base_num_channels=8)
    net = build(26, 8, P)
    x, ev, _ = O.make_inputs(1, 2, 100, 96, 26)
    with pytest.raises(RuntimeError):
        net(x=x.cuda(), event=ev.cuda())


def test_no_cpu_path():
    from refid_amd._lib import RefidHipError
    from refid_amd.archs import define_network
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=6, ev_chn=2, num_encoders=3,
                              base_num_channels=8, num_block=1))
    x, ev, _ = O.make_inputs(1, 2, 16, 16, 6)
    with pytest.raises(RefidHipError):
        net(x=x, event=ev)


def test_bf16_compute_path_psnr_parity(golden_dir):
    """BASELINE config 3 compute dtype: bf16 matrix-core operands.  The reference has no bf16 path of its
    own; the comparison target is its fp32 output: PSNR(bf16 vs fp32) far above the
    +-0.01 dB-on-36 dB criterion (a 55 dB perturbation moves a 36 dB PSNR by < 0.06 dB; measured ~60+)."""
    from refid_amd.archs import define_network
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "full26_train")
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                              base_num_channels=base, num_block=1, num_residual_blocks=2, compute_dtype="bf16"))
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    pred = net(x=x.cuda(), event=ev.cuda())
    ref = torch.from_numpy(z["out"])
    psnr = O.psnr_between(pred.detach().cpu(), ref)
    assert psnr > 50.0, psnr
    # quality metric parity: PSNR against the ground truth moves by less than 0.01 dB
    p16 = O.psnr_between(pred.detach().cpu().clamp(0, 1), gt)
    p32 = O.psnr_between(ref.clamp(0, 1), gt)
    assert abs(p16 - p32) < 0.01, (p16, p32)
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=2e-3)
    gn = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    ref_gn = z["grad_norms_all"]
    big = ref_gn > 1e-3 * ref_gn.max()
    np.testing.assert_allclose(gn[big], ref_gn[big], rtol=0.05)
    assert int((gn == 0).sum()) == 13


def _hip_activation_signs(net, x, ev, B, T):
    """Run the HIP forward with every activation-carrying ConvOp.fwd call recorded: key (conv name, occurrence) -> bool NCHW
    tensor "the positive branch was taken" (= the stored output is > 0: what BPTT's derivative masks test), in the oracle's
    element order (the event head runs time-major here, batch-major there)."""
    from refid_amd import engine
    signs, count = {}, {}
    real = engine.ConvOp.fwd

    def fwd(self, a, b=None, res=None, slope_pre=1.0, slope_post=1.0, out=None, pw=None, bias=True, plus=None):
        r = real(self, a, b, res, slope_pre, slope_post, out, pw, bias, plus)
        if slope_pre != 1.0 or slope_post != 1.0:
            o = r[0] if isinstance(r, tuple) else r
            k = count.get(self.name, 0)
            count[self.name] = k + 1
            pos = (o > 0).permute(0, 3, 1, 2)
            if self.name == "head.conv2d" and B > 1:                      # (T B, ...) time-major -> (B T, ...)
                pos = pos.reshape(T, B, *pos.shape[1:]).transpose(0, 1).reshape(B * T, *pos.shape[1:])
            signs[(self.name, k)] = pos.cpu()
        return r

    engine.ConvOp.fwd = fwd
    try:
        pred = net(x=x.cuda(), event=ev.cuda())
    finally:
        engine.ConvOp.fwd = real
    return pred, signs


@pytest.mark.parametrize("name", ["tiny26_train", "tiny6_train", "full26_train"])
def test_outputs_and_gradients_against_the_float64_oracle(golden_dir, name):
    """The tolerances above (2e-3 per gradient tensor) are set by the fp32 REFERENCE's own round-off, not by the HIP path:
    against the same train step evaluated in float64 (the exact answer both approximate) the HIP path is within 1e-6 on
    every tensor (tools/grad_error_report.py).  Bar here: 5e-6 of the tensor's largest entry for every gradient, 5e-6
    absolute for the outputs, 1e-6 relative for the loss -- against the float64 step GIVEN THE SAME ACTIVATION-SIGN DECISIONS
    (oracle/kink_tape.py; round 6): ReLU / LeakyReLU have a kink at 0, the full-size fixture holds 44 float64
    pre-activations within 1e-6 of their tensor's scale of it (the closest at 4e-8: below fp32's resolution of the sum), and
    which branch an fp32-class arithmetic takes there is chance -- the reference's own fp32 path takes the other branch at
    one of them and lands 5e-5 from the float64 gradient (tests/test_oracle_kinks.py shows it on CPU), the six-bf16-product
    Winograd forward happened to take none, the three-fp16-product forward (per layer no less accurate: tools/bench_wino6.py)
    takes one.  So: the signs the HIP path stored are compared with float64's; every differing element must lie within
    2e-6 of its tensor's scale of zero, there may be at most 8 of them, and the float64 step is re-evaluated with exactly
    those elements on the other branch before the unchanged 5e-6 comparison.  Without differing signs (both tiny fixtures)
    this is the old test."""
    z, P, x, ev, gt, img_chn, base = load(golden_dir, name)
    meta = [int(v) for v in z["meta"]]
    img_chn, base, B, T, H, W, seed = meta
    loss64, g64, out64, pre64 = K.run(*meta, dtype=torch.float64)
    net = build(img_chn, base, P)
    pred, signs = _hip_activation_signs(net, x, ev, B, T)
    assert float((pred.detach().double().cpu() - out64).abs().max()) < 5e-6
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(loss64), rtol=1e-6)
    # every activation the oracle applies after a conv is one recorded call here (the squeeze-excite MLP's ReLU on a (B, 32)
    # vector is computed inside conv3's workgroups, not by a ConvOp: its keys exist on the oracle's side only)
    missing = [k for k in pre64 if k not in signs and ".se_1." not in k[0]]
    assert not missing and all(k in pre64 for k in signs), (missing[:3], [k for k in signs if k not in pre64][:3])
    force, report = K.flips(pre64, signs)
    assert len(report) <= 8 and all(r[0] < 2e-6 for r in report), f"activation signs that differ from float64: {report[:10]}"
    if force:
        _, g64, _, _ = K.run(*meta, dtype=torch.float64, force=force)
    grads = {k: p.grad for k, p in net.named_parameters()}
    worst, where = K.worst_deviation(grads, g64)
    assert worst < 5e-6, f"largest gradient deviation from the float64 oracle ({len(report)} forced signs: {report}): {worst:.2e} at {where}"


def test_bf16x3_compute_path_stays_inside_the_fp32_bar(golden_dir):
    """compute_dtype 'bf16x3' (opt-in): 3x3 forward / input-gradient products as three bf16 MFMAs (2^-16 per product), fp32
    tensors and accumulation.  Outputs still meet the north-star bar against the reference's fp32 fixture (rtol 1e-3 /
    atol 1e-4), loss to 1e-5, every gradient tensor to 1 % of its norm; PSNR against the fp32 output > 90 dB."""
    from refid_amd.archs import define_network
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "full26_train")
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                              base_num_channels=base, num_block=1, num_residual_blocks=2, compute_dtype="bf16x3"))
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    assert any(getattr(o, "split", 0) == 3 for o in net.engine.all_ops)
    pred = net(x=x.cuda(), event=ev.cuda())
    ref = torch.from_numpy(z["out"])
    np.testing.assert_allclose(pred.detach().cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-4)
    assert O.psnr_between(pred.detach().cpu(), ref) > 90.0
    loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-5)
    gn = np.array([float(p.grad.norm()) for _, p in net.named_parameters()])
    ref_gn = z["grad_norms_all"]
    big = ref_gn > 1e-3 * ref_gn.max()
    np.testing.assert_allclose(gn[big], ref_gn[big], rtol=0.01)
    assert int((gn == 0).sum()) == 13


def test_gradient_accumulation_over_two_backward_calls(golden_dir):
    """gradients delivered through autograd ACCUMULATE like any module's -- including the EGACA
    convs whose beta/gamma are folded into the packed weights (their un-fold must never rescale what is already
    accumulated).  Two different batches, no zero_grad in between == the oracle's sum of the two gradients; and the
    engine-level contract 'backward accumulates into the arena' holds as well."""
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "tiny26_train")
    assert float(P["encoders_forward.1.atten_fuse.beta"].abs().max()) > 0        # fold is really active
    x2, ev2, gt2 = O.make_inputs(x.shape[0], ev.shape[1], x.shape[2], x.shape[3], img_chn, seed=77, mode="hash")
    refs = []
    for a, b, c in ((x, ev, gt), (x2, ev2, gt2)):
        Pc = {k: v.clone() for k, v in P.items()}
        refs.append(O.train_step(Pc, O.TrainState(Pc), a, b, c)[2])
    want = {k: refs[0][k].double() + refs[1][k].double() for k in P}
    net = build(img_chn, base, P)
    for a, b, c in ((x, ev, gt), (x2, ev2, gt2)):
        pred = net(x=a.cuda(), event=b.cuda())
        torch.sqrt((pred - c.cuda()) ** 2 + 1e-12).mean().backward()
    _grad_check(net, P, want)
    # engine level: two backward() calls on the same saved batch without zero_grad() == 2x one call
    eng = net.engine
    gpred = torch.full_like(pred, 1e-3)
    eng.zero_grad()
    eng.forward(x.cuda(), ev.cuda(), save=True)
    eng.backward(gpred)
    one = eng.arena.flat_g.clone()
    eng.forward(x.cuda(), ev.cuda(), save=True)
    eng.backward(gpred)
    two = eng.arena.flat_g
    scale = float(one.abs().max())
    assert float((two - 2 * one).abs().max()) <= 1e-5 * scale
    for k in ("encoders_forward.1.atten_fuse.beta", "encoders_backward.1.atten_fuse.gamma",
              "encoders_forward.1.atten_fuse.conv3.weight", "encoders_backward.1.atten_fuse.conv5.bias"):
        o, n = eng.arena.offsets[k]
        assert float(one[o:o + n].abs().max()) > 0, k
        np.testing.assert_allclose(two[o:o + n].cpu().numpy(), 2 * one[o:o + n].cpu().numpy(), rtol=1e-4,
                                   atol=1e-6 * scale, err_msg=k)


def test_parameter_hooks_fire_and_grads_do_not_alias_the_arena(golden_dir):
    """Parameter gradients arrive through autograd (refid_amd/autograd.py): per-parameter hooks fire -- which is what
    DistributedDataParallel's reducer hangs on -- and p.grad never aliases the engine's arena (the next backward
    overwrites the arena)."""
    z, P, x, ev, gt, img_chn, base = load(golden_dir, "tiny6_train")
    net = build(img_chn, base, P)
    fired = []
    for k, p in net.named_parameters():
        p.register_hook(lambda g, k=k: fired.append(k))
    pred = net(x=x.cuda(), event=ev.cuda())
    pred.mean().backward()
    assert sorted(fired) == sorted(k for k, _ in net.named_parameters())
    lo = net.engine.arena.flat_g.data_ptr()
    hi = lo + net.engine.arena.flat_g.numel() * 4
    assert all(not (lo <= p.grad.data_ptr() < hi) for p in net.parameters())
    net.requires_grad_(False)
    net.pred.conv2d.weight.requires_grad_(True)
    net.zero_grad(set_to_none=True)
    net(x=x.cuda(), event=ev.cuda()).mean().backward()
    assert [k for k, p in net.named_parameters() if p.grad is not None] == ["pred.conv2d.weight"]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16x3"])
def test_batched_weight_packing_equals_the_one_by_one_calls(dtype):
    """Engine.repack(): every packed copy of the weights (direct / Winograd / Winograd x six / split planes / folded biases) written
    by ONE launch (refid_pack_batch) is bit-identical to the per-packing calls."""
    import torch
    from refid_amd import engine as E
    from refid_amd.archs.final_bidirection_attenfusion_arch import FinalBidirectionAttenfusion
    torch.manual_seed(3)
    net = FinalBidirectionAttenfusion(img_chn=6, ev_chn=2, out_chn=3, num_encoders=3, base_num_channels=16, num_block=1,
                                      num_residual_blocks=2, compute_dtype=dtype).cuda()
    eng = net.engine
    names = ("wp", "wd", "wpp6", "wdp6", "wp6", "wd6", "wps", "wds", "b_eff")

    def snapshot():
        out = []
        for o in eng.all_ops:
            for nm in names:
                t = getattr(o, nm, None)
                if t is not None and t is not o.b:
                    out.append(t.clone())
        return out

    old = E.PACK_BATCH
    try:
        E.PACK_BATCH = True
        eng.mark_params_changed(); eng.repack()
        lazy = [o for o in eng.all_ops if o.wp_lazy or o.wd_lazy]
        assert dtype != "fp32" or lazy or not E.WINO6          # fp32 Winograd packings of the 16-bit-pipe Winograd convs: on demand
        for o in lazy:
            o.pack_fallbacks()
        a = snapshot()
        for o in eng.all_ops:                       # poison, then the one-by-one path
            for nm in names:
                t = getattr(o, nm, None)
                if t is not None and t is not o.b:
                    t.view(torch.uint8).fill_(0x5a)
        E.PACK_BATCH = False
        eng.mark_params_changed(); eng.repack()
        b = snapshot()
    finally:
        E.PACK_BATCH = old
    assert len(a) == len(b) and len(a) > 100
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
