"""The CPU oracle (oracle/refid_oracle.py) against the fixtures produced by the
reference's own code (oracle/make_golden.py -> tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import refid_oracle as O

RTOL, ATOL = 1e-4, 1e-5     # oracle vs reference: same torch ops, only graph structure differs


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    img_chn, base, B, T, H, W, seed = [int(v) for v in z["meta"][:7]]
    nb = int(z["meta"][7]) if len(z["meta"]) > 7 else 1                   # num_block, num_encoders (fixtures of round 6 on)
    ne = int(z["meta"][8]) if len(z["meta"]) > 8 else 3
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed, num_block=nb, num_encoders=ne)
    x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash")
    O_kw.clear(); O_kw.update(num_encoders=ne)
    return z, P, x, ev, gt, base


O_kw = {}          # forward / train_step keywords of the fixture loaded last (num_encoders)


def test_param_inventory_matches_reference_counts():
    # 15 928 355 / 15 912 355 / 15 909 955 parameters, 183 tensors
    for chn, n in ((26, 15928355), (6, 15912355), (3, 15909955)):
        sh = O.param_shapes(chn)
        assert len(sh) == 183
        assert sum(int(np.prod(s)) for s in sh.values()) == n


@pytest.mark.parametrize("name", ["tiny26_train", "tiny6_train", "tiny26_nb2_train", "tiny26_default_ctor_train", "tiny6_ne2_train"])
def test_forward_taps_and_train_step(golden_dir, name):
    """(tiny26_nb2: num_block = 2; tiny26_default_ctor: the reference ctor's defaults num_encoders = 4, num_block = 3; tiny6_ne2:
    two levels on a 24 x 40 image -- H, W need only be multiples of 2^num_encoders.)"""
    z, P, x, ev, gt, base = _load(golden_dir, name)
    taps = {}
    with torch.no_grad():
        out = O.forward(P, x, ev, taps=taps if O_kw["num_encoders"] == 3 else None, **O_kw)
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=RTOL, atol=ATOL)
    n_taps = 0
    for k in z.files:
        if k.startswith("tap/"):
            np.testing.assert_allclose(taps[k[4:]].numpy(), z[k], rtol=RTOL, atol=ATOL, err_msg=k)
            n_taps += 1
    assert n_taps >= 25 or not any(k.startswith("tap/") for k in z.files)
    st = O.TrainState(P)
    loss, gnorm, grads, _ = O.train_step(P, st, x, ev, gt, **O_kw)
    np.testing.assert_allclose(loss.numpy(), z["loss"], rtol=1e-5)
    np.testing.assert_allclose(float(gnorm), float(z["grad_norm"]), rtol=1e-4)
    keys = list(P.keys())
    gn = np.array([float(grads[k].norm()) for k in keys])
    np.testing.assert_allclose(gn, z["grad_norms_all"], rtol=2e-4, atol=1e-7)
    zero = [k for k, v in zip(keys, z["grad_norms_all"]) if v == 0.0]
    assert len(zero) == 13          # 13 tensors only ever see zero gradients
    for k in z.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(grads[k[5:]].numpy(), z[k], rtol=1e-3, atol=1e-6, err_msg=k)
        if k.startswith("gradsub7/"):
            np.testing.assert_allclose(grads[k[9:]].flatten()[::7].numpy(), z[k], rtol=1e-3, atol=1e-6)
        if k.startswith("after_step/"):
            np.testing.assert_allclose(P[k[11:]].numpy(), z[k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_full_width_train_step(golden_dir):
    z, P, x, ev, gt, base = _load(golden_dir, "full26_train")
    st = O.TrainState(P)
    loss, gnorm, grads, pred = O.train_step(P, st, x, ev, gt)
    np.testing.assert_allclose(pred.numpy(), z["out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss.numpy(), z["loss"], rtol=1e-5)
    np.testing.assert_allclose(float(gnorm), float(z["grad_norm"]), rtol=1e-4)
    gn = np.array([float(grads[k].norm()) for k in P.keys()])
    np.testing.assert_allclose(gn, z["grad_norms_all"], rtol=5e-4, atol=1e-7)
    for k in z.files:
        if k.startswith("after_step/"):
            np.testing.assert_allclose(P[k[11:]].numpy(), z[k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_odd_sizes_and_config1(golden_dir):
    z, P, x, ev, gt, _ = _load(golden_dir, "odd26_fwd")
    with torch.no_grad():
        out = O.forward(P, x, ev)
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=RTOL, atol=ATOL)
    z, P, x, ev, gt, _ = _load(golden_dir, "config1_fwd")      # BASELINE config 1
    with torch.no_grad():
        out = O.forward(P, x, ev)
    assert tuple(out.shape) == (1, 4, 3, 128, 128)
    np.testing.assert_allclose(out[..., ::4, ::4].numpy(), z["out_sub"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out.double().abs().mean().item(), float(z["out_mean_abs"]), rtol=1e-5)


def test_h_not_multiple_of_8_raises(golden_dir):
    assert bool(np.load(os.path.join(golden_dir, "negative.npz"))["h100_raises"])
    P = O.make_params(26, base_num_channels=8)
    x, ev, _ = O.make_inputs(1, 2, 100, 96, 26)
    with pytest.raises(RuntimeError):
        O.forward(P, x, ev)


def test_psnr_known_answers():
    # metrics/psnr_ssim.py:48-63 + utils/img_util.py:90-117 restated; hand-computed answers
    a = torch.full((3, 8, 8), 0.5)
    b = a + 10.0 / 255.0
    ia, ib = O.tensor2img_u8(a), O.tensor2img_u8(b)
    assert int(ia[0, 0, 0]) == 128 and int(ib[0, 0, 0]) == 138     # round(127.5)=128 (banker's: 128)
    assert abs(O.psnr_u8(ia, ib) - 20 * np.log10(255.0 / 10.0)) < 1e-9
    assert O.psnr_u8(ia, ia) == float("inf")
    assert int(O.tensor2img_u8(torch.full((3, 2, 2), 1.7))[0, 0, 0]) == 255
    assert int(O.tensor2img_u8(torch.full((3, 2, 2), -0.3))[0, 0, 0]) == 0


def test_cosine_lr_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=1e-7)
    for i in range(1, 20):
        opt.step(); sch.step()
        assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(2e-4, i, 50, 1e-7)) < 1e-12


# ---------------------------------------------------------------------------------------------
# SingleMultiConnectEVHINet restatement vs the reference class's own outputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["evhinet_tiny_train", "evhinet_odd_train", "evhinet_full_fwd"])
def test_evhinet_oracle_matches_reference(golden_dir, name):
    from oracle import evhinet_oracle as E
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    wf, B, H, W, seed = [int(v) for v in z["meta"]]
    P = E.make_params(seed=seed, wf=wf)
    x, ev, gt = E.make_inputs(B, H, W, seed=seed)
    train = "loss" in z.files
    if train:
        for p in P.values():
            p.requires_grad_(True)
    out = E.forward(P, x, ev)
    sub = 2 if name == "evhinet_full_fwd" else 1
    np.testing.assert_allclose(out.detach().numpy()[..., ::sub, ::sub], z["out"], rtol=1e-4, atol=1e-5)
    if not train:
        return
    loss = E.psnr_loss(out, gt)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    dead = set(str(k) for k in z["no_grad_keys"])
    for k, p in P.items():
        ref = z["grad/" + k]
        if k in dead:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(p.grad.numpy() - ref).max()) <= 2e-4 * scale + 1e-9, k
    # the dead set is what the module docstring says it is
    live_prefixes = ("conv_ev1", "down_path_ev.0", "down_path_ev.1", "conv_01", "down_path_1", "skip_conv_1",
                     "up_path_1", "sam12.conv2")
    for k in P:
        if not k.startswith(live_prefixes):
            assert k in dead, k
    assert "down_path_ev.1.downsample.weight" in dead
