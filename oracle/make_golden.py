#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE's own code (this container only).

Test infrastructure.  Imports the reference's three hot-path files
(archs/XXNet_final_attenfusion_arch.py, archs/recurrent_sub_modules.py,
archs/fusion_modules.py) and losses/losses.py straight from /root/reference
-- with import stubs for the packages the image lacks (torchvision, the
basicsr.utils logger) and WITHOUT copying or editing any reference source --
runs them on closed-form inputs/weights (oracle.refid_oracle.hash_fill) and
stores inputs' recipe + expected outputs as small fixtures.

The fixtures are data; they travel to the GPU box, the reference does not.
Run:  python oracle/make_golden.py            (writes tests/golden/)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle import refid_oracle as O  # noqa: E402


def import_reference():
    """SURVEY.md 8(c) recipe: pre-seed sys.modules so only the hot-path files load."""
    sys.dont_write_bytecode = True

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    pkg("basicsr", f"{REF}/basicsr")
    pkg("basicsr.models", f"{REF}/basicsr/models")
    pkg("basicsr.models.archs", f"{REF}/basicsr/models/archs")
    pkg("basicsr.models.losses", f"{REF}/basicsr/models/losses")
    utils = types.ModuleType("basicsr.utils")
    import logging
    utils.get_root_logger = lambda *a, **k: logging.getLogger("basicsr")
    sys.modules["basicsr.utils"] = utils
    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo
    arch = importlib.import_module("basicsr.models.archs.XXNet_final_attenfusion_arch")
    losses = importlib.import_module("basicsr.models.losses.losses")
    return arch, losses


def build_ref(arch, img_chn, base, num_block=1, num_encoders=3):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):      # the ctor prints
        net = arch.FinalBidirectionAttenfusion(img_chn=img_chn, ev_chn=2, num_encoders=num_encoders,
                                               base_num_channels=base, num_block=num_block,
                                               num_residual_blocks=2)
    return net


def run_case(arch, losses, name, img_chn, base, B, T, H, W, seed, train, taps_wanted, out_dir,
             store_output="full", num_block=1, num_encoders=3):
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed, num_block=num_block, num_encoders=num_encoders)
    net = build_ref(arch, img_chn, base, num_block, num_encoders)
    assert not taps_wanted or num_encoders == 3, "the tap hooks below are written for three levels"
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state-dict key order/naming mismatch"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    net.load_state_dict(P, strict=True)
    x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash")

    rec = {}
    calls = {}

    def hook(tag, pick=None, when=None):
        def fn(mod, inp, out):
            n = calls.get(tag, 0)
            calls[tag] = n + 1
            o = out if pick is None else out[pick]
            if when is None or n in when:
                key = tag if when is None else f"{tag}_call{n}"
                rec[key] = o.detach().clone()
            rec["_last_" + tag] = o.detach().clone()
        return fn

    hs = []
    if taps_wanted:
        hs.append(net.head_img.register_forward_hook(hook("head")))
        hs.append(net.head.register_forward_hook(hook("e")))
        for i in range(3):
            hs.append(net.img_encoders[i].register_forward_hook(hook(f"x_block{i}")))
            hs.append(net.encoders_backward[i].register_forward_hook(hook(f"bstate{i}", 1, ())))
            hs.append(net.encoders_forward[i].register_forward_hook(hook(f"fwd_out{i}", 0, (0, T - 1))))
            hs.append(net.encoders_forward[i].register_forward_hook(hook(f"fwd_state{i}", 1, (0, T - 1))))
            hs.append(net.decoders[i].register_forward_hook(hook(f"dec_state{i}", 1, ())))
        hs.append(net.encoders_forward[1].atten_fuse.register_forward_hook(hook("egaca_out", None, ())))
        hs.append(net.encoders_forward[1].atten_fuse.se_1.register_forward_hook(hook("egaca_se", None, ())))
        hs.append(net.resblocks[1].register_forward_hook(hook("bottleneck", None, ())))

    out = {}
    net.train()
    if train:
        # the reference train step: twoImage_event_recurrent_model.py:273-310 with the
        # shipped optimiser settings (options/train/GoPro/*.yml: AdamW 2e-4, wd 1e-4, betas .9/.99)
        opt = torch.optim.AdamW([{"params": list(net.parameters())}], lr=2e-4, weight_decay=1e-4,
                                betas=(0.9, 0.99))
        cri = losses.CharbonnierLoss(loss_weight=1, reduction="mean")
        opt.zero_grad()
        pred = net(x=x, event=ev)
        l_total = cri(pred, gt)
        l_total = l_total + 0 * sum(p.sum() for p in net.parameters())
        l_total.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.01)
        # (grads are now clipped in place; store the unclipped ones via the norm)
        coef = min(1.0, 0.01 / (float(gnorm) + 1e-6))
        named = dict(net.named_parameters())
        out["loss"] = l_total.detach().numpy()
        out["grad_norm"] = np.float32(float(gnorm))
        pick = ["head.conv2d.weight", "head_img.conv2d.bias", "pred.conv2d.weight",
                "encoders_forward.0.recurrent_block.forward_trunk.main.0.weight",
                "encoders_backward.2.recurrent_block.forward_trunk.main.2.0.conv2.bias",
                "encoders_forward.1.atten_fuse.beta", "encoders_forward.1.atten_fuse.gamma",
                "encoders_backward.1.atten_fuse.norm1_e.weight",
                "encoders_forward.1.atten_fuse.conv2_e.weight",
                "encoders_forward.1.atten_fuse.se_1.1.weight",
                "encoders_forward.2.fuse_two_dir.conv2d.weight",
                "encoders_forward.0.down.weight",
                "decoders.0.transposed_conv2d.weight", "decoders.2.transposed_conv2d.bias",
                "img_encoders.1.identity.weight", "resblocks.0.conv1.weight",
                # three of the 13 parameters that only ever get zero gradients
                "encoders_forward.1.conv.conv2d.weight", "encoders_backward.2.down.weight",
                "encoders_forward.1.atten_fuse.se_2.1.bias"]
        for k in pick:
            if k not in named:                               # (a level that this num_encoders does not have)
                continue
            g = named[k].grad.detach() / coef                # undo the in-place clip
            if g.numel() > 50000:                            # keep fixtures small
                out["gradsub7/" + k] = g.flatten()[::7].numpy().astype(np.float32)
            else:
                out["grad/" + k] = g.numpy().astype(np.float32)
        # per-parameter gradient L2 norms for ALL parameters (cheap, pins every wgrad)
        out["grad_norms_all"] = np.array([float(named[k].grad.norm()) / coef for k in P.keys()],
                                         dtype=np.float64)
        opt.step()
        for k in ["pred.conv2d.weight", "encoders_forward.1.conv.conv2d.weight",
                  "encoders_forward.0.recurrent_block.forward_trunk.main.0.bias"]:
            out["after_step/" + k] = named[k].detach().numpy().astype(np.float32)
        pred = pred.detach()
    else:
        net.eval()
        with torch.no_grad():
            pred = net(x=x, event=ev)
    for h in hs:
        h.remove()

    if store_output == "full":
        out["out"] = pred.numpy().astype(np.float32)
    else:   # strided sample + moments
        out["out_sub"] = pred[..., ::store_output, ::store_output].numpy().astype(np.float32)
        out["out_mean_abs"] = np.float64(pred.double().abs().mean().item())
        out["out_sum"] = np.float64(pred.double().sum().item())
    if taps_wanted:
        ren = {"head": "head", "x_block0": "x_block0", "x_block1": "x_block1", "x_block2": "x_block2"}
        for k, v in ren.items():
            out["tap/" + v] = rec[k].numpy()
        out["tap/e"] = rec["e"].reshape(B, T, -1, H, W).numpy()
        for i in range(3):
            out[f"tap/final_bstate{i}"] = rec[f"_last_bstate{i}"].numpy()
            for t in (0, T - 1):
                out[f"tap/fwd_out{i}_t{t}"] = rec[f"fwd_out{i}_call{t}"].numpy()
                out[f"tap/fwd_state{i}_t{t}"] = rec[f"fwd_state{i}_call{t}"].numpy()
            out[f"tap/dec_state{i}_tlast"] = rec[f"_last_dec_state{i}"].numpy()
        out["tap/egaca_out"] = rec["_last_egaca_out"].numpy()
        out["tap/egaca_se"] = rec["_last_egaca_se"].numpy()
        out["tap/bottleneck_tlast"] = rec["_last_bottleneck"].numpy()
    out["meta"] = np.array([img_chn, base, B, T, H, W, seed] + ([num_block] if (num_block, num_encoders) != (1, 3) else []) +
                           ([num_encoders] if num_encoders != 3 else []), dtype=np.int64)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.0f} KiB), "
          f"out mean|.|={pred.abs().mean():.4f}")


def run_evhinet(losses, name, wf, B, H, W, seed, train, out_dir, sub=1):
    """the reference's SingleMultiConnectEVHINet on closed-form weights / inputs.
    arch_util.py imports timm (absent here) for classes this path never touches: stub the three names."""
    from oracle import evhinet_oracle as E
    for nm in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    lay = sys.modules["timm.models.layers"]
    lay.DropPath, lay.trunc_normal_, lay.to_2tuple = torch.nn.Identity, torch.nn.init.trunc_normal_, (lambda v: (v, v))
    mod = importlib.import_module("basicsr.models.archs.single_multiconnect_evhinet_arch")
    net = mod.SingleMultiConnectEVHINet(wf=wf)
    P = E.make_params(seed=seed, wf=wf)
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state-dict keys/order differ from oracle.evhinet_oracle.param_shapes"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    net.load_state_dict(P, strict=True)
    x, ev, gt = E.make_inputs(B, H, W, seed=seed)
    out = net(x=x, event=ev)
    assert isinstance(out, list) and len(out) == 1
    rec = {"meta": np.array([wf, B, H, W, seed]), "out": out[0].detach().numpy()[..., ::sub, ::sub]}
    if train:
        loss = losses.PSNRLoss()(out[0], gt)
        loss.backward()
        rec["loss"] = np.array(float(loss))
        for k, p in net.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            rec["grad/" + k] = g.numpy()
        rec["no_grad_keys"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(name, "out mean", float(out[0].mean()), "loss", float(rec.get("loss", np.nan)))


def run_host_logic(arch, out_dir):
    """Host-side contracts pinned to the reference's own code:
    (1) A11 initialisation (recurrent_sub_modules.py:752-753,776-804 default_init_weights on ResidualBlockNoBN, torch
        defaults elsewhere, LN 1/0, beta/gamma 0): per-parameter moments of the reference ctor at torch seed 0;
    (2) learning-rate sequences of the schedulers base_model.py:77-108 can build (models/lr_scheduler.py + torch
        CosineAnnealingLR), stepped the way base_model.py:158-180 does (step() from iteration 2, linear warm-up that
        overwrites the group's lr with initial_lr / warmup_iter * iter)."""
    torch.manual_seed(0)
    net = build_ref(arch, 26, 32)
    keys, mom = [], []
    for k, p in net.state_dict().items():
        v = p.double().flatten()
        keys.append(k)
        mom.append([v.numel(), float(v.mean()), float(v.std(unbiased=False)), float(v.abs().max()), float(v.min()),
                    float(v.max())])
    rec = {"init_keys": np.array(keys), "init_moments": np.array(mom, dtype=np.float64)}
    lrs = importlib.import_module("basicsr.models.lr_scheduler")
    cases = {
        "TrueCosineAnnealingLR": (dict(T_max=30, eta_min=1e-7), -1, None),
        "TrueCosineAnnealingLR_warm": (dict(T_max=30, eta_min=1e-7), 6, None),
        "MultiStepLR": (dict(milestones=[5, 5, 12, 20], gamma=0.5), -1, None),
        "MultiStepRestartLR_warm": (dict(milestones=[8, 16, 30], gamma=0.5, restarts=[0, 20], restart_weights=[1, 0.5]), 4, None),
        "CosineAnnealingRestartLR": (dict(periods=[10, 10, 20], restart_weights=[1, 0.5, 0.25], eta_min=1e-7), -1, None),
        "LinearLR": ({}, -1, 40),
        "VibrateLR": ({}, -1, 800),                      # lr_scheduler.py:71-112: T = total_iter // 80 = 10, Th = 5
        "VibrateLR_warm": ({}, 5, 1600),
    }
    for name, (cfg, warm, total) in cases.items():
        kind = name.split("_")[0]
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=2e-4)
        if kind == "TrueCosineAnnealingLR":
            sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, **cfg)
        elif kind in ("MultiStepLR", "MultiStepRestartLR"):
            sch = lrs.MultiStepRestartLR(opt, **cfg)
        elif kind == "CosineAnnealingRestartLR":
            sch = lrs.CosineAnnealingRestartLR(opt, **cfg)
        elif kind == "VibrateLR":
            sch = lrs.VibrateLR(opt, total)
        else:
            sch = lrs.LinearLR(opt, total)
        seq = []
        for it in range(1, 40):
            if it > 1:                                   # base_model.py:166-168
                opt.step()
                sch.step()
            if it < warm:                                # base_model.py:170-180
                for g in opt.param_groups:
                    g["lr"] = g["initial_lr"] / warm * it
            seq.append(opt.param_groups[0]["lr"])
        rec["lr/" + name] = np.array(seq, dtype=np.float64)
        rec["lrcfg/" + name] = np.array(repr((kind, cfg, warm, total)))
    np.savez_compressed(os.path.join(out_dir, "host_logic.npz"), **rec)
    print("host_logic: init moments of", len(keys), "tensors;", len(cases), "lr sequences")


def run_metrics(out_dir):
    """the validation tail as the reference computes it -- utils/img_util.py:59-121 (tensor2img) and
    metrics/psnr_ssim.py:9-63 (calculate_psnr), :135-182,225-303 (calculate_ssim -> _ssim_3d) -- imported from
    /root/reference.  The image lacks cv2 / skimage / torchvision.utils and a GPU, so: cv2 is a stub holding the two
    functions those line ranges call (getGaussianKernel = OpenCV's closed form for ksize > 7 / sigma > 0:
    exp(-(i-(n-1)/2)^2 / (2 sigma^2)) normalised to sum 1, as an (n,1) float64 column; cvtColor(RGB2BGR) = channel
    flip), skimage is an empty stub (only referenced in comments), and Tensor.cuda()/Module.cuda() are identity for the
    duration of the call (the reference runs the same fp32 Conv3d on its GPU).  No reference source is edited or copied."""
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_RGB2BGR = 4

    def get_gaussian_kernel(ksize, sigma):
        assert ksize > 7 and sigma > 0            # below that OpenCV switches to fixed tables / derived sigma
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(x * x) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)

    def cvt_color(img, code):
        assert code == cv2.COLOR_RGB2BGR and img.shape[2] == 3
        return np.ascontiguousarray(img[:, :, ::-1])

    cv2.getGaussianKernel, cv2.cvtColor = get_gaussian_kernel, cvt_color
    sys.modules["cv2"] = cv2
    for nm in ("skimage", "skimage.metrics"):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = None                          # only reached for 4-D mini-batches
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision"].utils = tvu
    utils = sys.modules["basicsr.utils"]
    utils.__path__ = [f"{REF}/basicsr/utils"]      # submodules (matlab_functions, img_util) load from the reference
    m = types.ModuleType("basicsr.metrics")
    m.__path__ = [f"{REF}/basicsr/metrics"]
    sys.modules["basicsr.metrics"] = m
    img_util = importlib.import_module("basicsr.utils.img_util")
    ps = importlib.import_module("basicsr.metrics.psnr_ssim")

    rec = {}
    cases = {"a": (40, 48, 21), "b": (32, 56, 22), "c": (24, 24, 23)}
    saved = (torch.Tensor.cuda, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        for name, (h, w, salt) in cases.items():
            gt = (O.hash_fill((3, h, w), salt) * 0.5 + 0.5).float()
            noise = O.hash_fill((3, h, w), salt + 100)
            pred = (gt + (0.35 if name == "c" else 0.06) * noise).float()     # leaves [0,1]: exercises the clamp
            if name == "c":
                pred[:, :4] = gt[:, :4]                                        # a band of identical rows
            sr_img = img_util.tensor2img([pred.clone()])                       # the call of :429 (uint8, BGR)
            gt_img = img_util.tensor2img([gt.clone()])                         # :432
            rec[f"{name}/pred"], rec[f"{name}/gt"] = pred.numpy(), gt.numpy()
            rec[f"{name}/pred_u8_bgr"], rec[f"{name}/gt_u8_bgr"] = sr_img, gt_img
            rec[f"{name}/psnr"] = np.float64(ps.calculate_psnr(sr_img, gt_img, crop_border=0))
            rec[f"{name}/ssim"] = np.float64(ps.calculate_ssim(sr_img, gt_img, crop_border=0))
            print(f"metrics {name}: psnr {float(rec[f'{name}/psnr']):.4f} ssim {float(rec[f'{name}/ssim']):.6f}")
        same = img_util.tensor2img([gt.clone()])
        rec["same/psnr"] = np.float64(ps.calculate_psnr(same, same, crop_border=0))      # inf
        rec["same/ssim"] = np.float64(ps.calculate_ssim(same, same, crop_border=0))
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = saved
    rec["names"] = np.array(list(cases))
    np.savez_compressed(os.path.join(out_dir, "metrics.npz"), **rec)


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)
    arch, losses = import_reference()
    run_host_logic(arch, out_dir)
    if os.environ.get("ONLY") == "host":
        return
    if os.environ.get("ONLY") == "metrics":
        run_metrics(out_dir)
        return
    if os.environ.get("ONLY") == "nb2":
        run_case(arch, losses, "tiny26_nb2_train", 26, 8, 2, 3, 32, 32, 7, True, True, out_dir, num_block=2)
        return
    if os.environ.get("ONLY") == "ne":
        run_case(arch, losses, "tiny26_default_ctor_train", 26, 8, 2, 3, 32, 32, 8, True, False, out_dir, num_block=3, num_encoders=4)
        run_case(arch, losses, "tiny6_ne2_train", 6, 8, 1, 3, 24, 40, 9, True, False, out_dir, num_encoders=2)
        return
    if os.environ.get("ONLY") != "refid":
        run_evhinet(losses, "evhinet_tiny_train", 8, 2, 32, 32, 1, True, out_dir)
        run_evhinet(losses, "evhinet_odd_train", 16, 1, 40, 24, 2, True, out_dir)
        run_evhinet(losses, "evhinet_full_fwd", 64, 1, 64, 64, 3, False, out_dir, sub=2)
        if os.environ.get("ONLY") == "evhinet":
            return
    # dense taps + full train step, blur-VFI (26 ch) and sharp-VFI (5-D x, 6 ch)
    run_case(arch, losses, "tiny26_train", 26, 8, 2, 3, 32, 32, 1, True, True, out_dir)
    run_case(arch, losses, "tiny6_train", 6, 8, 2, 3, 32, 32, 2, True, True, out_dir)
    # non-square, H/8 odd: 40x24 -> 5x3 at the bottleneck
    run_case(arch, losses, "odd26_fwd", 26, 8, 1, 2, 40, 24, 3, False, False, out_dir)
    # full-width network end to end with a train step
    run_case(arch, losses, "full26_train", 26, 32, 1, 5, 64, 64, 4, True, False, out_dir)
    # two ResidualBlockNoBN per trunk (num_block=2; the YAMLs use 1, the reference ctor's default is 3): round 6
    run_case(arch, losses, "tiny26_nb2_train", 26, 8, 2, 3, 32, 32, 7, True, True, out_dir, num_block=2)
    # the reference ctor's own defaults (num_encoders=4, num_block=3: arch:90-92) and the smallest net (two levels): round 6
    run_case(arch, losses, "tiny26_default_ctor_train", 26, 8, 2, 3, 32, 32, 8, True, False, out_dir, num_block=3, num_encoders=4)
    run_case(arch, losses, "tiny6_ne2_train", 6, 8, 1, 3, 24, 40, 9, True, False, out_dir, num_encoders=2)
    # BASELINE config 1: img_chn=3, 128x128, 5-bin voxel -> T=4, forward only
    run_case(arch, losses, "config1_fwd", 3, 32, 1, 4, 128, 128, 5, False, False, out_dir,
             store_output=4)
    # negative fixture: H=100 must raise in the reference
    net = build_ref(arch, 26, 8)
    x, ev, _ = O.make_inputs(1, 2, 104, 96, 26, seed=6)
    try:
        net(x=x[:, :, :100, :], event=ev[:, :, :, :100, :])
        raised = False
    except Exception as ex:  # noqa: BLE001
        raised = True
        print("H=100 raises in the reference:", type(ex).__name__)
    np.savez(os.path.join(out_dir, "negative.npz"), h100_raises=np.array(raised))
    # event voxelisation: the reference function as shipped uses the removed `np.int` alias
    # (event_util.py:39,44); alias it back for the run -- no source edit.
    if not hasattr(np, "int"):
        np.int = int
    sys.modules["basicsr.utils"].Timer = object
    sys.modules["basicsr.utils"].CudaTimer = object
    sys.modules.setdefault("basicsr.data", types.ModuleType("basicsr.data")).__path__ = [f"{REF}/basicsr/data"]
    eu = importlib.import_module("basicsr.data.event_util")
    rng = np.random.Generator(np.random.PCG64(11))
    for name, (n_ev, bins, hh, ww) in {"voxel_a": (6000, 5, 24, 32), "voxel_b": (3000, 24, 16, 16)}.items():
        t = np.sort(rng.random(n_ev) * 0.05 + 1.5)
        ev = np.stack([t, rng.integers(0, ww, n_ev).astype(np.float64), rng.integers(0, hh, n_ev).astype(np.float64),
                       rng.integers(0, 2, n_ev).astype(np.float64)], axis=1)
        vox = eu.events_to_voxel_grid(ev.copy(), bins, ww, hh)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), events=ev, voxel=vox.astype(np.float32),
                            meta=np.array([bins, hh, ww]))
        print(name, "voxel sum", float(vox.sum()))
    run_metrics(out_dir)


if __name__ == "__main__":
    main()
