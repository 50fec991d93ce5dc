#!/usr/bin/env python3
"""How far are the HIP path's outputs / gradients from the oracle on the committed train fixtures?  Prints the largest
output deviation and the largest per-tensor gradient deviation (max |g - ref| / max |ref|) per fixture and compute dtype --
the numbers the tolerances in tests/test_hip_network.py are set from."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import refid_oracle as O
from refid_amd.archs import define_network

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

for name in ("tiny26_train", "tiny6_train", "full26_train"):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    img_chn, base, B, T, H, W, seed = [int(v) for v in z["meta"]]
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed)
    x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash")
    Pc = {k: v.clone() for k, v in P.items()}
    loss_ref, gnorm_ref, grads_ref, _ = O.train_step(Pc, O.TrainState(Pc), x, ev, gt)
    # the same step in float64 = the exact answer both fp32 computations approximate
    P64 = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=seed, dtype=torch.float64)
    x64, ev64, gt64 = O.make_inputs(B, T, H, W, img_chn, seed=seed, mode="hash", dtype=torch.float64)
    _, _, g64, _ = O.train_step({k: v.clone() for k, v in P64.items()}, O.TrainState(P64), x64, ev64, gt64)
    ref_err = {k: float((grads_ref[k].double() - g64[k]).abs().max()) / max(float(g64[k].abs().max()), 1e-30) for k in g64}
    for dt in ("fp32", "bf16x3", "bf16"):
        net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                                  base_num_channels=base, num_block=1, num_residual_blocks=2, compute_dtype=dt))
        net.load_state_dict(P, strict=True)
        net = net.cuda()
        pred = net(x=x.cuda(), event=ev.cuda())
        loss = torch.sqrt((pred - gt.cuda()) ** 2 + 1e-12).mean()
        loss.backward()
        out_err = float((pred.detach().cpu() - torch.from_numpy(z["out"])).abs().max())
        worst = []
        for k, p in net.named_parameters():
            r = grads_ref[k].double()
            s = float(r.abs().max())
            if s > 0:
                worst.append((float((p.grad.double().cpu() - r).abs().max()) / s, k))
        worst.sort(reverse=True)
        ratio = []
        for k, p in net.named_parameters():
            s64 = float(g64[k].abs().max())
            if s64 > 0:
                e = float((p.grad.double().cpu() - g64[k]).abs().max()) / s64
                ratio.append((e / max(ref_err[k], 1e-9), e, ref_err[k], k))
        ratio.sort(reverse=True)
        print(f"    vs float64 oracle: worst (hip err / fp32-oracle err) {ratio[0][0]:.2f} (hip {ratio[0][1]:.1e}, oracle32 {ratio[0][2]:.1e}, "
              f"{ratio[0][3]});  largest hip err {max(r[1] for r in ratio):.1e}, largest oracle32 err {max(r[2] for r in ratio):.1e}")
        print(f"{name:14s} {dt:7s} out max|d| {out_err:.2e}  loss rel {abs(loss.item() - float(z['loss'])) / float(z['loss']):.1e}  "
              f"grad worst {worst[0][0]:.2e} ({worst[0][1]})  median {worst[len(worst) // 2][0]:.2e}", flush=True)
