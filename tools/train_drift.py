#!/usr/bin/env python3
"""Longer-horizon parity: N fused HIP train steps vs the CPU oracle's train_step on the same tiny problem
(base 8, 32x32, T=3, B=2); prints both loss curves and the final parameter distance relative to the displacement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import refid_oracle as O
from refid_amd.train import TwoImageEventRecurrentRestorationModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
opt = {"name": "t", "is_train": True, "num_gpu": 1,
       "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3, base_num_channels=8,
                         num_block=1, num_residual_blocks=2),
       "path": {"pretrain_network_g": None},
       "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                 "scheduler": dict(type="TrueCosineAnnealingLR", T_max=200, eta_min=1e-7),
                 "pixel_opt": dict(type="CharbonnierLoss", loss_weight=1, reduction="mean")}, "val": {}}
model = TwoImageEventRecurrentRestorationModel(opt)
P0 = O.make_params(26, base_num_channels=8, mode="hash", seed=5)
model.net_g.load_state_dict(P0)
x, ev, gt = O.make_inputs(2, 3, 32, 32, 26, seed=21, mode="hash")
P = {k: v.clone() for k, v in P0.items()}
st = O.TrainState(P)
hip, ref = [], []
for it in range(1, N + 1):
    model.update_learning_rate(it)
    model.feed_data({"lq": x, "voxel": ev, "gt": gt})
    model.optimize_parameters(it)
    hip.append(model.get_current_log()["l_pix"])
    lr = O.cosine_lr(2e-4, it - 1, 200, 1e-7)
    loss, gnorm, grads, pred = O.train_step(P, st, x, ev, gt, lr=lr, weight_decay=1e-4)
    ref.append(float(loss))
    if it in (1, 2, 5, 10, 20, N):
        print(f"step {it:3d}  hip {hip[-1]:.6f}  oracle {ref[-1]:.6f}  diff {abs(hip[-1]-ref[-1]):.2e}")
sd = model.net_g.state_dict()
worst = 0.0
for k in sd:
    disp = (P[k].double() - P0[k].double()).abs().max().item()
    d = (sd[k].double().cpu() - P[k].double()).abs().max().item()
    if disp > 0: worst = max(worst, d / disp)
print(f"after {N} steps: max |hip - oracle| / max displacement over all parameters = {worst:.3e}")
