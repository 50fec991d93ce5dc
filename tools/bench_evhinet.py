#!/usr/bin/env python3
"""Train-step and inference throughput of the HIP SingleMultiConnectEVHINet on synthetic
256x256 batches; CPU oracle (torch) timed beside it on a bounded sample.  images/s; not the BASELINE metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refid_amd.train import ImageEventRestorationModel

B = int(os.environ.get("B", 8)); S = int(os.environ.get("SIZE", 256))
opt = {"name": "b", "is_train": True, "num_gpu": 1, "network_g": dict(type="SingleMultiConnectEVHINet"),
       "path": {"pretrain_network_g": None},
       "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                 "scheduler": dict(type="TrueCosineAnnealingLR", T_max=1000, eta_min=1e-7),
                 "pixel_opt": dict(type="PSNRLoss", loss_weight=0.5, reduction="mean")}, "val": {}}
torch.manual_seed(0)
m = ImageEventRestorationModel(opt)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(B, 3, S, S, device="cuda", generator=g); ev = torch.randn(B, 6, S, S, device="cuda", generator=g)
gt = torch.rand(B, 3, S, S, device="cuda", generator=g)
m.feed_data({"lq": x, "voxel": ev, "gt": gt})
for it in range(1, 4):
    m.update_learning_rate(it); m.optimize_parameters(it)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
for it in range(4, 4 + n):
    m.update_learning_rate(it); m.optimize_parameters(it)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step  B={B} {S}x{S}: {dt*1e3:7.2f} ms  {B/dt:8.1f} images/s   loss {m.get_current_log()['l_pix']:.4f}")
m.test(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): m.test()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"inference   B={B} {S}x{S}: {dt*1e3:7.2f} ms  {B/dt:8.1f} images/s")
if os.environ.get("CPU", "1") != "0":
    from oracle import evhinet_oracle as E
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.net_g.state_dict().items()}
    xc, ec, gc = x[:1].cpu(), ev[:1].cpu(), gt[:1].cpu()
    t0 = time.perf_counter()
    E.psnr_loss(E.forward(P, xc, ec), gc).backward()
    dt = time.perf_counter() - t0
    print(f"CPU oracle fwd+bwd B=1 ({torch.get_num_threads()} threads): {dt*1e3:.0f} ms  {1/dt:.2f} images/s")
