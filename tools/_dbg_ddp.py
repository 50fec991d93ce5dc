import os, sys, socket
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, torch.multiprocessing as mp
from oracle import refid_oracle as O
import test_hip_ddp as T
if __name__ == "__main__":
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); mgr = ctx.Manager(); ret = mgr.dict()
    mp.spawn(T._worker, args=(2, port, ret), nprocs=2, join=True)
    loss0, gn0, sd0 = ret[0]
    model = TwoImageEventRecurrentRestorationModel(T._opt(26, 8))
    P = O.make_params(26, base_num_channels=8, mode="hash", seed=5)
    model.net_g.load_state_dict(P)
    x, ev, gt = O.make_inputs(2, 3, 32, 32, 26, seed=21, mode="hash")
    for it in (1, 2):
        model.update_learning_rate(it); model.feed_data({"lq": x, "voxel": ev, "gt": gt}); model.optimize_parameters(it)
    print("loss", model.get_current_log()["l_pix"], loss0, "gn", model.grad_norm(), gn0)
    sd = model.net_g.state_dict(); bad = 0
    for k in sd:
        a, b = sd[k].double().cpu(), sd0[k].double()
        disp = (a - P[k].double()).abs().max().item()
        r = (a - b).abs().max().item() / (disp + 1e-12)
        if r > 0.02: print("BAD %.3f %s" % (r, k)); bad += 1
    print("bad keys", bad, "of", len(sd))
