"""GPU: the fused train step (S1-S4 mirror, refid_amd/train.py) against the oracle's restatement of
optimize_parameters over several iterations: loss, clipped AdamW trajectory (incl. decoupled weight
decay on the 13 gradient-less tensors), cosine LR stepping, eval chunking, checkpoint round trip."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import refid_oracle as O

pytestmark = pytest.mark.gpu


def _opt(img_chn, base, T_max=50, clip=True, dtype="fp32"):
    return {
        "name": "t", "is_train": True, "num_gpu": 1,
        "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                          base_num_channels=base, num_block=1, num_residual_blocks=2, compute_dtype=dtype),
        "path": {"pretrain_network_g": None},
        "train": {"optim_g": dict(type="AdamW", lr=2e-4, weight_decay=1e-4, betas=[0.9, 0.99]),
                  "scheduler": dict(type="TrueCosineAnnealingLR", T_max=T_max, eta_min=1e-7),
                  "pixel_opt": dict(type="CharbonnierLoss", loss_weight=1, reduction="mean"),
                  "use_grad_clip": clip},
        "val": {"max_minibatch": 2},
    }


def test_three_train_steps_match_oracle():
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    img_chn, base, B, T, H, W = 26, 8, 2, 3, 32, 32
    P = O.make_params(img_chn, base_num_channels=base, mode="hash", seed=5)
    model = TwoImageEventRecurrentRestorationModel(_opt(img_chn, base))
    model.net_g.load_state_dict(P, strict=True)
    Pc = {k: v.clone() for k, v in P.items()}
    st = O.TrainState(Pc)
    epoch = 0
    for it in range(1, 4):
        x, ev, gt = O.make_inputs(B, T, H, W, img_chn, seed=10 + it, mode="hash")
        model.update_learning_rate(it)
        if it > 1:
            epoch += 1
        lr = O.cosine_lr(2e-4, epoch, 50, 1e-7)
        assert abs(model.get_current_learning_rate()[0] - lr) < 1e-15
        model.feed_data({"lq": x, "voxel": ev, "gt": gt})
        model.optimize_parameters(it)
        loss_ref, gnorm_ref, _, _ = O.train_step(Pc, st, x, ev, gt, lr=lr)
        assert abs(model.get_current_log()["l_pix"] - float(loss_ref)) < 2e-5 * abs(float(loss_ref)) + 1e-7
        assert abs(model.grad_norm() - float(gnorm_ref)) < 2e-3 * float(gnorm_ref)
    sd = model.net_g.state_dict()
    worst = 0.0
    for k in P:
        a, b = sd[k].double().cpu(), Pc[k].double()
        # parameters move by ~lr per step (normalised-gradient AdamW): compare the DISPLACEMENT
        disp = (b - P[k].double()).abs().max().item()
        err = (a - b).abs().max().item()
        worst = max(worst, err / max(disp, 1e-12))
        assert err <= 0.05 * disp + 1e-9, (k, err, disp)
    # gradient-less tensors: pure decoupled weight decay p *= (1 - lr*wd) each step
    k = "encoders_forward.1.conv.conv2d.weight"
    np.testing.assert_allclose(sd[k].cpu().numpy(), Pc[k].numpy(), rtol=1e-6, atol=1e-9)


def test_eval_step_chunks_and_checkpoint_roundtrip(tmp_path):
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    opt = _opt(6, 8)
    opt["path"].update(models=str(tmp_path), training_states=str(tmp_path))
    model = TwoImageEventRecurrentRestorationModel(opt)
    P = O.make_params(6, base_num_channels=8, mode="hash", seed=2)
    model.net_g.load_state_dict(P)
    x, ev, gt = O.make_inputs(3, 2, 16, 24, 6, seed=3)
    model.feed_data({"lq": x, "voxel": ev, "gt": gt})
    model.test()                                               # max_minibatch 2 -> chunks of 2 + 1
    with torch.no_grad():
        ref = O.forward(P, x, ev)
    np.testing.assert_allclose(model.output.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-4)
    path = model.save_network(model.net_g, "net_g", 1)         # base_model.py:188-219 signature / file name
    assert path == os.path.join(tmp_path, "net_g_1.pth")
    assert model.save_network(model.net_g, "net_g", -1).endswith("net_g_latest.pth")
    ck = torch.load(path)
    assert list(ck.keys()) == ["params"] and list(ck["params"].keys()) == list(P.keys())
    m2 = TwoImageEventRecurrentRestorationModel({**_opt(6, 8), "path": {"pretrain_network_g": path, "strict_load_g": True}})
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(m2.net_g.state_dict().values(), model.net_g.state_dict().values()))
    # a DDP-style "module." prefixed checkpoint loads too (base_model.py:256-281)
    torch.save({"params": {"module." + k: v for k, v in ck["params"].items()}}, path)
    m3 = TwoImageEventRecurrentRestorationModel({**_opt(6, 8), "path": {"pretrain_network_g": path}})
    assert torch.equal(m3.net_g.state_dict()["pred.conv2d.weight"].cpu(), P["pred.conv2d.weight"])


def test_save_and_resume_training_state(tmp_path):
    """save(epoch, iter) (twoImage_event_recurrent_model.py:552-554) + resume_training (base_model.py:308-323): a run
    resumed from the saved network + state continues on the same trajectory; a `.state` in the REFERENCE's format
    (torch.optim.AdamW.state_dict() + scheduler.state_dict(), base_model.py:297-303) resumes to the same bits as this
    class's own flat-arena format."""
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    opt = _opt(6, 8)
    opt["path"].update(models=str(tmp_path), training_states=str(tmp_path))
    P = O.make_params(6, base_num_channels=8, mode="hash", seed=9)
    x, ev, gt = O.make_inputs(1, 2, 16, 16, 6, seed=4, mode="hash")
    a = TwoImageEventRecurrentRestorationModel(opt)
    a.net_g.load_state_dict(P)
    a.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (1, 2):
        a.update_learning_rate(it)
        a.optimize_parameters(it)
    a.save(0, 2)
    for it in (3, 4):
        a.update_learning_rate(it)
        a.optimize_parameters(it)
    o2 = _opt(6, 8)
    o2["path"].update(pretrain_network_g=os.path.join(tmp_path, "net_g_2.pth"))
    b = TwoImageEventRecurrentRestorationModel(o2)
    b.resume_training(torch.load(os.path.join(tmp_path, "2.state")))
    b.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (3, 4):
        b.update_learning_rate(it)
        b.optimize_parameters(it)
    assert b.get_current_learning_rate() == a.get_current_learning_rate()
    assert b.step_count == a.step_count == 4
    for (k, va), vb in zip(a.net_g.state_dict().items(), b.net_g.state_dict().values()):
        disp = (va.double().cpu() - P[k].double()).abs().max().item()
        assert (va.double() - vb.double()).abs().max().item() <= 0.02 * disp + 1e-9, k
    # the same state written the way the reference writes it: per-parameter torch AdamW state in parameter order
    own = torch.load(os.path.join(tmp_path, "2.state"))
    arena = a.net_g.engine.arena
    st = {}
    for i, (k, (off, n)) in enumerate(arena.offsets.items()):
        st[i] = {"step": torch.tensor(float(own["optimizers"][0]["step"])),
                 "exp_avg": own["optimizers"][0]["exp_avg"][off:off + n].view(arena.shapes[k]).clone(),
                 "exp_avg_sq": own["optimizers"][0]["exp_avg_sq"][off:off + n].view(arena.shapes[k]).clone()}
    ref_state = {"epoch": 0, "iter": 2,
                 "optimizers": [{"state": st, "param_groups": [{"lr": own["schedulers"][0]["lr"], "params": list(range(len(st)))}]}],
                 "schedulers": [{"last_epoch": own["schedulers"][0]["last_epoch"], "_last_lr": [own["schedulers"][0]["lr"]]}]}
    d = TwoImageEventRecurrentRestorationModel(o2)
    d.resume_training(ref_state)
    d.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (3, 4):
        d.update_learning_rate(it)
        d.optimize_parameters(it)
    assert d.step_count == 4
    for (k, vb), vd in zip(b.net_g.state_dict().items(), d.net_g.state_dict().values()):
        assert torch.equal(vb, vd), k                       # no atomics on the path: the same bits
    # ... and written by save_training_state itself in the reference's layout: torch's own AdamW loads it (what basicsr's
    # resume_training does, base_model.py:316-320), and it resumes here to the same bits
    b2 = TwoImageEventRecurrentRestorationModel(o2)
    b2.resume_training(torch.load(os.path.join(tmp_path, "2.state")))
    ref_dir = os.path.join(tmp_path, "ref_layout")
    os.makedirs(ref_dir)
    b2.opt["path"]["training_states"] = ref_dir
    b2.save_training_state(0, 2, reference_layout=True)
    written = torch.load(os.path.join(ref_dir, "2.state"))
    plist = [torch.nn.Parameter(torch.zeros(arena.shapes[k])) for k in arena.offsets]
    # (an optimizer built the way the reference builds it, twoImage_event_recurrent_model.py:88-90: two param groups, the
    #  low-lr one -- 'module.offsets' / 'module.dcns' -- empty for this network; a one-group dict is refused by torch)
    topt = torch.optim.AdamW([{"params": plist}, {"params": [], "lr": 1e-4 * 0.1}], lr=1e-4)
    topt.load_state_dict(written["optimizers"][0])                       # torch accepts it
    assert len(written["optimizers"][0]["param_groups"]) == 2 and written["optimizers"][0]["param_groups"][1]["params"] == []
    tsch = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=100, eta_min=1e-7)
    tsch.load_state_dict(written["schedulers"][0])
    assert len(tsch.base_lrs) == 2 and len(tsch._last_lr) == 2
    assert topt.state_dict()["param_groups"][0]["betas"] == tuple(opt["train"]["optim_g"]["betas"])
    assert torch.equal(topt.state[plist[3]]["exp_avg"], st[3]["exp_avg"]) and float(topt.state[plist[3]]["step"]) == 2.0
    e = TwoImageEventRecurrentRestorationModel(o2)
    e.resume_training(written)
    e.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (3, 4):
        e.update_learning_rate(it)
        e.optimize_parameters(it)
    for (k, vb), ve in zip(b.net_g.state_dict().items(), e.net_g.state_dict().values()):
        assert torch.equal(vb, ve), k
    # a real torch `.state` is SPARSE when some parameter never received a gradient (atten_fuse.se_2 is unused in forward,
    # fusion_modules.py:261 vs :312-315): entries missing -> zero moments; here the se_2 entries are dropped, whose moments
    # are exactly zero anyway (zero gradient), so the resumed run must stay on the same bits
    keys = list(arena.offsets)
    sparse = {i: e for i, e in st.items() if ".se_2." not in keys[i]}
    assert len(sparse) == len(st) - 8
    sp_state = dict(ref_state, optimizers=[{"state": sparse, "param_groups": ref_state["optimizers"][0]["param_groups"]}])
    e = TwoImageEventRecurrentRestorationModel(o2)
    e.exp_avg.fill_(7.0)                                    # stale moments must not survive a resume
    e.resume_training(sp_state)
    assert e.step_count == 2
    e.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (3, 4):
        e.update_learning_rate(it)
        e.optimize_parameters(it)
    for (k, vb), ve in zip(b.net_g.state_dict().items(), e.net_g.state_dict().values()):
        assert torch.equal(vb, ve), k
    bad = {"optimizers": [{"state": {len(st): st[0]}, "param_groups": []}], "schedulers": ref_state["schedulers"]}
    with pytest.raises(ValueError, match="outside this network"):
        d.resume_training(bad)
    bad = {"optimizers": [{"state": {1: st[0]}, "param_groups": []}], "schedulers": ref_state["schedulers"]}
    with pytest.raises(ValueError, match="values, parameter"):
        d.resume_training(bad)
    with pytest.raises(ValueError, match="unknown optimizer"):
        d.resume_training({"optimizers": [{}], "schedulers": ref_state["schedulers"]})
    # a resume that forgets the optimizer state is NOT on that trajectory (the check above has teeth)
    c = TwoImageEventRecurrentRestorationModel(o2)
    c.feed_data({"lq": x, "voxel": ev, "gt": gt})
    for it in (3, 4):
        c.update_learning_rate(it)
        c.optimize_parameters(it)
    k = "pred.conv2d.weight"
    assert (c.net_g.state_dict()[k] - a.net_g.state_dict()[k]).abs().max().item() > \
        10 * (b.net_g.state_dict()[k] - a.net_g.state_dict()[k]).abs().max().item() + 1e-9


def test_unsupported_training_options_raise():
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    o = _opt(6, 8)
    o["train"]["optim_g"]["type"] = "SGD"
    with pytest.raises(NotImplementedError):
        TwoImageEventRecurrentRestorationModel(o)
    o = _opt(6, 8)
    o["train"]["pixel_opt"] = None
    with pytest.raises(ValueError):
        TwoImageEventRecurrentRestorationModel(o)


def test_six_steps_track_the_oracle_loss_curve():
    """Longer horizon than the two-step parity above: six fused HIP steps and six oracle steps on the same tiny problem
    keep the same loss curve (tools/train_drift.py prints 30 steps: |diff| <= 3e-8, parameters within 0.5 % of their
    displacement)."""
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    opt = {"name": "t", "is_train": True, "num_gpu": 1,
           "network_g": dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3,
                             base_num_channels=8, num_block=1, num_residual_blocks=2),
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
    for it in range(1, 7):
        model.update_learning_rate(it)
        model.feed_data({"lq": x, "voxel": ev, "gt": gt})
        model.optimize_parameters(it)
        loss, _, _, _ = O.train_step(P, st, x, ev, gt, lr=O.cosine_lr(2e-4, it - 1, 200, 1e-7), weight_decay=1e-4)
        assert abs(model.get_current_log()["l_pix"] - float(loss)) < 2e-6, it
    sd = model.net_g.state_dict()
    for k in sd:
        disp = (P[k].double() - P0[k].double()).abs().max().item()
        assert (sd[k].double().cpu() - P[k].double()).abs().max().item() <= 0.03 * disp + 1e-9, k


def test_graph_auto_falls_back_to_eager_when_capture_fails(monkeypatch):
    """`train.graph_replay: auto` / `bench.py --graph auto` must never cost a run: a capture that raises (before any parameter
    update of that step) sends this and every later step down the eager path -- same bits as a run that never tried; an
    explicit set_graph_mode(True) still raises."""
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    P = O.make_params(26, base_num_channels=8, mode="hash", seed=5)
    batches = [O.make_inputs(1, 3, 32, 32, 26, seed=30 + i, mode="hash") for i in range(3)]

    def run(mode, broken):
        m = TwoImageEventRecurrentRestorationModel(_opt(26, 8, T_max=6))
        m.net_g.load_state_dict(P)
        if broken:
            monkeypatch.setattr(m, "_graph_capture", lambda: (_ for _ in ()).throw(RuntimeError("capture refused")))
        m.set_graph_mode(mode)
        for it, (x, ev, gt) in enumerate(batches, start=1):
            m.update_learning_rate(it)
            m.feed_data({"lq": x, "voxel": ev, "gt": gt})
            m.optimize_parameters(it)
        return m, {k: v.cpu() for k, v in m.net_g.state_dict().items()}

    me, sde = run(False, False)
    with pytest.warns(UserWarning, match="capture failed"):
        ma, sda = run("auto", True)
    assert ma.graph_on is False and "capture refused" in ma.graph_fallback and ma.step_count == 3
    for k in sde:
        assert torch.equal(sde[k], sda[k]), k
    with pytest.raises(RuntimeError, match="capture refused"):
        run(True, True)
    mg, sdg = run("auto", False)                             # ... and a working capture is used (B H W = 1024 <= 2 x 256^2)
    assert mg.graph_on is True and mg._graph is not None
    for k in sde:
        assert torch.equal(sde[k], sdg[k]), k


def test_graph_replayed_steps_equal_eager_steps():
    """the train step captured into a hipGraph (zero_grad .. AdamW, weight-gradient side stream forked and
    joined inside the capture, lr / bias corrections read from device memory) and replayed == the same steps launched
    eagerly: the same loss curve and parameter trajectory BIT FOR BIT (no atomics on the path; round 6: the replay's bias
    corrections are formed exactly as refid_clip_adamw forms them), over changing inputs and a changing learning rate."""
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    P = O.make_params(26, base_num_channels=8, mode="hash", seed=5)
    batches = [O.make_inputs(2, 3, 32, 32, 26, seed=30 + i, mode="hash") for i in range(4)]

    def run(graph):
        m = TwoImageEventRecurrentRestorationModel(_opt(26, 8, T_max=6))
        m.net_g.load_state_dict(P)
        m.set_graph_mode(graph)
        losses, norms = [], []
        for it, (x, ev, gt) in enumerate(batches, start=1):
            m.update_learning_rate(it)
            m.feed_data({"lq": x, "voxel": ev, "gt": gt})
            m.optimize_parameters(it)
            losses.append(m.get_current_log()["l_pix"])
            norms.append(m.grad_norm())
        assert m.step_count == len(batches)
        return losses, norms, {k: v.double().cpu() for k, v in m.net_g.state_dict().items()}, m

    le, ne, sde, _ = run(False)
    lg, ng, sdg, mg = run(True)
    assert mg._graph is not None and len(mg._graph["graphs"]) == 1
    assert lg == le and ng == ne, (lg, le, ng, ne)
    for k in sde:
        assert torch.equal(sdg[k], sde[k]), k
    # a different input shape re-captures; switching the mode off frees the graphs and continues eagerly
    x, ev, gt = O.make_inputs(1, 2, 16, 16, 26, seed=40, mode="hash")
    mg.update_learning_rate(5)
    mg.feed_data({"lq": x, "voxel": ev, "gt": gt})
    mg.optimize_parameters(5)
    assert mg._graph["key"][0] == (1, 26, 16, 16) and np.isfinite(mg.get_current_log()["l_pix"])
    mg.set_graph_mode(False)
    assert mg._graph is None
    mg.optimize_parameters(6)
    assert mg.step_count == 6


def test_at_most_one_step_in_flight_and_prefetcher_hand_over():
    """The host enqueues a step ~5x faster than the GPU runs it; unbounded, it ran many steps ahead in round 4 and the caching
    allocator reached the whole 288 GB (a 20-step bench.py run: 3 s per step with the prefetched batch as the last straw).
    optimize_parameters therefore waits for the PREVIOUS step's end event before enqueuing (the reference synchronises every
    iteration through reduce_loss_dict's .item(), base_model.py:325-350).  Also: batches handed over by CUDAPrefetcher
    (data/prefetch_dataloader.py:84-125 mirror) give the same trajectory as resident ones."""
    from refid_amd.data import CUDAPrefetcher
    from refid_amd.train import TwoImageEventRecurrentRestorationModel
    P = O.make_params(6, base_num_channels=8, mode="hash", seed=9)
    batches = []
    for s in (4, 5):
        x, ev, gt = O.make_inputs(1, 2, 16, 16, 6, seed=s, mode="hash")
        batches.append({"lq": x.pin_memory(), "voxel": ev.pin_memory(), "gt": gt.pin_memory(), "seq": "name"})

    class Loader:
        def __iter__(self):
            return iter([dict(b) for b in batches] * 2)

    def run(prefetch):
        m = TwoImageEventRecurrentRestorationModel(_opt(6, 8))
        m.net_g.load_state_dict(P)
        pre = CUDAPrefetcher(Loader(), {"num_gpu": 1}, time_waits=True) if prefetch else None
        ends = []
        for it in range(1, 5):
            data = pre.next() if prefetch else {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batches[(it - 1) % 2].items()}
            assert data is not None and data["lq"].is_cuda and data["seq"] == "name"
            m.feed_data(data)
            m.update_learning_rate(it)
            prev = getattr(m, "_step_done", None)
            m.optimize_parameters(it)
            if prev is not None:
                assert prev.query()                      # the step before this one had finished when this one was enqueued
            ends.append(m._step_done)
        if prefetch:
            assert pre.next() is None                    # end of the epoch, like the reference's prefetcher
            pre.reset()
            assert pre.next() is not None
            assert pre.exposed_ms() >= 0.0
        return {k: v.clone() for k, v in m.net_g.state_dict().items()}

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
