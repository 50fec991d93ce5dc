"""CPU-only tests of the host side: discovery, option parsing, parameter inventory, arena
layout, schedules, bucket planning, and that the product path refuses to run without a GPU."""
import os
import textwrap

import numpy as np
import pytest
import torch

from oracle import refid_oracle as O


def test_define_network_contract():
    from refid_amd.archs import define_network
    opt = dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3, base_num_channels=32,
               num_block=1, num_residual_blocks=2)
    net = define_network(opt)
    assert "type" not in opt                                  # popped destructively like the reference
    assert type(net).__name__ == "FinalBidirectionAttenfusion"
    with pytest.raises(ValueError, match="NoSuchNet is not found."):
        define_network(dict(type="NoSuchNet"))


def test_arch_registry_surface():
    """BASELINE north_star's `@ARCH_REGISTRY.register()` surface (later BasicSR; the reference's vintage scans *_arch.py
    files instead, archs/__init__.py:9-46 -- define_network serves both)."""
    from refid_amd import archs
    from refid_amd.registry import ARCH_REGISTRY, Registry
    assert set(ARCH_REGISTRY.keys()) >= {"FinalBidirectionAttenfusion", "SingleMultiConnectEVHINet"}
    cls = ARCH_REGISTRY.get("FinalBidirectionAttenfusion")
    assert cls is archs.final_bidirection_attenfusion_arch.FinalBidirectionAttenfusion
    with pytest.raises(KeyError):
        ARCH_REGISTRY.get("NoSuchNet")
    reg = Registry("t")

    @reg.register()
    class A:                                              # noqa: N801
        pass

    @reg.register
    class B:                                              # noqa: N801
        pass
    assert reg.get("A") is A and reg.get("B") is B and "A" in reg
    with pytest.raises(AssertionError, match="already registered"):
        reg.register(A)
    # a class that is only discoverable by the reference's module scan still resolves (registry first, scan second)
    class OnlyScanned:                                    # noqa: N801
        def __init__(self, k):
            self.k = k
    archs._arch_modules[0].OnlyScanned = OnlyScanned
    try:
        assert archs.define_network(dict(type="OnlyScanned", k=3)).k == 3
    finally:
        del archs._arch_modules[0].OnlyScanned


@pytest.mark.parametrize("img_chn,count", [(26, 15928355), (6, 15912355), (3, 15909955)])
def test_state_dict_keys_shapes_and_counts(img_chn, count):
    from refid_amd.archs import define_network
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=img_chn, ev_chn=2, num_encoders=3,
                              base_num_channels=32, num_block=1, num_residual_blocks=2))
    sd = net.state_dict()
    ref = O.param_shapes(img_chn)                   # pinned to the reference by tests/test_oracle_golden.py
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 183
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sum(p.numel() for p in net.parameters()) == count
    # a reference-style checkpoint loads strictly, also with DDP's "module." prefix stripped by the wrapper
    P = O.make_params(img_chn)
    net.load_state_dict(P, strict=True)
    assert torch.equal(net.state_dict()["pred.conv2d.weight"], P["pred.conv2d.weight"])
    # zero-initialised beta/gamma, LayerNorm 1/0 after reset (fm:287-288)
    net.reset_parameters()
    sd = net.state_dict()
    assert float(sd["encoders_forward.1.atten_fuse.beta"].abs().max()) == 0.0
    assert float(sd["encoders_forward.1.atten_fuse.norm1.weight"].min()) == 1.0


def test_unsupported_options_fail_loudly():
    from refid_amd.archs import define_network
    base = dict(type="FinalBidirectionAttenfusion", img_chn=6, ev_chn=2, num_encoders=3, num_block=1)
    for bad in (dict(num_encoders=5), dict(num_encoders=1), dict(num_block=0), dict(skip_type="concat"), dict(norm="BN"),
                dict(use_recurrent_upsample_conv=False)):
        with pytest.raises(NotImplementedError):
            define_network({**base, **bad})
    # num_block (ResidualBlockNoBN per encoder trunk; the YAMLs use 1, the reference ctor's default is 3): any >= 1 builds, with
    # the reference's state-dict keys -- the decoders' trunks keep their single block (rsm:375-384 does not pass num_block on)
    net3 = define_network({**base, "num_block": 3, "base_num_channels": 8})
    keys = list(net3.state_dict().keys())
    assert len(keys) == 183 + 2 * 6 * 4 and "encoders_forward.2.recurrent_block.forward_trunk.main.2.2.conv2.bias" in keys
    assert "decoders.0.forward_trunk.main.2.1.conv1.weight" not in keys
    assert keys == list(O.param_shapes(6, base_num_channels=8, num_block=3).keys())
    # the reference ctor's own defaults (arch:90-92: num_encoders=4, num_block=3) build, with the reference's keys (pinned to the
    # reference by tests/golden/tiny26_default_ctor_train.npz: oracle/make_golden.py asserts key order and shapes against it)
    dflt = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=6, ev_chn=2, base_num_channels=8))
    assert dflt.num_encoders == 4 and dflt.num_block == 3
    assert list(dflt.state_dict().keys()) == list(O.param_shapes(6, base_num_channels=8, num_block=3, num_encoders=4).keys())
    with pytest.raises(AssertionError):
        define_network({**base, "img_chn": 0})
    # ignored-by-the-reference keywords are accepted
    define_network({**base, "recurrent_block_type": "convgru", "activation": "tanh", "use_first_dcn": True})


def test_no_cpu_fallback():
    from refid_amd._lib import RefidHipError
    from refid_amd.archs import define_network
    from refid_amd import ops
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=6, ev_chn=2, num_encoders=3,
                              base_num_channels=8, num_block=1))
    x, ev, _ = O.make_inputs(1, 2, 16, 16, 6)
    with pytest.raises(RefidHipError, match="no CPU path"):
        net(x=x, event=ev)
    with pytest.raises(RefidHipError):
        ops.conv2d(torch.zeros(1, 8, 8, 64), torch.zeros(10), torch.zeros(1, 8, 8, 64), kh=3, kw=3, pad=1,
                   cout=64, cout_pad=64)


def test_param_arena_layout():
    from refid_amd.engine import ParamArena, param_shapes
    sh = param_shapes(26)
    A = ParamArena(sh, torch.device("cpu"))
    prev_end = 0
    for k, (off, n) in A.offsets.items():
        assert off % 4 == 0 and off >= prev_end
        prev_end = off + n
    assert A.total % 4 == 0 and A.total >= 15928355
    A.p("pred.conv2d.bias").fill_(3.0)
    off, n = A.offsets["pred.conv2d.bias"]
    assert float(A.flat_p[off:off + n].sum()) == 9.0 and A.g("pred.conv2d.bias").shape == (3,)


def test_bucket_plan_covers_arena_once():
    from refid_amd.dist import bucket_slices, EARLY_PREFIXES, shard_batch
    from refid_amd.engine import ParamArena, param_shapes
    A = ParamArena(param_shapes(26), torch.device("cpu"))
    runs = bucket_slices(A.offsets, A.total)
    spans = sorted(runs["early"] + runs["late"])
    assert spans[0][0] == 0 and spans[-1][1] == A.total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for k, (off, n) in A.offsets.items():
        phase = "early" if k.startswith(EARLY_PREFIXES) else "late"
        assert any(s <= off and off + n <= e for s, e in runs[phase]), k
    assert len(runs["early"]) == 2 and len(runs["late"]) == 2
    assert sorted(shard_batch(8, 0, 4) + shard_batch(8, 1, 4) + shard_batch(8, 2, 4) + shard_batch(8, 3, 4)) == list(range(8))


def test_options_parse_and_shapes(tmp_path):
    from refid_amd import options
    yml = tmp_path / "train_debug.yml"
    yml.write_text(textwrap.dedent("""
        name: Final_debug_run
        model_type: TwoImageEventRecurrentRestorationModel
        scale: 1
        num_gpu: 8
        manual_seed: 10
        datasets:
          train:
            name: gopro-twoblur-train
            type: GoProEventRecurrentDataset
            dataroot: ~/data
            num_end_interpolation: 11
            num_inter_interpolation: 1
            return_deblur_voxel: true
            gt_size: 256
            batch_size_per_gpu: 1
        network_g:
          type: FinalBidirectionAttenfusion
          img_chn: 26
          ev_chn: 2
          num_encoders: 3
          base_num_channels: 32
          num_block: 1
          num_residual_blocks: 2
        path:
          pretrain_network_g: ~
          strict_load_g: true
          resume_state: ~
        train:
          optim_g: {type: AdamW, lr: !!float 2e-4, weight_decay: !!float 1e-4, betas: [0.9, 0.99]}
          scheduler: {type: TrueCosineAnnealingLR, T_max: 200000, eta_min: !!float 1e-7}
          total_iter: 200000
          pixel_opt: {type: CharbonnierLoss, loss_weight: 1, reduction: mean}
        val: {val_freq: !!float 5e4, max_minibatch: 2, grids: ~}
        logger: {print_freq: 200, save_checkpoint_freq: !!float 25000}
        dist_params: {backend: nccl, port: 29500}
    """))
    opt = options.parse(str(yml), is_train=True)
    assert opt["is_train"] and opt["datasets"]["train"]["phase"] == "train" and opt["datasets"]["train"]["scale"] == 1
    assert opt["val"]["val_freq"] == 8 and opt["logger"]["print_freq"] == 1          # debug-mode overrides
    assert opt["path"]["models"].endswith(os.path.join("experiments", "Final_debug_run", "models"))
    assert opt["train"]["optim_g"]["lr"] == 2e-4 and opt["dist_params"]["backend"] == "nccl"
    assert options.shapes_from_dataset_opt(opt["datasets"]["train"]) == (23, 26)
    assert options.shapes_from_dataset_opt(dict(num_end_interpolation=11, num_inter_interpolation=3,
                                                return_deblur_voxel=True)) == (25, 26)
    assert options.shapes_from_dataset_opt(dict(num_end_interpolation=1, num_inter_interpolation=7)) == (7, 6)
    t = options.parse(str(yml), is_train=False)
    assert "results_root" in t["path"]


def test_cosine_schedule_matches_reference_stepping():
    """base_model.py:158-180: scheduler.step() from iteration 2 on; torch CosineAnnealingLR."""
    import math
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100, eta_min=1e-7)
    epoch = 0
    for it in range(1, 30):
        if it > 1:
            opt.step(); sch.step(); epoch += 1
        mine = 1e-7 + (2e-4 - 1e-7) * (1 + math.cos(math.pi * epoch / 100)) / 2
        assert abs(opt.param_groups[0]["lr"] - mine) < 1e-12


def test_evhinet_module_mirrors_reference_state_dict_on_cpu():
    """keys / shapes / order of the HIP module == the reference's (via the pinned oracle inventory);
    on CPU the module exists but refuses to run (no fallback)."""
    from oracle import evhinet_oracle as E
    from refid_amd.archs import define_network
    from refid_amd._lib import RefidHipError
    net = define_network(dict(type="SingleMultiConnectEVHINet"))
    want = E.param_shapes()
    sd = net.state_dict()
    assert list(sd.keys()) == list(want.keys()) and len(sd) == 156
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    with pytest.raises(RefidHipError):
        net(x=torch.zeros(1, 3, 16, 16), event=torch.zeros(1, 6, 16, 16))


def test_init_matches_reference_moments(golden_dir):
    """A11: reset_parameters() against the reference ctor's own initialisation
    (recurrent_sub_modules.py:752-753,776-804: ResidualBlockNoBN convs Kaiming-normal x0.1 with zero bias; torch
    defaults elsewhere; LayerNorm2d 1/0; beta/gamma 0) -- per-parameter moments stored by oracle/make_golden.py."""
    import math
    from refid_amd.archs import define_network
    z = np.load(os.path.join(golden_dir, "host_logic.npz"))
    keys, mom = [str(k) for k in z["init_keys"]], z["init_moments"]
    torch.manual_seed(123)                                   # a DIFFERENT seed: only the distributions must agree
    net = define_network(dict(type="FinalBidirectionAttenfusion", img_chn=26, ev_chn=2, num_encoders=3,
                              base_num_channels=32, num_block=1, num_residual_blocks=2))
    sd = net.state_dict()
    assert list(sd.keys()) == keys
    n_kaiming = 0
    for k, (numel, mean, std, amax, lo, hi) in zip(keys, mom):
        v = sd[k].double().flatten()
        assert v.numel() == numel, k
        if std == 0.0:                                       # constants: LN weight 1 / bias 0, beta, gamma, resblock biases
            assert float(v.min()) == lo and float(v.max()) == hi, k
            continue
        my_std = float(v.std(unbiased=False))
        if ".main.2.0.conv" in k:                            # Kaiming normal (fan_in, gain sqrt(2)) x 0.1
            n_kaiming += 1
            assert k.endswith("weight")
            fan_in = sd[k].shape[1] * 9
            want = 0.1 * math.sqrt(2.0 / fan_in)
            assert abs(std - want) < 0.02 * want, (k, std, want)          # the reference itself
            assert abs(my_std - want) < 0.02 * want, (k, my_std, want)
            assert float(v.abs().max()) > 3.0 * want                      # normal tails, not a uniform
        else:                                                # torch default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            w = sd[k[:-4] + "weight"] if k.endswith("bias") else sd[k]
            bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
            assert amax <= bound * (1 + 1e-6) and float(v.abs().max()) <= bound * (1 + 1e-6), k
            tol = 0.02 if numel >= 10000 else (0.1 if numel >= 512 else 0.45)
            assert abs(my_std - bound / math.sqrt(3)) < tol * bound, (k, my_std, bound)
            assert abs(std - bound / math.sqrt(3)) < tol * bound, (k, std, bound)
            if numel >= 512:
                assert float(v.abs().max()) > 0.9 * bound, k
    assert n_kaiming == 2 * (6 + 3)                          # conv1/conv2 of 6 EvR trunks + 3 decoder trunks


def test_schedulers_match_reference_sequences(golden_dir):
    """base_model.py:77-108,158-180: every scheduler type the reference can build, stepped from iteration 2,
    with the linear warm-up that scales the INITIAL lr; sequences recorded from the reference's own scheduler classes."""
    import ast
    from refid_amd import train
    z = np.load(os.path.join(golden_dir, "host_logic.npz"))
    names = [k[3:] for k in z.files if k.startswith("lr/")]
    assert len(names) == 8 and "VibrateLR" in names
    for name in names:
        kind, cfg, warm, total = ast.literal_eval(str(z["lrcfg/" + name]))
        m = train.TwoImageEventRecurrentRestorationModel.__new__(train.TwoImageEventRecurrentRestorationModel)
        m.base_lr = 2e-4
        m.cur_lr = train.scheduler_initial_lr(kind, cfg, 2e-4, total)      # (what init_training_settings sets)
        m.sched_type, m.sched_cfg, m.sched_epoch, m.total_iter = kind, cfg, 0, total
        seq = []
        for it in range(1, 40):
            m.update_learning_rate(it, warmup_iter=warm)
            seq.append(m.get_current_learning_rate()[0])
        np.testing.assert_allclose(np.array(seq), z["lr/" + name], rtol=1e-9, atol=1e-15, err_msg=name)
    with pytest.raises(NotImplementedError):
        train.scheduler_lr("CosineAnnealingLR", {}, 1, 2e-4, 2e-4)     # the reference rejects this spelling too


def test_forward_wavefront_streams_only_for_small_batches(monkeypatch):
    """engine.use_pipeline: the three forward-wavefront streams exist only while the batch leaves the chip room (the device has
    four hardware queues: DESIGN.md section 5); REFID_PIPELINE=0 / 1 force it."""
    from refid_amd import engine
    monkeypatch.setattr(engine, "PIPELINE", "auto")
    assert engine.use_pipeline(1, 256, 256) and engine.use_pipeline(4, 256, 256) and engine.use_pipeline(1, 512, 512)
    assert not engine.use_pipeline(8, 256, 256) and not engine.use_pipeline(2, 512, 512)
    monkeypatch.setattr(engine, "PIPELINE", False)
    assert not engine.use_pipeline(1, 64, 64)
    monkeypatch.setattr(engine, "PIPELINE", True)
    assert engine.use_pipeline(8, 256, 256)


def test_split_tile_grid_policy_follows_the_split_k_policy():
    """engine._fills_gpu: conv_down goes to the split tile only when its smallest grid gives each of the 256 CUs a
    workgroup -- by the real batch under the 'auto' split-K policy, as if 8 samples otherwise (batch invariance)."""
    from refid_amd import engine, ops
    old = ops.WINO_SPLIT
    try:
        ops.WINO_SPLIT = 2
        assert engine._fills_gpu(8, 128, 128, 64, 1) and engine._fills_gpu(2, 128, 128, 64, 1)
        assert not engine._fills_gpu(1, 128, 128, 64, 1)            # 128 workgroups: the fp32 tile's split-K form instead
        assert engine._fills_gpu(1, 128, 128, 64, 4)                # input gradient: four parity classes
        ops.WINO_SPLIT = 1
        assert engine._fills_gpu(1, 128, 128, 64, 1) == engine._fills_gpu(8, 128, 128, 64, 1)
        assert engine._fills_gpu(1, 16, 32, 64, 1) == engine._fills_gpu(5, 16, 32, 64, 1)
    finally:
        ops.WINO_SPLIT = old


REF_OPTIONS = "/root/reference/options"


@pytest.mark.skipif(not os.path.isdir(REF_OPTIONS), reason="needs the reference checkout (build container only)")
def test_every_shipped_yaml_of_the_reference_parses_and_builds(monkeypatch):
    """The config contract with the REAL files: all options/**/*.yml of the reference go through
    refid_amd.options.parse (same dict as the reference's own parser, utils/options.py:31-95, up to the root path), the
    dataset blocks give the (T, img_chn) the network block expects, define_network builds the 183-key
    module, and the train / test model wrapper accepts the train / val blocks (engine stubbed: no GPU here)."""
    import glob
    import importlib.util
    import types
    from refid_amd import options, train
    from refid_amd.archs import define_network
    from refid_amd.engine import ParamArena, param_shapes
    spec = importlib.util.spec_from_file_location("ref_options", "/root/reference/basicsr/utils/options.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    files = sorted(glob.glob(os.path.join(REF_OPTIONS, "**", "*.yml"), recursive=True))
    assert len(files) == 16

    def stubbed(net_opt):
        net = define_network(net_opt)                      # CPU module: parameters exist, the HIP engine does not
        arena = ParamArena(param_shapes(net.img_chn, net.ev_chn, net.out_chn, net.base_num_channels,
                                        net.num_residual_blocks), torch.device("cpu"))
        net._engine = types.SimpleNamespace(arena=arena, mark_params_changed=lambda: None)
        net.to = lambda *a, **k: net                       # (.to() would re-bind the engine to the device and drop the stub)
        return net

    monkeypatch.setattr(train, "define_network", stubbed)
    expect_T = {"1skip": 23, "1attenfusion.yml": 23, "3skip": 25, "7skip": 7, "15skip": 15}
    for f in files:
        is_train = os.sep + "train" + os.sep in f
        opt = options.parse(f, is_train=is_train)
        want = ref.parse(f, is_train=is_train)
        got_path, want_path = opt.pop("path"), want.pop("path")
        assert opt == want, f
        assert sorted(got_path) == sorted(want_path), f
        opt["path"] = got_path
        ng = opt["network_g"]
        assert ng["type"] == "FinalBidirectionAttenfusion" and ng["ev_chn"] == 2 and ng["num_encoders"] == 3
        T = next(v for k, v in expect_T.items() if k in os.path.basename(f))
        for name, ds in opt["datasets"].items():
            assert options.shapes_from_dataset_opt(ds) == (T, ng["img_chn"]), (f, name)
        net = define_network(dict(ng))
        sd = net.state_dict()
        assert len(sd) == 183 and sd["head_img.conv2d.weight"].shape == (32, ng["img_chn"], 5, 5), f
        opt["num_gpu"] = 0                                 # CPU tensors for the optimizer arenas of the stub
        opt["path"]["pretrain_network_g"] = None           # (the shipped files point at checkpoints that are not here)
        model = train.TwoImageEventRecurrentRestorationModel(opt)
        if is_train:
            tr = opt["train"]
            assert model.sched_type == tr["scheduler"]["type"] and model.base_lr == float(tr["optim_g"]["lr"])
            assert model.betas == tuple(tr["optim_g"]["betas"]) and model.weight_decay == float(tr["optim_g"]["weight_decay"])
            assert model.exp_avg.numel() == model.net_g.engine.arena.total
            for it in (1, 2, 3):
                model.update_learning_rate(it, warmup_iter=tr.get("warmup_iter", -1))
            lr = model.get_current_learning_rate()[0]
            assert 0 < lr <= model.base_lr
        else:
            assert not hasattr(model, "exp_avg") and opt["val"].get("max_minibatch", 1) >= 1


def test_batched_finish_always_flushes_and_fallback_packs_stay_out_of_the_plan(monkeypatch):
    """Host logic of two round-5 changes, no GPU: (1) engine.finish_wgrads queues every op's element-wise reduction stage
    (phase 4) and ALWAYS ends with refid_wgrad_finish_flush -- also when an op fails, so nothing stays queued in the library;
    with the side-stream switch on it falls back to per-op launches.  (2) The fp32 Winograd packings of convs that run on their
    Winograd x six planes are not in the per-step pack plan (packed on demand), everything else is."""
    from refid_amd import engine, ops
    calls = []
    monkeypatch.setattr(ops, "wgrad_finish_flush", lambda: calls.append("flush"))

    class Op:
        def __init__(self, fail=False):
            self.fail = fail

        def finish_wgrad(self, batched=False):
            calls.append(("finish", batched))
            if self.fail:
                raise RuntimeError("boom")

    monkeypatch.setattr(engine, "FINISH_BATCH", True)
    monkeypatch.setattr(engine, "OVERLAP_WGRAD", 0)
    engine.finish_wgrads([Op(), Op()])
    assert calls == [("finish", True), ("finish", True), "flush"]
    calls.clear()
    with pytest.raises(RuntimeError):
        engine.finish_wgrads([Op(), Op(fail=True), Op()])
    assert calls == [("finish", True), ("finish", True), "flush"]
    calls.clear()
    monkeypatch.setattr(engine, "FINISH_BATCH", False)
    engine.finish_wgrads([Op()])
    assert calls == [("finish", False)]

    planned = []
    monkeypatch.setattr(ops.PackPlan, "build", lambda self: planned.extend(self.items) or self)
    monkeypatch.setattr(ops.PackPlan, "run", lambda self: None)
    monkeypatch.setattr(ops, "add", lambda a, b, out=None: out)
    eng = engine.Engine(26, device="cpu")
    eng.repack()
    lazy = [o for o in eng.all_ops if o.wp_lazy or o.wd_lazy]
    assert len(lazy) >= 40 and eng.arena.pack_epoch == 1
    dsts = {it[3].data_ptr() for it in planned}
    for o in eng.all_ops:
        assert (o.wp.data_ptr() in dsts) == (not o.wp_lazy), o.name
        if o.wd is not None:
            assert (o.wd.data_ptr() in dsts) == (not o.wd_lazy), o.name
        for nm in ("wp6", "wd6", "wps", "wds"):
            t = getattr(o, nm)
            assert t is None or t.data_ptr() in dsts, (o.name, nm)
    # ConvTranspose2d ops: pointwise-tile GEMMs both ways, weight gradient with the swapped-role 2x2 stride-2 geometry
    t2 = eng.dec[0]["t2"]
    assert t2.kind == "convT" and t2.f_algo == 3 and t2.d_algo == 3 and t2.d_role == ops.ROLE_CONVT_DGRAD_PW
    assert t2._wg_geo() == dict(kh=2, kw=2, stride=2, pad=0, i_total=t2.co)
    assert eng.pred.wp6 is not None and eng.pred.f_algo == 0 and not eng.pred.wp_lazy      # thin output on the Winograd x six form


def test_weight_gradient_group_is_capped_by_free_memory():
    """engine.wgrad_group_cap: a waiting weight-gradient call keeps its step's tensors alive (~4.9 KB per pixel and
    step); the group shrinks when the operands of `n` steps would not fit into half of the free HBM instead of running out of
    memory at a larger crop / T / batch.  Config 2 (B=8, 256 x 256) on a 288 GB part keeps all 23 steps."""
    from refid_amd import engine
    pix = 8 * 256 * 256
    per = pix * engine.WGRAD_KEEP_BYTES_PER_PIXEL_STEP
    assert engine.wgrad_group_cap(23, pix, "cpu", free_bytes=160 << 30) == 23          # what is free when config-2 BPTT starts
    assert engine.wgrad_group_cap(23, pix, "cpu", free_bytes=2 * per * 8) == 8
    assert engine.wgrad_group_cap(23, pix, "cpu", free_bytes=per) == 1                 # never below one step per launch
    assert engine.wgrad_group_cap(23, 4 * pix, "cpu", free_bytes=160 << 30) == 8       # four times the pixels: a third of the steps
    assert engine.wgrad_group_cap(1, pix, "cpu", free_bytes=0) == 1 and engine.wgrad_group_cap(23, 0, "cpu") == 23
    assert engine.wgrad_group_cap(23, pix, "cpu") == 23                                # (no CUDA device: unchanged)
    assert engine.WGRAD_GROUP == (8 if engine.OVERLAP_WGRAD else 24)                   # derived from overlap_wgrad(), not the raw env string


def test_streaming_weight_gradient_eligibility_mirrors_the_library():
    """ConvOp._pws_plan_ok (Python) decides which weight gradients are handed to the streaming kernel; the library decides the
    same thing again (refid_wgrad_pws_ok) and would reject a mismatch at launch.  No GPU needed: refid_wgrad_workspace_bytes is
    host arithmetic and returns 0 for an algo-8 request (ConvTranspose2d's weight gradient as a streaming patch GEMM) exactly when
    the library would refuse it -- swept over channel counts and row widths."""
    import ctypes as C
    from refid_amd import _lib
    from refid_amd.engine import ConvOp
    L = _lib.lib()
    for ci in (32, 48, 64, 96, 128, 256):                 # layer input channels = the GEMM's output rows
        for co in (8, 16, 24, 32, 64, 128):               # layer output channels: K = 4 co
            for w_lo in (8, 16, 24, 32, 48, 64):
                d = _lib.WgradDesc()
                d.g, d.in_a, d.dw, d.slabs = 4096, 8192, 12288, 16384          # (never dereferenced)
                d.ld_g, d.c_o = ci, ci
                d.ld_a, d.c_a = co, co
                d.n, d.ho, d.wo, d.h, d.w = 2, 8, w_lo, 16, 2 * w_lo
                d.kh, d.kw, d.stride, d.pad = 2, 2, 2, 0
                d.i_base, d.i_total, d.o_real, d.algo, d.phase = 0, co, ci, 8, 1
                lib_ok = L.refid_wgrad_workspace_bytes(C.byref(d)) > 0
                py_ok = ConvOp._pws_plan_ok(ci, 2 * co, 2 * co, 4 * co)
                # (the row width must fill whole ring buffers of 16 or 32 pixels: the library's workspace query says so too --
                #  it used to report a size for a geometry its launch then refused --, the engine asks for a multiple of 32)
                if w_lo % 32 == 0:
                    assert lib_ok == py_ok, (ci, co, w_lo, lib_ok, py_ok)
                elif w_lo % 16:
                    assert not lib_ok, (ci, co, w_lo)
                else:
                    assert not lib_ok or py_ok, (ci, co, w_lo)


def test_cached_conv_descriptors_equal_the_validating_path_byte_for_byte(monkeypatch):
    """ops.DESC_CACHE (REFID_DESC_CACHE=1): after the first call with a signature, conv2d copies a recorded descriptor and
    refreshes only the pointers.  Against a recording library handle on CPU tensors: for a mix of geometries, epilogues, two
    sources, second outputs, row ranges and fresh tensors per call, the bytes handed to refid_conv2d by the cached path equal the
    bytes the full path builds -- so whatever the library does with one it does with the other."""
    import ctypes as C
    import torch
    from refid_amd import ops, _lib
    real = _lib.lib()
    sent = []

    class Lib:
        def __getattr__(self, name):
            if name == "refid_conv2d":
                return lambda dref, st: sent.append(C.string_at(C.addressof(dref._obj), C.sizeof(_lib.ConvDesc))) or 0
            return getattr(real, name)

    handle = Lib()
    monkeypatch.setattr(ops, "lib", lambda: handle)
    monkeypatch.setattr(ops, "_stream", lambda: C.c_void_p(0))
    monkeypatch.setattr(ops, "_REQUIRE_CUDA", False)
    monkeypatch.setattr(ops, "_DESC_CACHE", {})
    wsbuf = {}

    def workspace(nbytes, device, kind="wgrad"):           # (the real one asks torch.cuda for the current stream)
        if kind not in wsbuf or wsbuf[kind].numel() * 4 < nbytes:
            wsbuf[kind] = torch.zeros((nbytes + 3) // 4)
        return wsbuf[kind]

    monkeypatch.setattr(ops, "_workspace", workspace)
    gen = torch.Generator().manual_seed(0)

    def t(*shape):
        return torch.zeros(*shape)

    def calls():
        w = t(1 << 16)
        for n, h, wd in ((1, 8, 32), (2, 16, 16)):
            a, b, o = t(n, h, wd, 64), t(n, h, wd, 64), t(n, h, wd, 64)
            r, m, p2, o2 = t(n, h, wd, 64), t(n, h, wd, 64), t(n, h, wd, 64), t(n, h, wd, 64)
            bias = t(64)
            yield dict(args=(a, w, o), kw=dict(kh=3, kw=3, pad=1, cout=64, cout_pad=64, algo=5, bias=bias, slope_pre=0.1))
            yield dict(args=(a, w, o), kw=dict(kh=3, kw=3, pad=1, cout=64, cout_pad=64, algo=5, in_b=b, res=r, mask=m, slope_mask=0.1,
                                                add2=p2, out2=o2))
            yield dict(args=(a, w, o[..., :32]), kw=dict(kh=3, kw=3, pad=1, cout=32, cout_pad=64, co_base=32, algo=1, res=r[..., 32:]))
            yield dict(args=(a, w, o), kw=dict(kh=1, kw=1, pad=0, cout=64, cout_pad=64, algo=3, in_b=b, bias=bias, mask=m, mask_mode=1))
            yield dict(args=(a, w, t(n, h // 2, wd // 2, 64)), kw=dict(kh=4, kw=4, stride=2, pad=1, cout=64, cout_pad=64, algo=4, terms=6))
            yield dict(args=(a, w, t(n, 2 * h, 2 * wd, 32)), kw=dict(kh=1, kw=1, pad=0, mode=1, cout=128, cout_pad=128, algo=3, bias=t(32)))

    def run(cached):
        monkeypatch.setattr(ops, "DESC_CACHE", cached)
        sent.clear()
        torch.manual_seed(0)
        out = []
        for rep in range(3):                               # fresh tensors (pointers) every repetition, the same signatures
            for c in calls():
                ops.conv2d(*c["args"], **c["kw"])
                out.append((tuple(x.data_ptr() for x in c["args"]), sent[-1]))
        return out

    plain = run(False)
    cached = run(True)
    assert len(plain) == len(cached) == 36 and len(ops._DESC_CACHE) == 12
    # (ws / ws_bytes: address and CAPACITY of the grow-only workspace the wrapper lends -- both paths ask for it when needed)
    off = {n: getattr(_lib.ConvDesc, n).offset for n in ops._DESC_PTRS + ("ws", "ws_bytes")}
    for (pp, pb), (cp, cb) in zip(plain, cached):
        # the two runs allocated different tensors: compare with the pointer fields masked, and the pointer fields by role
        pbm, cbm = bytearray(pb), bytearray(cb)
        for o in off.values():
            pbm[o:o + 8] = cbm[o:o + 8] = bytes(8)
        assert pbm == cbm
        dp, dc = _lib.ConvDesc.from_buffer_copy(pb), _lib.ConvDesc.from_buffer_copy(cb)
        for nm in ops._DESC_PTRS + ("ws",):
            assert (getattr(dp, nm) is None) == (getattr(dc, nm) is None), nm
        assert (dp.ws_bytes == 0) == (dc.ws_bytes == 0) and (dp.ws is None) == (dp.ws_bytes == 0)
        need = real.refid_conv_workspace_bytes(C.byref(dp))
        assert dp.ws_bytes >= need and dc.ws_bytes >= need
        assert (dp.in_a, dp.w_packed, dp.out) == pp and (dc.in_a, dc.w_packed, dc.out) == cp
    # a misaligned operand is still refused on the cached path
    a, w, o = t(1, 8, 32, 64), t(1 << 16), t(1, 8, 32, 68)[..., 2:66]
    with pytest.raises(_lib.RefidHipError):
        ops.conv2d(t(1, 8, 32, 64), w, o, kh=3, kw=3, pad=1, cout=64, cout_pad=64, algo=5, bias=t(64), slope_pre=0.1)
