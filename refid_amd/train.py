"""Train / eval step harness mirroring the reference's model wrapper for this path.

Reference: basicsr/models/twoImage_event_recurrent_model.py -- __init__ :18-35,
init_training_settings :37-65, setup_optimizers :67-95, feed_data :97-113,
optimize_parameters :273-310, test :312-330; base_model.py :77-108 (schedulers),
:158-180 (update_learning_rate), :188-219/:256-281 (save/load network), :325-350 (loss reduce).

Only what a train step / eval step needs is mirrored; datasets,
logging, validation image dumps are out of scope.  The step itself is MI355X-native:
forward + BPTT in the HIP engine, Charbonnier forward+backward in one kernel, global grad
norm + clip + AdamW fused over the flat arenas, gradient all-reduce (RCCL) overlapped with BPTT.
"""
import math
import os
from collections import OrderedDict
from copy import deepcopy

import torch

from . import ops
from .archs import define_network
from .dist import GradSync, get_dist_info


# REFID_RUN_AHEAD_BOUND=0: let the host enqueue as many steps ahead as the HIP queue takes (A/B only; see _bound_run_ahead)
RUN_AHEAD_BOUND = os.environ.get("REFID_RUN_AHEAD_BOUND", "1") != "0"


def scheduler_lr(kind, cfg, epoch, base_lr, prev_lr, total_iter=None):
    """Learning rate after `epoch` scheduler steps -- the schedulers base_model.py:77-108 can build
    (models/lr_scheduler.py:6-177 + torch CosineAnnealingLR), restated per step for ONE param group.
    prev_lr: the current lr (the multi-step scheme is recursive on it)."""
    if kind == "none":
        return base_lr
    if kind == "TrueCosineAnnealingLR":
        # torch.optim.lr_scheduler.CosineAnnealingLR.get_lr: RECURSIVE on the group's current lr (so a warm-up that
        # overwrote the lr carries over, exactly as in the reference); equals the closed form otherwise
        eta, tm = cfg["eta_min"], cfg["T_max"]
        if (epoch - 1 - tm) % (2 * tm) == 0:
            return prev_lr + (base_lr - eta) * (1 - math.cos(math.pi / tm)) / 2
        return (1 + math.cos(math.pi * epoch / tm)) / (1 + math.cos(math.pi * (epoch - 1) / tm)) * (prev_lr - eta) + eta
    if kind in ("MultiStepLR", "MultiStepRestartLR"):     # lr_scheduler.py:6-46
        restarts, weights = list(cfg.get("restarts", (0,))), list(cfg.get("restart_weights", (1,)))
        if epoch in restarts:
            return base_lr * weights[restarts.index(epoch)]
        n = list(cfg["milestones"]).count(epoch)
        return prev_lr * cfg.get("gamma", 0.1) ** n if n else prev_lr
    if kind == "CosineAnnealingRestartLR":                # lr_scheduler.py:117-177
        periods, weights = list(cfg["periods"]), list(cfg.get("restart_weights", (1,)))
        cum = [sum(periods[:i + 1]) for i in range(len(periods))]
        idx = next(i for i, c in enumerate(cum) if epoch <= c)
        near = 0 if idx == 0 else cum[idx - 1]
        eta = cfg.get("eta_min", 0)
        return eta + weights[idx] * 0.5 * (base_lr - eta) * (1 + math.cos(math.pi * (epoch - near) / periods[idx]))
    if kind == "LinearLR":                                # lr_scheduler.py:48-67
        return (1 - epoch / total_iter) * base_lr
    if kind == "VibrateLR":                               # lr_scheduler.py:71-112: a sawtooth of period total_iter // 80 under a
        process = epoch / total_iter                      # three-stage envelope (1 -> 0 over the first 3/8, then 0.2, then 0.1)
        f = 0.1
        if process < 3 / 8:
            f = 1 - process * 8 / 3
        elif process < 5 / 8:
            f = 0.2
        period = total_iter // 80
        half = period // 2
        t = epoch % period
        f2 = t / half
        if t >= half:
            f2 = 2 - f2
        weight = f * f2
        if epoch < half:
            weight = max(0.1, weight)
        return weight * base_lr
    raise NotImplementedError(f"Scheduler {kind} is not implemented yet.")


def scheduler_initial_lr(kind, cfg, base_lr, total_iter=None):
    """The lr a freshly built scheduler leaves in the optimizer (torch's _LRScheduler.__init__ steps once: last_epoch = 0).
    Every scheduler above returns base_lr there except VibrateLR, whose sawtooth starts at its floor (0.1 base_lr)."""
    return scheduler_lr(kind, cfg, 0, base_lr, base_lr, total_iter) if kind == "VibrateLR" else base_lr


SCHEDULERS = ("TrueCosineAnnealingLR", "MultiStepLR", "MultiStepRestartLR", "CosineAnnealingRestartLR", "LinearLR", "VibrateLR")


class TwoImageEventRecurrentRestorationModel:
    PIXEL_LOSSES = ("CharbonnierLoss",)            # the only one the reference's configs for this model use
    LOWLR_RATIO = 0.1                              # twoImage_event_recurrent_model.py:81 (`ratio`)

    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt.get("num_gpu", 1) != 0 else "cpu")
        self.is_train = opt.get("is_train", True)
        self.net_g = define_network(deepcopy(opt["network_g"])).to(self.device)
        load_path = opt.get("path", {}).get("pretrain_network_g", None)
        if load_path is not None:
            self.load_network(self.net_g, load_path, opt["path"].get("strict_load_g", True),
                              opt["path"].get("param_key", "params"))
        self.rank, self.world = get_dist_info()
        self.log_dict = OrderedDict()
        if self.is_train:
            self.init_training_settings()

    # ---- S3: optimiser / scheduler set-up -------------------------------------------------------
    def init_training_settings(self):
        train_opt = self.opt["train"]
        pix = train_opt.get("pixel_opt")
        if not pix:
            raise ValueError("Both pixel and perceptual losses are None.")
        if pix.get("type") not in self.PIXEL_LOSSES or pix.get("reduction", "mean") != "mean":
            raise NotImplementedError(f"pixel loss {pix.get('type')!r}: supported here: {self.PIXEL_LOSSES} (reduction=mean)")
        self.pixel_type = pix["type"]
        self.loss_weight = float(pix.get("loss_weight", 1.0))
        self.loss_eps = float(pix.get("eps", 1e-12))
        og = dict(train_opt["optim_g"])
        optim_type = og.pop("type")
        if optim_type != "AdamW":
            raise NotImplementedError(f"optimizer {optim_type} is not supperted yet.")
        self.base_lr = float(og["lr"])
        self.weight_decay = float(og.get("weight_decay", 1e-2))
        self.betas = tuple(og.get("betas", (0.9, 0.999)))
        self.adam_eps = float(og.get("eps", 1e-8))
        sch = dict(train_opt.get("scheduler", {"type": "none"}))
        self.sched_type = sch.pop("type", "none")
        if self.sched_type == "TrueCosineAnnealingLR":
            sch = dict(T_max=int(sch["T_max"]), eta_min=float(sch.get("eta_min", 0.0)))
        elif self.sched_type not in SCHEDULERS + ("none",):
            # (base_model.py:77-108: everything else raises there too)
            raise NotImplementedError(f"Scheduler {self.sched_type} is not implemented yet. (supported: {SCHEDULERS})")
        self.sched_cfg = sch
        self.total_iter = train_opt.get("total_iter")
        # train.graph_replay: "off" (default: eager launches, as every earlier round), "on", or "auto" (small batches only)
        self._graph_opt = train_opt.get("graph_replay", "off")
        self.use_grad_clip = train_opt.get("use_grad_clip", True)
        eng = self.net_g.engine
        self.exp_avg = torch.zeros_like(eng.arena.flat_p)
        self.exp_avg_sq = torch.zeros_like(eng.arena.flat_p)
        self.sqnorm = torch.zeros(ops.SQNORM_WORDS, dtype=torch.float64, device=self.device)
        self.step_count = 0
        self.cur_lr = scheduler_initial_lr(self.sched_type, self.sched_cfg, self.base_lr, self.total_iter)
        # the reference's optimizer has a SECOND, empty param group at lr * 0.1 ('module.offsets' / 'module.dcns' parameters,
        # which this network does not have: twoImage_event_recurrent_model.py:72-90); its lr is tracked only so that a
        # reference-layout `.state` carries the two groups / two-entry scheduler lists torch's load_state_dict insists on
        self.cur_lr_low = self.cur_lr * self.LOWLR_RATIO
        self.sched_epoch = 0
        # the collectives also run in a 1-rank process group (REFID_FORCE_GRADSYNC=1): lets a single-GPU box
        # exercise the exact RCCL code path of the multi-GPU job
        self.dist_on = self.world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                          and os.environ.get("REFID_FORCE_GRADSYNC") == "1")
        self.grad_sync = GradSync(eng.arena.flat_g, eng.arena.offsets) if self.dist_on else None
        if self._graph_opt in ("on", "auto", True):
            self.set_graph_mode("auto" if self._graph_opt == "auto" else True)
        if self.dist_on:            # DDP's parameter broadcast from rank 0 (base_model.py:66-72)
            torch.distributed.broadcast(eng.arena.flat_p, src=0)
            eng.mark_params_changed()

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        """base_model.py:158-180: scheduler.step() from the second iteration on; during warm-up the lr is the
        INITIAL lr scaled linearly (init_lr / warmup_iter * current_iter), not the scheduled one."""
        if current_iter > 1:
            self.sched_epoch += 1
            self.cur_lr = scheduler_lr(self.sched_type, self.sched_cfg, self.sched_epoch, self.base_lr, self.cur_lr,
                                       self.total_iter)
            self.cur_lr_low = scheduler_lr(self.sched_type, self.sched_cfg, self.sched_epoch, self.base_lr * self.LOWLR_RATIO,
                                           getattr(self, "cur_lr_low", self.base_lr * self.LOWLR_RATIO), self.total_iter)
        if current_iter < warmup_iter:
            self.cur_lr = self.base_lr / warmup_iter * current_iter
            self.cur_lr_low = self.base_lr * self.LOWLR_RATIO / warmup_iter * current_iter

    def get_current_learning_rate(self):
        return [self.cur_lr]

    # ---- data -------------------------------------------------------------------------------------
    def feed_data(self, data):
        self.lq = data["lq"].to(self.device, non_blocking=True)
        self.voxel = data["voxel"].to(self.device, non_blocking=True)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)

    # ---- S2: one optimisation step ----------------------------------------------------------------
    # ---- hipGraph replay of the step (MI355X: ~5000 kernel launches per step; at 1-2 samples per GPU the Python /
    # ctypes enqueue is as long as the GPU work, and 8 ranks contend for the host cores) -----------------------------
    def set_graph_mode(self, on):
        """on: capture the whole train step (zero_grad .. AdamW) into hipGraphs at the next optimize_parameters() and
        replay it afterwards -- one graph on a single GPU; three graphs (forward + forward-sweep BPTT | backward-sweep
        BPTT | clip + AdamW) with the two RCCL all-reduce phases issued eagerly in between when data-parallel.  The
        captured step is re-captured when the input shapes change.  off: eager launches (the default)."""
        if on == "auto":                           # decided per batch geometry at the next optimize_parameters()
            self._graph_auto = hasattr(self.net_g.engine, "backward_early")
            on = False
        else:
            self._graph_auto = False
        on = bool(on)
        if on and not hasattr(self.net_g.engine, "backward_early"):
            raise NotImplementedError("graph mode is implemented for FinalBidirectionAttenfusion's engine only")
        if on != getattr(self, "graph_on", False) or not on:
            self._graph = None                     # drops the captured graphs and their private memory pool
        self.graph_on = on

    def _graph_capture(self):
        eng = self.net_g.engine
        dev = self.device
        g = dict(key=(tuple(self.lq.shape), tuple(self.voxel.shape), tuple(self.gt.shape)))
        g["lq"], g["voxel"], g["gt"] = (torch.empty_like(t, device=dev).copy_(t) for t in (self.lq, self.voxel, self.gt))
        g["hyper"] = torch.zeros(4, dtype=torch.float32, device=dev)
        # per-step scalars travel through a small ring of pinned buffers (the host may run several replays ahead of the
        # GPU; a slot is rewritten only after the copy that last read it has executed)
        g["hyper_host"] = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(8)]
        g["hyper_evt"] = [None] * 8
        max_norm = 0.01 if self.use_grad_clip else 0.0
        # warm-up WITHOUT a parameter update: every lazily created resource (kernel attributes, split-K / weight-
        # gradient slabs, side stream) exists before capture; nothing may allocate with hipMalloc or synchronise inside
        eng.zero_grad()
        pred = eng.forward(g["lq"], g["voxel"], save=True)
        saved, self.gt = self.gt, g["gt"]
        try:
            gpred, _, _ = self._loss_and_grad(pred)
            eng.backward(gpred)
            torch.cuda.synchronize()
            del pred, gpred
            graphs = []

            def seg(fn, pool):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, pool=pool):
                    out = fn()
                graphs.append(gr)
                return out

            def part1():
                eng.mark_params_changed()                        # the weights change every step: repack is part of the graph
                eng.zero_grad()
                pred = eng.forward(g["lq"], g["voxel"], save=True)
                gpred, loss_sum, n = self._loss_and_grad(pred)
                return pred, loss_sum, n, eng.backward_early(gpred)

            def part3():
                if max_norm > 0:
                    ops.grad_sqnorm(eng.arena.flat_g, out=self.sqnorm)
                ops.clip_adamw_dev(eng.arena.flat_p, eng.arena.flat_g, self.exp_avg, self.exp_avg_sq, self.sqnorm,
                                   g["hyper"], max_norm=max_norm, betas=self.betas, eps=self.adam_eps,
                                   weight_decay=self.weight_decay, grad_scale=1.0 / self.world)

            if self.dist_on:
                g["pred"], g["loss_sum"], g["n"], state = seg(part1, None)
                pool = graphs[0].pool()
                seg(lambda: eng.backward_late(state), pool)
                seg(part3, pool)
            else:
                def whole():
                    pred, loss_sum, n, state = part1()
                    eng.backward_late(state)
                    part3()
                    return pred, loss_sum, n
                g["pred"], g["loss_sum"], g["n"] = seg(whole, None)
            state = None
        finally:
            self.gt = saved
        g["graphs"] = graphs
        g["alloc_epoch"] = ops.ALLOC_EPOCH       # scratch buffers baked into the graphs are the current ones
        return g

    def _step_graph(self):
        g = getattr(self, "_graph", None)
        key = (tuple(self.lq.shape), tuple(self.voxel.shape), tuple(self.gt.shape))
        if g is None or g["key"] != key or g["alloc_epoch"] != ops.ALLOC_EPOCH:
            # (an eager call at a larger geometry since the capture replaced a split-K workspace / weight-gradient slab
            #  buffer whose address the graphs hold: capture again)
            self._graph = None
            try:
                g = self._graph = self._graph_capture()
            except Exception as ex:                  # noqa: BLE001
                # "auto" must never cost a run: a capture that fails (before any parameter update or collective of this step:
                # the capture's warm-up pass runs without either) sends this and every later step down the eager path; an
                # explicit set_graph_mode(True) raises
                if not getattr(self, "_graph_auto", False):
                    raise
                import warnings
                warnings.warn(f"graph_replay auto: capture failed ({type(ex).__name__}: {ex}); continuing with eager launches")
                self.set_graph_mode(False)
                self.graph_fallback = f"{type(ex).__name__}: {ex}"
                torch.cuda.synchronize()
                return False
        else:
            g["lq"].copy_(self.lq, non_blocking=True)
            g["voxel"].copy_(self.voxel, non_blocking=True)
            g["gt"].copy_(self.gt, non_blocking=True)
        self.step_count += 1
        slot = self.step_count % len(g["hyper_host"])
        if g["hyper_evt"][slot] is not None:
            g["hyper_evt"][slot].synchronize()
        h = g["hyper_host"][slot]
        # the bias corrections exactly as refid_clip_adamw forms them (float betas, double pow, one rounding to float): the
        # replayed step then equals the eager step bit for bit
        b1, b2 = (float(torch.tensor(b, dtype=torch.float32)) for b in self.betas)
        h[0] = self.cur_lr
        h[1] = 1.0 - b1 ** self.step_count
        h[2] = math.sqrt(1.0 - b2 ** self.step_count)
        g["hyper"].copy_(h, non_blocking=True)
        g["hyper_evt"][slot] = torch.cuda.Event()
        g["hyper_evt"][slot].record()
        gr = g["graphs"]
        gr[0].replay()
        if self.dist_on:
            self.grad_sync("early")
            gr[1].replay()
            self.grad_sync("late")
            gr[2].replay()
            torch.distributed.all_reduce(g["loss_sum"])
        self.net_g.engine.mark_params_changed()
        self.output = g["pred"]
        self._loss_sum, self._loss_n = g["loss_sum"], g["n"]
        self.log_dict = None

    def _bound_run_ahead(self):
        """At most ONE step in flight.  The host enqueues a step ~5x faster than the GPU executes it; left alone it runs many
        steps ahead, and every step's BPTT stash (~90 GB at B=8) is then allocated before the previous step's blocks -- freed by
        Python, but still recorded on the side streams that read them -- may be reused: the caching allocator grows to the whole
        288 GB (284 GiB reserved in a 20-step run), and one more tensor per step (the prefetched batch) tips it into
        free-everything-and-retry, 3 s per step.  Waiting for the previous step's end event costs one launch latency (the GPU is
        fed again within microseconds) and keeps the footprint at one step's 154 GiB.  The reference's loop has the same bound:
        its `reduce_loss_dict` calls `.item()` every iteration (base_model.py:325-350)."""
        ev = getattr(self, "_step_done", None)
        if ev is not None and RUN_AHEAD_BOUND:
            ev.synchronize()

    def _mark_step_end(self):
        self._step_done = torch.cuda.Event()
        self._step_done.record()

    # "auto": replay the step from hipGraphs while the per-GPU batch is small -- B H W <= GRAPH_AUTO_MAX_PIX (two samples at
    # 256 x 256): there a step is ~4800 launches of ~17 us of GPU time each and the Python / ctypes enqueue is part of what the
    # step waits for (round 6, B=1: 83.1 replayed vs 85.9-87.5 ms eager on one box); at B=8 the GPU is the limiter and the
    # replay buys nothing.  Same results as eager launches (tests/test_hip_ddp.py, test_hip_train_step.py).
    GRAPH_AUTO_MAX_PIX = 2 * 256 * 256

    def optimize_parameters(self, current_iter):
        self._bound_run_ahead()
        if getattr(self, "_graph_auto", False):
            small = self.lq.shape[0] * self.lq.shape[-2] * self.lq.shape[-1] <= self.GRAPH_AUTO_MAX_PIX
            if small != getattr(self, "graph_on", False):
                self.set_graph_mode(small)
                self._graph_auto = True
        if getattr(self, "graph_on", False):
            if self._step_graph() is not False:
                return self._mark_step_end()
        eng = self.net_g.engine
        eng.zero_grad()                                          # optimizer_g.zero_grad()
        pred = eng.forward(self.lq, self.voxel, save=True)       # net_g(x=lq, event=voxel)
        gpred, loss_sum, n = self._loss_and_grad(pred)           # cri_pix + d/dpred
        eng.backward(gpred, grad_sync=self.grad_sync)            # l_total.backward() (+ RCCL all-reduce)
        flat_g = eng.arena.flat_g
        max_norm = 0.01 if self.use_grad_clip else 0.0           # clip_grad_norm_(params, 0.01)
        if max_norm > 0:
            ops.grad_sqnorm(flat_g, out=self.sqnorm)
        self.step_count += 1
        ops.clip_adamw(eng.arena.flat_p, flat_g, self.exp_avg, self.exp_avg_sq, self.sqnorm, max_norm=max_norm,
                       lr=self.cur_lr, betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                       step=self.step_count, grad_scale=1.0 / self.world)     # optimizer_g.step()
        eng.mark_params_changed()
        self.output = pred
        # reduce_loss_dict (base_model.py:325-350) runs inside optimize_parameters on EVERY rank (a loop that only
        # logs on rank 0 must not desynchronise the collectives); only the .item() is deferred to get_current_log
        if self.dist_on:
            torch.distributed.all_reduce(loss_sum)          # 8 bytes; every rank ends up with the mean, not only rank 0
        self._loss_sum, self._loss_n = loss_sum, n
        self.log_dict = None
        self._mark_step_end()

    def _loss_and_grad(self, pred):
        """Returns (dL/dpred, un-weighted loss sum (1-element tensor), count): l_pix = weight * sum / count."""
        gpred = torch.empty_like(pred)
        n = pred.numel()
        loss_sum = ops.charbonnier(pred, self.gt.contiguous(), gpred, eps=self.loss_eps, grad_scale=self.loss_weight / n)
        return gpred, loss_sum, n

    def get_current_log(self):
        """The loss of the last step, averaged over ranks (reduce_loss_dict).  No collective here -- it ran inside
        optimize_parameters -- just one device sync for the .item()."""
        if self.log_dict is None:
            l = float(self._loss_sum.item())
            if self.dist_on:
                l /= self.world
            self.log_dict = OrderedDict(l_pix=l * self.loss_weight / self._loss_n)
        return self.log_dict

    def grad_norm(self):
        return math.sqrt(float(self.sqnorm[0].item())) / self.world

    # ---- S4: evaluation -----------------------------------------------------------------------------
    def test(self):
        self.net_g.eval()
        with torch.no_grad():
            n = self.lq.size(0)
            m = self.opt.get("val", {}).get("max_minibatch", n) or n
            outs, i = [], 0
            while i < n:
                j = min(i + m, n)
                outs.append(self.net_g(x=self.lq[i:j], event=self.voxel[i:j]))
                i = j
            self.output = torch.cat(outs, dim=0)
        self.net_g.train()

    # ---- checkpoints (state-dict key names are a compatibility contract) ------
    def save_network(self, net, net_label, current_iter, param_key="params"):
        """base_model.py:188-219: {opt['path']['models']}/{net_label}_{iter|latest}.pth holding {param_key: state_dict}
        with any 'module.' prefix removed; rank 0 only (@master_only)."""
        if self.rank != 0:
            return None
        if current_iter == -1:
            current_iter = "latest"
        save_path = os.path.join(self.opt["path"]["models"], f"{net_label}_{current_iter}.pth")
        net = getattr(net, "module", net) if isinstance(net, (torch.nn.DataParallel,
                                                               torch.nn.parallel.DistributedDataParallel)) else net
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v.detach().cpu().clone())
                         for k, v in net.state_dict().items())
        torch.save({param_key: sd}, save_path)
        return save_path

    def save(self, epoch, current_iter):
        """twoImage_event_recurrent_model.py:552-554."""
        self.save_network(self.net_g, "net_g", current_iter)
        self.save_training_state(epoch, current_iter)

    def save_training_state(self, epoch, current_iter, reference_layout=None):
        """base_model.py:283-306: {opt['path']['training_states']}/{iter}.state with one optimizer / scheduler entry.
        Default: the fused AdamW's state as it lives here, two flat arenas.  reference_layout=True (or
        opt['path']['state_layout'] == 'reference'): the layout the REFERENCE writes -- `torch.optim.AdamW.state_dict()`
        (per-parameter `step` / `exp_avg` / `exp_avg_sq` keyed by the parameter's index in named_parameters order; TWO
        param groups like the reference's optimizer, twoImage_event_recurrent_model.py:88-90: every parameter in the first,
        the second -- lr * 0.1 -- empty) and a torch scheduler-style dict with two-entry `base_lrs` / `_last_lr` -- so basicsr's own `resume_training` (base_model.py:308-323:
        `optimizer.load_state_dict` / `scheduler.load_state_dict`) can continue a run started here.  `resume_training`
        below reads both."""
        if self.rank != 0 or current_iter == -1:
            return None
        if reference_layout is None:
            reference_layout = self.opt.get("path", {}).get("state_layout") == "reference"
        if reference_layout:
            arena = self.net_g.engine.arena
            m, v = self.exp_avg.cpu(), self.exp_avg_sq.cpu()
            st = {}
            for i, (k, (off, n)) in enumerate(arena.offsets.items()):
                st[i] = {"step": torch.tensor(float(self.step_count)),
                         "exp_avg": m[off:off + n].view(arena.shapes[k]).clone(),
                         "exp_avg_sq": v[off:off + n].view(arena.shapes[k]).clone()}
            group = {"lr": self.cur_lr, "betas": tuple(self.betas), "eps": self.adam_eps, "weight_decay": self.weight_decay,
                     "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                     "fused": None, "initial_lr": self.base_lr, "params": list(range(len(st)))}
            low = dict(group, lr=self.cur_lr_low, initial_lr=self.base_lr * self.LOWLR_RATIO, params=[])
            sched = {"last_epoch": self.sched_epoch, "_step_count": self.sched_epoch + 1,
                     "base_lrs": [self.base_lr, self.base_lr * self.LOWLR_RATIO],
                     "_last_lr": [self.cur_lr, self.cur_lr_low], "_get_lr_called_within_step": False}
            sched.update({k: v for k, v in self.sched_cfg.items() if isinstance(v, (int, float, list, tuple))})
            state = {"epoch": epoch, "iter": current_iter,
                     "optimizers": [{"state": st if self.step_count else {}, "param_groups": [group, low]}],
                     "schedulers": [sched]}
        else:
            state = {"epoch": epoch, "iter": current_iter,
                     "optimizers": [{"type": "refid_amd.fused_adamw", "step": self.step_count,
                                     "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()}],
                     "schedulers": [{"type": self.sched_type, "last_epoch": self.sched_epoch, "lr": self.cur_lr,
                                     "lr_low": self.cur_lr_low}]}
        save_path = os.path.join(self.opt["path"]["training_states"], f"{current_iter}.state")
        torch.save(state, save_path)
        return save_path

    def resume_training(self, resume_state):
        """base_model.py:308-323.  Accepts this class's own `.state` files (flat AdamW arenas) AND the reference's
        (`torch.optim.AdamW.state_dict()` / `scheduler.state_dict()`, base_model.py:297-303): the per-parameter
        `exp_avg` / `exp_avg_sq` are laid into the arenas in parameter order (= state-dict order of the network)."""
        assert len(resume_state["optimizers"]) == 1, "Wrong lengths of optimizers"
        assert len(resume_state["schedulers"]) == 1, "Wrong lengths of schedulers"
        o, sc = resume_state["optimizers"][0], resume_state["schedulers"][0]
        arena = self.net_g.engine.arena
        if o.get("type") == "refid_amd.fused_adamw":
            if o["exp_avg"].numel() != self.exp_avg.numel() or o["exp_avg_sq"].numel() != self.exp_avg_sq.numel():
                raise ValueError(f"resume_training: the saved AdamW arenas hold {o['exp_avg'].numel()} values, this network's "
                                 f"{self.exp_avg.numel()} (different img_chn / base_num_channels?)")
            self.step_count = int(o["step"])
            self.exp_avg.copy_(o["exp_avg"])
            self.exp_avg_sq.copy_(o["exp_avg_sq"])
        elif "state" in o and "param_groups" in o:
            # torch keys the state by the parameter's index in param_groups (= named_parameters order, setup_optimizers
            # :67-95) and creates an entry lazily, at the first step() that sees a gradient for it.  atten_fuse.se_2 is
            # never used in forward (fusion_modules.py:261 vs :312-315): it only has a gradient because of the wrapper's
            # `0 * sum(p.sum())` term (:301), so a `.state` written by a loop without that term holds SPARSE integer
            # keys.  Parameters without an entry keep zero moments -- exactly what torch creates on first use.
            keys = list(arena.offsets)
            st = o["state"]
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            for idx, e in st.items():
                i = int(idx)
                if not 0 <= i < len(keys):
                    raise ValueError(f"resume_training: optimizer state index {i} outside this network's {len(keys)} "
                                     "parameters")
                k = keys[i]
                off, n = arena.offsets[k]
                if e["exp_avg"].numel() != n or e["exp_avg_sq"].numel() != n:
                    raise ValueError(f"resume_training: optimizer state {i} has {e['exp_avg'].numel()} values, parameter "
                                     f"{k} has {n}")
                self.exp_avg[off:off + n].copy_(e["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
                steps.add(int(e["step"]))
            if len(steps) > 1:
                raise ValueError(f"resume_training: per-parameter step counts differ ({sorted(steps)[:4]} ...)")
            # (an empty state = an optimizer that never stepped)
            self.step_count = steps.pop() if steps else 0
        else:
            raise ValueError("resume_training: unknown optimizer entry (neither refid_amd.fused_adamw arenas nor a "
                             "torch.optim.AdamW state_dict)")
        if "lr" in sc:
            self.sched_epoch, self.cur_lr = int(sc["last_epoch"]), float(sc["lr"])
            self.cur_lr_low = float(sc.get("lr_low", self.cur_lr * self.LOWLR_RATIO))
        elif "_last_lr" in sc:                                   # torch scheduler.state_dict()
            self.sched_epoch, self.cur_lr = int(sc["last_epoch"]), float(sc["_last_lr"][0])
            self.cur_lr_low = float(sc["_last_lr"][1]) if len(sc["_last_lr"]) > 1 else self.cur_lr * self.LOWLR_RATIO
        else:
            raise ValueError("resume_training: unknown scheduler entry (no 'lr' / '_last_lr')")

    def load_network(self, net, load_path, strict=True, param_key="params"):
        load_net = torch.load(load_path, map_location="cpu")
        if param_key is not None and param_key in load_net:
            load_net = load_net[param_key]
        load_net = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in load_net.items())
        net.load_state_dict(load_net, strict=strict)


class ImageEventRestorationModel(TwoImageEventRecurrentRestorationModel):
    """Single-image event deblurring model (reference image_event_restoration_model.py:21-345) around
    ``SingleMultiConnectEVHINet``: same method names (``feed_data``, ``optimize_parameters``, ``test``, ...), 4-D tensors,
    ``net_g`` returns a list whose last element is the output (:275-280, :337-340).  Pixel losses: ``PSNRLoss``
    (losses.py:95-120, toY=False) and ``CharbonnierLoss``; clip_grad_norm_(0.01) + AdamW as in :317-320."""
    PIXEL_LOSSES = ("PSNRLoss", "CharbonnierLoss")

    def _loss_and_grad(self, pred):
        if self.pixel_type != "PSNRLoss":
            return super()._loss_and_grad(pred)
        # PSNRLoss = w * 10/ln10 * mean_b log(mse_b + 1e-8) and its gradient: two kernels (csrc/train.hip)
        gpred = torch.empty_like(pred)
        loss = ops.psnr_loss(pred.contiguous(), self.gt.contiguous(), gpred, weight=self.loss_weight)
        return gpred, loss, (self.loss_weight or 1.0)        # the kernel's value already carries the weight: l_pix = w * value / w

    def test(self):
        self.net_g.eval()
        with torch.no_grad():
            n = self.lq.size(0)
            m = self.opt.get("val", {}).get("max_minibatch", n) or n
            outs, i = [], 0
            while i < n:
                j = min(i + m, n)
                pred = self.net_g(x=self.lq[i:j], event=self.voxel[i:j])
                outs.append(pred[-1] if isinstance(pred, list) else pred)
                i = j
            self.output = torch.cat(outs, dim=0)
        self.net_g.train()
