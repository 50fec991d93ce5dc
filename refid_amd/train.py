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


class TwoImageEventRecurrentRestorationModel:
    PIXEL_LOSSES = ("CharbonnierLoss",)            # the only one the reference's configs for this model use

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
        self.sched_type = sch.get("type", "none")
        if self.sched_type in ("CosineAnnealingLR", "TrueCosineAnnealingLR"):
            self.t_max, self.eta_min = int(sch["T_max"]), float(sch.get("eta_min", 0.0))
        elif self.sched_type != "none":
            raise NotImplementedError(f"Scheduler {self.sched_type} is not implemented yet.")
        self.use_grad_clip = train_opt.get("use_grad_clip", True)
        eng = self.net_g.engine
        self.exp_avg = torch.zeros_like(eng.arena.flat_p)
        self.exp_avg_sq = torch.zeros_like(eng.arena.flat_p)
        self.sqnorm = torch.zeros(ops.SQNORM_WORDS, dtype=torch.float64, device=self.device)
        self.step_count = 0
        self.cur_lr = self.base_lr
        self.sched_epoch = 0
        # the collectives also run in a 1-rank process group (REFID_FORCE_GRADSYNC=1): lets a single-GPU box
        # exercise the exact RCCL code path of the multi-GPU job
        self.dist_on = self.world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                          and os.environ.get("REFID_FORCE_GRADSYNC") == "1")
        self.grad_sync = GradSync(eng.arena.flat_g, eng.arena.offsets) if self.dist_on else None
        if self.dist_on:            # DDP's parameter broadcast from rank 0 (base_model.py:66-72)
            torch.distributed.broadcast(eng.arena.flat_p, src=0)
            eng.mark_params_changed()

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        """base_model.py:158-180: scheduler.step() from the second iteration on (+ linear warm-up)."""
        if current_iter > 1:
            self.sched_epoch += 1
        lr = self.base_lr
        if self.sched_type != "none":
            lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.sched_epoch / self.t_max)) / 2
        if current_iter < warmup_iter:
            lr = lr / warmup_iter * current_iter
        self.cur_lr = lr

    def get_current_learning_rate(self):
        return [self.cur_lr]

    # ---- data -------------------------------------------------------------------------------------
    def feed_data(self, data):
        self.lq = data["lq"].to(self.device, non_blocking=True)
        self.voxel = data["voxel"].to(self.device, non_blocking=True)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)

    # ---- S2: one optimisation step ----------------------------------------------------------------
    def optimize_parameters(self, current_iter):
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
        self._loss_sum, self._loss_n = loss_sum, n
        self.log_dict = None

    def _loss_and_grad(self, pred):
        """Returns (dL/dpred, un-weighted loss sum (1-element tensor), count): l_pix = weight * sum / count."""
        gpred = torch.empty_like(pred)
        n = pred.numel()
        loss_sum = ops.charbonnier(pred, self.gt.contiguous(), gpred, eps=self.loss_eps, grad_scale=self.loss_weight / n)
        return gpred, loss_sum, n

    def get_current_log(self):
        """reduce_loss_dict (base_model.py:325-350): mean over ranks, evaluated lazily (one sync)."""
        if self.log_dict is None:
            l = self._loss_sum.clone()
            if self.dist_on:
                torch.distributed.all_reduce(l)
                l /= self.world
            self.log_dict = OrderedDict(l_pix=float(l.item()) * self.loss_weight / self._loss_n)
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
    def save_network(self, net, save_path, param_key="params"):
        sd = OrderedDict((k.replace("module.", "", 1) if k.startswith("module.") else k, v.detach().cpu().clone())
                         for k, v in net.state_dict().items())
        if self.rank == 0:
            torch.save({param_key: sd}, save_path)

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
        # PSNRLoss = w * 10/ln10 * mean_b log(mse_b + 1e-8): a (B,C,H,W)-sized elementwise expression, on the GPU
        scale = 10.0 / math.log(10.0)
        d = pred - self.gt
        mse = (d * d).mean(dim=(1, 2, 3))
        loss = scale * torch.log(mse + 1e-8).mean()
        coef = self.loss_weight * scale * 2.0 / (d[0].numel() * d.shape[0]) / (mse + 1e-8)
        return d * coef.view(-1, 1, 1, 1), loss.double().reshape(1), 1

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
