"""Drop-in ``FinalBidirectionAttenfusion`` backed by the MI355X HIP engine.

Mirrors the reference class of the same name
(/root/reference/basicsr/models/archs/XXNet_final_attenfusion_arch.py:81-218): same
constructor keywords, the same 183 state-dict keys/shapes (so released ``['params']``
checkpoints load with ``strict=True``), called as ``net_g(x=lq, event=voxel)``
(twoImage_event_recurrent_model.py:276,324), returns a new (B,T,out_chn,H,W) tensor that is
attached to autograd in grad mode (``refid_amd/autograd.py``: parameter gradients are delivered through
autograd, so ``DistributedDataParallel`` -- base_model.py:66-72 -- wraps it like any module).  All arithmetic
runs in librefid_hip.so; there is no torch/CPU fallback -- calling it without the built extension or off-GPU raises.
"""
import torch
from torch import nn

from ..registry import ARCH_REGISTRY
from .. import autograd as hip_autograd
from ..engine import Engine, param_shapes
from .._lib import RefidHipError


class _Node(nn.Module):
    """Bare container used to reproduce the reference's parameter paths."""

    def extra_repr(self):
        return ""


@ARCH_REGISTRY.register()
class FinalBidirectionAttenfusion(nn.Module):
    def __init__(self, img_chn, ev_chn, out_chn=3, skip_type='sum', recurrent_block_type='convlstm',
                 activation='sigmoid', num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None,
                 use_recurrent_upsample_conv=True, num_block=3, use_first_dcn=False, use_reversed_voxel=False,
                 compute_dtype='fp32'):
        """compute_dtype (extension, keyword-only in practice; YAML: network_g.compute_dtype): 'fp32'
        (default, the reference's arithmetic) or 'bf16' (BASELINE config 3: bf16 matrix-core operands for
        the conv forward / input gradients, fp32 everything else)."""
        super().__init__()
        if compute_dtype not in ('fp32', 'bf16x3', 'bf16'):
            raise ValueError(f"compute_dtype must be 'fp32', 'bf16x3' or 'bf16', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        assert ev_chn > 0 and img_chn > 0 and out_chn > 0                      # arch:45-47
        unsupported = []
        if num_encoders not in (2, 3, 4):
            unsupported.append(f"num_encoders={num_encoders} (2, 3 or 4: level 1 is the attention-fusion level; the YAMLs use 3, "
                               "the reference ctor's default is 4)")
        if num_block < 1:
            unsupported.append(f"num_block={num_block} (at least one ResidualBlockNoBN per trunk)")
        if skip_type != 'sum':
            unsupported.append(f"skip_type={skip_type!r}")
        if norm is not None:
            unsupported.append(f"norm={norm!r}")
        if not use_recurrent_upsample_conv:
            unsupported.append("use_recurrent_upsample_conv=False")
        if num_residual_blocks < 1:
            unsupported.append("num_residual_blocks<1")
        if unsupported:
            raise NotImplementedError("FinalBidirectionAttenfusion (HIP): unsupported options: " + ", ".join(unsupported)
                                      + " -- every options/*.yml of the reference uses num_encoders=3, num_block=1, skip_type='sum', "
                                      "no norm (the reference ctor's own defaults, num_encoders=4 / num_block=3, build here too)")
        # recurrent_block_type / activation / use_first_dcn / use_reversed_voxel are accepted and
        # ignored, exactly like the reference (arch:59,92; rsm:251-257)
        self.img_chn, self.ev_chn, self.out_chn = img_chn, ev_chn, out_chn
        self.base_num_channels, self.num_residual_blocks = base_num_channels, num_residual_blocks
        self.num_block, self.num_encoders = num_block, num_encoders
        self._shapes = param_shapes(img_chn, ev_chn, out_chn, base_num_channels, num_residual_blocks, num_block, num_encoders)
        self._engine = None
        self._grad_sync = None                 # optional callable(phase) run inside BPTT (refid_amd.dist.GradSync)
        self._params = {}
        for key, shape in self._shapes.items():
            node = self
            parts = key.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            p = nn.Parameter(torch.empty(shape))
            node.register_parameter(parts[-1], p)
            self._params[key] = p
        self.reset_parameters()

    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self):
        """torch defaults; ResidualBlockNoBN convs Kaiming-normal x0.1 with zero bias
        (rsm:752-753,776-800); LayerNorm 1/0; beta, gamma 0 (fm:287-288)."""
        import math
        for k, p in self._params.items():
            if ".norm" in k:
                p.fill_(1.0 if k.endswith("weight") else 0.0)
            elif k.endswith((".beta", ".gamma")):
                p.zero_()
            elif ".main.2." in k and ".conv" in k:          # every ResidualBlockNoBN of a trunk (rsm:752-753)
                if k.endswith("weight"):
                    nn.init.kaiming_normal_(p)
                    p.mul_(0.1)
                else:
                    p.zero_()
            else:
                w = self._params[k[:-4] + "weight"] if k.endswith("bias") else p
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
                bound = 1.0 / math.sqrt(fan_in)
                p.uniform_(-bound, bound)
        self._touch()

    def _touch(self):
        if self._engine is not None:
            self._engine.mark_params_changed()

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._rebind()
        return r

    def _rebind(self):
        """(Re)create the engine on the parameters' device and alias the parameters to its arena."""
        dev = next(iter(self._params.values())).device
        if dev.type != "cuda":
            self._engine = None
            return
        if self._engine is not None and self._engine.device == dev and \
                all(p.data_ptr() == self._engine.arena.p(k).data_ptr() for k, p in self._params.items()):
            return
        eng = Engine(self.img_chn, self.ev_chn, self.out_chn, self.base_num_channels, self.num_residual_blocks,
                     device=dev, compute_dtype=self.compute_dtype, num_block=self.num_block, num_encoders=self.num_encoders)
        with torch.no_grad():
            for k, p in self._params.items():
                if p.dtype != torch.float32:
                    raise RefidHipError("FinalBidirectionAttenfusion (HIP): parameters must stay float32")
                eng.arena.p(k).copy_(p.data)
                p.data = eng.arena.p(k)
                p.grad = None
        self._engine = eng

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._touch()
        return r

    # ---------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            raise RefidHipError("FinalBidirectionAttenfusion (HIP): the network must be on a CUDA (ROCm) device; "
                                "there is no CPU path")
        return self._engine

    def notify_params_changed(self):
        """Call after modifying parameters in place outside an optimizer this module knows of."""
        self._touch()

    def forward(self, x, event):
        eng = self.engine
        # parameters are updated in place by optimizers: repack when any version counter moved
        ver = sum(p._version for p in self._params.values())
        if ver != getattr(self, "_seen_version", None):
            eng.mark_params_changed()
            self._seen_version = ver
        grad_mode = torch.is_grad_enabled() and any(p.requires_grad for p in self._params.values())
        if not grad_mode:
            return eng.forward(x, event, save=False)
        # every parameter is an input of the autograd node and receives its gradient through autograd
        # (hooks, DistributedDataParallel, gradient accumulation all behave as for any nn.Module)
        return hip_autograd.apply(self, x, event)
