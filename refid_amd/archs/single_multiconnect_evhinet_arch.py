"""Drop-in ``SingleMultiConnectEVHINet`` backed by the MI355X HIP engine.

Mirrors the reference class of the same name (archs/single_multiconnect_evhinet_arch.py:66-167): same constructor
keywords, the same 156 state-dict keys/shapes in the same order (wf=64), called as ``net_g(x=lq, event=voxel)``
(image_event_restoration_model.py:275,337); returns ``[out_1]`` -- a list with the one (B,in_chn,H,W) tensor, attached to
autograd in grad mode -- exactly as the reference's "single version" forward does.  No torch/CPU fallback.
"""
import math

import torch
from torch import nn

from ..registry import ARCH_REGISTRY
from .. import autograd as hip_autograd
from .._lib import RefidHipError
from ..evhinet import EvhinetEngine, param_shapes


class _Node(nn.Module):
    def extra_repr(self):
        return ""


@ARCH_REGISTRY.register()
class SingleMultiConnectEVHINet(nn.Module):
    def __init__(self, in_chn=3, ev_chn=6, wf=64, depth=3, fac_place=2, fac_kernel_size=1, fac_before_downsample=True,
                 event_feature_transfer=False, relu_slope=0.2, hin_position_left=0, hin_position_right=4):
        super().__init__()
        unsupported = []
        if fac_kernel_size != 1:
            unsupported.append(f"fac_kernel_size={fac_kernel_size}")      # FAC_bias ignores it anyway (au:421-426)
        if not fac_before_downsample:
            unsupported.append("fac_before_downsample=False")
        if unsupported:
            raise NotImplementedError("SingleMultiConnectEVHINet (HIP): unsupported options: " + ", ".join(unsupported))
        # event_feature_transfer is stored and never read by the reference's forward (evh:73,83)
        self._cfg = dict(in_chn=in_chn, ev_chn=ev_chn, wf=wf, depth=depth, fac_place=fac_place, relu_slope=relu_slope,
                         hin_position_left=hin_position_left, hin_position_right=hin_position_right)
        self._shapes = param_shapes(in_chn, ev_chn, wf, depth, fac_place, hin_position_left, hin_position_right)
        self._engine = None
        self._grad_sync = None
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

    @torch.no_grad()
    def reset_parameters(self):
        """torch defaults (Conv2d / ConvTranspose2d: U(+-1/sqrt(fan_in)); InstanceNorm2d affine: 1 / 0)."""
        for k, p in self._params.items():
            if ".norm." in k:
                p.fill_(1.0 if k.endswith("weight") else 0.0)
                continue
            w = self._params[k[:-4] + "weight"] if k.endswith("bias") else p
            fan_in = (w.shape[0] if ".up." in k else w.shape[1]) * w.shape[2] * w.shape[3]
            if ".up." in k:                      # ConvTranspose2d: fan_in is computed from weight.size(1) * k * k
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        self._touch()

    def _initialize(self):
        """evh:174-180: orthogonal init with the leaky-relu gain, zero bias."""
        gain = nn.init.calculate_gain('leaky_relu', 0.20)
        with torch.no_grad():
            for k, p in self._params.items():
                if p.dim() == 4 and ".up." not in k:
                    nn.init.orthogonal_(p, gain=gain)
                elif k.endswith("bias") and ".norm." not in k and ".up." not in k:
                    p.zero_()
        self._touch()

    def _touch(self):
        if self._engine is not None:
            self._engine.mark_params_changed()

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._rebind()
        return r

    def _rebind(self):
        dev = next(iter(self._params.values())).device
        if dev.type != "cuda":
            self._engine = None
            return
        if self._engine is not None and self._engine.device == dev and \
                all(p.data_ptr() == self._engine.arena.p(k).data_ptr() for k, p in self._params.items()):
            return
        eng = EvhinetEngine(device=dev, **self._cfg)
        with torch.no_grad():
            for k, p in self._params.items():
                if p.dtype != torch.float32:
                    raise RefidHipError("SingleMultiConnectEVHINet (HIP): parameters must stay float32")
                eng.arena.p(k).copy_(p.data)
                p.data = eng.arena.p(k)
                p.grad = None
        self._engine = eng

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._touch()
        return r

    @property
    def engine(self):
        if self._engine is None:
            raise RefidHipError("SingleMultiConnectEVHINet (HIP): the network must be on a CUDA (ROCm) device; "
                                "there is no CPU path")
        return self._engine

    def notify_params_changed(self):
        self._touch()

    def get_input_chn(self, in_chn):
        return in_chn

    def forward(self, x, event, mask=None):
        eng = self.engine
        ver = sum(p._version for p in self._params.values())
        if ver != getattr(self, "_seen_version", None):
            eng.mark_params_changed()
            self._seen_version = ver
        grad_mode = torch.is_grad_enabled() and any(p.requires_grad for p in self._params.values())
        if not grad_mode:
            return [eng.forward(x, event, save=False)]
        return [hip_autograd.apply(self, x, event)]
