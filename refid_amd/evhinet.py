"""HIP engine for ``SingleMultiConnectEVHINet``, the repository's single-image event
deblurring network: reference archs/single_multiconnect_evhinet_arch.py ("evh") + arch_util.py::FAC_bias.

A feed-forward HINet U-Net (depth 3) whose two shallow encoder outputs are modulated by an event encoder
(``feat * w + b``), one decoder, SAM image head.  Reuses the conv tiles of the recurrent network
(``engine.ConvOp``: Winograd 3x3, pointwise 1x1, 4x4/s2, ConvTranspose 2x2, their input / weight gradients) and
adds two streaming kernels (csrc/evhinet.hip): half-instance-norm + LeakyReLU and FAC_bias, forward + backward.

Only what reaches the output is evaluated (results identical to the reference, see oracle/evhinet_oracle.py):
``down_path_ev[2]``, ``sam12.conv1/conv3`` are computed-and-discarded by the reference; stage 2 / csff / cat12 /
last are never called.  Their parameters exist in the state dict and receive zero gradients.
"""
from collections import OrderedDict

import torch

from . import ops
from ._lib import RefidHipError
from .engine import ConvOp, ParamArena, WGRAD_STREAM, _pad4, finish_wgrads


def param_shapes(in_chn=3, ev_chn=6, wf=64, depth=3, fac_place=2, hin_position_left=0, hin_position_right=4):
    """State-dict keys / shapes in the reference's registration order (evh:68-125, 200-224, 262-285, 318-322)."""
    S = OrderedDict()

    def conv(name, co, ci, k, bias=True):
        S[name + ".weight"] = (co, ci, k, k)
        if bias:
            S[name + ".bias"] = (co,)

    def block(name, ci, co, downsample, csff, hin, ev=False):
        conv(name + ".identity", co, ci, 1)
        conv(name + ".conv_1", co, ci, 3)
        conv(name + ".conv_2", co, co, 3)
        if ev:
            conv(name + ".conv_before_merge", 2 * co, co, 1)
        if downsample and csff:
            for n in ("csff_enc", "csff_dec", "csff_enc_mask", "csff_dec_mask"):
                conv(f"{name}.{n}", co, co, 3)
        if hin:
            S[name + ".norm.weight"] = (co // 2,)
            S[name + ".norm.bias"] = (co // 2,)
        if downsample:
            conv(name + ".downsample", co, co, 4, bias=False)

    def chans(i):
        return (wf if i == 0 else (2 ** (i - 1)) * wf), (2 ** i) * wf

    def hin(i):
        return hin_position_left <= i <= hin_position_right

    for stage, csff in (("down_path_1", False), ("down_path_2", True)):
        for i in range(depth):
            ci, co = chans(i)
            block(f"{stage}.{i}", ci, co, i + 1 < depth, csff, hin(i))
    conv("conv_01", wf, in_chn, 3)
    conv("conv_02", wf, in_chn, 3)
    for i in range(min(depth, fac_place + 1)):
        ci, co = chans(i)
        block(f"down_path_ev.{i}", ci, co, i + 1 < depth, False, hin(i), ev=True)
    conv("conv_ev1", wf, ev_chn, 3)
    for stage in ("up_path_1", "up_path_2"):
        for j, i in enumerate(reversed(range(depth - 1))):
            co = (2 ** i) * wf
            S[f"{stage}.{j}.up.weight"] = (2 * co, co, 2, 2)
            S[f"{stage}.{j}.up.bias"] = (co,)
            block(f"{stage}.{j}.conv_block", 2 * co, co, False, False, False)
    for stage in ("skip_conv_1", "skip_conv_2"):
        for j, i in enumerate(reversed(range(depth - 1))):
            conv(f"{stage}.{j}", (2 ** i) * wf, (2 ** i) * wf, 3)
    conv("sam12.conv1", wf, wf, 3)
    conv("sam12.conv2", 3, wf, 3)
    conv("sam12.conv3", wf, 3, 3)
    conv("cat12", wf, 2 * wf, 1)
    conv("last", in_chn, wf, 3)
    return S


class _Block:
    """UNetConvBlock / UNetEVConvBlock (evh:200-260, 262-315)."""

    def __init__(self, arena, name, need_dgrad=True, down=False, merge=False):
        self.name = name
        self.conv_1 = ConvOp(arena, name + ".conv_1", need_dgrad=need_dgrad)
        self.conv_2 = ConvOp(arena, name + ".conv_2")
        self.identity = ConvOp(arena, name + ".identity", need_dgrad=need_dgrad)
        self.hin = (name + ".norm.weight") in arena.shapes
        self.gamma = arena.p(name + ".norm.weight") if self.hin else None
        self.beta = arena.p(name + ".norm.bias") if self.hin else None
        self.dgamma = arena.g(name + ".norm.weight") if self.hin else None
        self.dbeta = arena.g(name + ".norm.bias") if self.hin else None
        self.down = ConvOp(arena, name + ".downsample", kind="down") if down else None
        self.merge = ConvOp(arena, name + ".conv_before_merge") if merge else None

    def ops(self):
        return [o for o in (self.conv_1, self.conv_2, self.identity, self.down, self.merge) if o is not None]


class EvhinetEngine:
    def __init__(self, in_chn=3, ev_chn=6, wf=64, depth=3, fac_place=2, relu_slope=0.2, hin_position_left=0,
                 hin_position_right=4, device=None):
        if device is None or torch.device(device).type != "cuda":
            raise RefidHipError("EvhinetEngine needs a CUDA (ROCm) device; there is no CPU path")
        if depth < 2 or fac_place < depth - 2:
            raise NotImplementedError("SingleMultiConnectEVHINet (HIP): needs depth >= 2 and fac_place >= depth - 2 "
                                      "(every encoder level that has a skip is event-modulated, as in the shipped defaults)")
        if wf % 8:
            raise NotImplementedError("SingleMultiConnectEVHINet (HIP): wf must be a multiple of 8")
        self.device = torch.device(device)
        self.in_chn, self.ev_chn, self.wf, self.depth, self.slope = in_chn, ev_chn, wf, depth, relu_slope
        self.shapes = param_shapes(in_chn, ev_chn, wf, depth, fac_place, hin_position_left, hin_position_right)
        self.arena = A = ParamArena(self.shapes, self.device)
        self.conv_01 = ConvOp(A, "conv_01", need_dgrad=False)
        self.conv_ev1 = ConvOp(A, "conv_ev1", need_dgrad=False)
        self.enc = [_Block(A, f"down_path_1.{i}", down=i + 1 < depth) for i in range(depth)]
        # deepest event block is dead code in the reference (its filter is never applied): not built;
        # the last live event block's downsample only feeds that dead block
        self.evb = [_Block(A, f"down_path_ev.{i}", down=i + 2 < depth, merge=True) for i in range(depth - 1)]
        self.up, self.upb, self.skip = [], [], []
        for j in range(depth - 1):
            self.up.append(ConvOp(A, f"up_path_1.{j}.up", kind="convT"))
            self.upb.append(_Block(A, f"up_path_1.{j}.conv_block"))
            self.skip.append(ConvOp(A, f"skip_conv_1.{j}"))
        self.sam2 = ConvOp(A, "sam12.conv2")
        self.all_ops = [self.conv_01, self.conv_ev1, self.sam2] + self.up + self.skip
        for b in self.enc + self.evb + self.upb:
            self.all_ops += b.ops()
        self.param_version, self.packed_version = 1, 0
        self.ctx = None

    # -------------------------------------------------------------------------------------------
    def mark_params_changed(self):
        self.param_version += 1

    def repack(self):
        if self.packed_version != self.param_version:
            for o in self.all_ops:
                o.repack()
            self.packed_version = self.param_version

    def zero_grad(self):
        self.arena.flat_g.zero_()

    # ---- one conv block -------------------------------------------------------------------------
    def _block_fwd(self, B, x, xb=None, save=None):
        s = self.slope
        if B.hin:
            c1 = B.conv_1.fwd(x, xb)
            a1, stats = ops.hin_lrelu_fwd(c1, B.gamma, B.beta, s)
        else:
            c1, stats = None, None
            a1 = B.conv_1.fwd(x, xb, slope_pre=s)
        oc2 = B.conv_2.fwd(a1, slope_pre=s)
        out = B.identity.fwd(x, xb, res=oc2)                     # relu_2(conv_2(.)) + identity(x)   (evh:237-238)
        if save is not None:
            save.update(x=x, xb=xb, c1=c1, a1=a1, stats=stats, oc2=oc2)
        return out

    def _block_bwd(self, B, g_out, st, need_input_grad=True):
        s = self.slope
        x, xb, a1 = st["x"], st["xb"], st["a1"]
        B.identity.wgrad(g_out, x, xb)
        gz2 = ops.act_bwd(g_out, st["oc2"], s)
        B.conv_2.wgrad(gz2, a1)
        g_a1 = B.conv_2.dgrad(gz2)
        if B.hin:
            g_c1 = ops.hin_lrelu_bwd(g_a1, a1, st["c1"], B.gamma, st["stats"], B.dgamma, B.dbeta, s)
        else:
            g_c1 = ops.act_bwd(g_a1, a1, s)
        B.conv_1.wgrad(g_c1, x, xb)
        if not need_input_grad:
            return None
        if xb is None:
            return B.conv_1.dgrad(g_c1, res=B.identity.dgrad(g_out))
        ca, cb = x.shape[3], xb.shape[3]
        g_x = B.conv_1.dgrad(g_c1, rows=(0, ca), res=B.identity.dgrad(g_out, rows=(0, ca)))
        g_xb = B.conv_1.dgrad(g_c1, rows=(ca, cb), res=B.identity.dgrad(g_out, rows=(ca, cb)))
        return g_x, g_xb

    # -------------------------------------------------------------------------------------------
    def forward(self, x, event, save=False):
        """x (B,in_chn,H,W), event (B,ev_chn,H,W) -> out_1 (B,in_chn,H,W)   (evh:127-167, single version)."""
        if x.dim() != 4 or event.dim() != 4 or x.shape[0] != event.shape[0] or x.shape[2:] != event.shape[2:]:
            raise RuntimeError(f"SingleMultiConnectEVHINet: bad input shapes x={tuple(x.shape)} event={tuple(event.shape)}")
        if x.shape[1] != self.in_chn or event.shape[1] != self.ev_chn:
            raise RuntimeError(f"SingleMultiConnectEVHINet: expected {self.in_chn}/{self.ev_chn} channels, got "
                               f"{x.shape[1]}/{event.shape[1]}")
        Bn, _, H, W = x.shape
        f = 2 ** (self.depth - 1)
        if H % f or W % f:
            raise RuntimeError(f"SingleMultiConnectEVHINet: H, W must be multiples of {f} (got {H}x{W})")
        self.repack()
        x = x.to(self.device, torch.float32).contiguous()
        event = event.to(self.device, torch.float32).contiguous()
        c = {"enc": [{} for _ in self.enc], "ev": [{} for _ in self.evb], "up": [{} for _ in self.upb]} if save else None
        img4 = ops.nchw_to_nhwc(x, _pad4(self.in_chn))
        ev_in = ops.nchw_to_nhwc(event, _pad4(self.ev_chn))
        # event encoder: filters for every encoder level that owns a skip connection
        filt, ev_out = [], []
        e1 = self.conv_ev1.fwd(ev_in)
        for i, Bk in enumerate(self.evb):
            out = self._block_fwd(Bk, e1, save=c["ev"][i] if save else None)
            filt.append(Bk.merge.fwd(out))                        # conv_before_merge, merge before downsample
            ev_out.append(out)
            if Bk.down is not None:
                e1 = Bk.down.fwd(out)
        # stage 1 encoder
        x1 = self.conv_01.fwd(img4)
        outs, encs = [], []
        for i, Bk in enumerate(self.enc):
            out = self._block_fwd(Bk, x1, save=c["enc"][i] if save else None)
            if Bk.down is not None:
                fo = ops.fac_fwd(out, filt[i])                    # FAC_bias (evh:246-247)
                outs.append(out); encs.append(fo)
                x1 = Bk.down.fwd(fo)
            else:
                x1 = out
        # decoder
        ups, bridges_in = [], []
        for j in range(self.depth - 1):
            ups.append(x1)
            up = self.up[j].fwd(x1)
            bridge = self.skip[j].fwd(encs[-j - 1])
            x1 = self._block_fwd(self.upb[j], up, bridge, save=c["up"][j] if save else None)
        out4 = torch.zeros((Bn, H, W, _pad4(self.in_chn)), dtype=torch.float32, device=self.device)
        self.sam2.fwd(x1, res=img4[..., :self.in_chn], out=out4[..., :self.in_chn])     # SAM: img = conv2(x) + x_img
        result = torch.empty((Bn, self.in_chn, H, W), dtype=torch.float32, device=self.device)
        ops.nhwc_to_nchw(out4[..., :self.in_chn], self.in_chn, result)
        if save:
            c.update(img4=img4, ev_in=ev_in, filt=filt, ev_out=ev_out, outs=outs, encs=encs, ups=ups, last=x1,
                     shape=(Bn, H, W))
            self.ctx = c
        return result

    def backward(self, gout, grad_sync=None):
        """Parameter gradients ACCUMULATE into the arena (zero_grad() first)."""
        c = self.ctx
        if c is None:
            raise RefidHipError("backward: no saved forward (call forward(save=True) first)")
        WGRAD_STREAM.pending.clear()          # leftovers of a backward that raised must never be launched
        for o in self.all_ops:                # ... nor may its half-filled slabs be added to / reduced
            o.w_calls, o.w_last, o.w_pend = 0, None, []
        Bn, H, W = c["shape"]
        gout = gout.to(self.device, torch.float32).contiguous()
        g4 = ops.nchw_to_nhwc(gout, _pad4(self.in_chn))
        self.sam2.wgrad(g4, c["last"])
        g_x1 = self.sam2.dgrad(g4)
        depth = self.depth
        g_enc = [None] * (depth - 1)
        for j in range(depth - 2, -1, -1):
            g_up, g_bridge = self._block_bwd(self.upb[j], g_x1, c["up"][j])
            lvl = depth - 2 - j                                   # encs[-j-1]
            self.skip[j].wgrad(g_bridge, c["encs"][lvl])
            g_enc[lvl] = self.skip[j].dgrad(g_bridge)
            self.up[j].wgrad(g_up, c["ups"][j])
            g_x1 = self.up[j].dgrad(g_up)
        g_filt = [None] * (depth - 1)
        for i in range(depth - 1, -1, -1):
            Bk = self.enc[i]
            if Bk.down is not None:
                Bk.down.wgrad(g_x1, c["encs"][i])
                g_fo = Bk.down.dgrad(g_x1, res=g_enc[i])
                g_out, g_filt[i] = ops.fac_bwd(g_fo, c["outs"][i], c["filt"][i])
            else:
                g_out = g_x1
            g_x1 = self._block_bwd(Bk, g_out, c["enc"][i])
        self.conv_01.wgrad(g_x1, c["img4"])
        g_e1 = None
        for i in range(depth - 2, -1, -1):
            Bk = self.evb[i]
            Bk.merge.wgrad(g_filt[i], c["ev_out"][i])
            g_out = Bk.merge.dgrad(g_filt[i])
            if Bk.down is not None:
                Bk.down.wgrad(g_e1, c["ev_out"][i])
                g_out = Bk.down.dgrad(g_e1, res=g_out)
            g_e1 = self._block_bwd(Bk, g_out, c["ev"][i])
        self.conv_ev1.wgrad(g_e1, c["ev_in"])
        finish_wgrads(self.all_ops)
        WGRAD_STREAM.join(self.device)
        if grad_sync is not None:
            grad_sync("early")
            grad_sync("late")
        self.ctx = None
