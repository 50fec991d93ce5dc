"""Timestep scheduler of the MI355X-native REFID hot path.

Drives the HIP kernels (through ``refid_amd.ops`` -> librefid_hip.so) for the forward pass
of ``FinalBidirectionAttenfusion`` and for a hand-written BPTT backward pass; torch is used
for device memory, streams and the autograd hand-off only -- there is no torch arithmetic
and no CPU/eager fallback in this file.

Reference behaviour followed (paths relative to /root/reference/basicsr/models/archs):
  XXNet_final_attenfusion_arch.py:130-218  forward orchestration, incl. the list-aliasing
      of backward states (:167,180-181): every forward step fuses the FINAL backward state;
  recurrent_sub_modules.py:41-49, 74-84, 270-296, 386-408, 488-503, 659-678, 719-726, 755-758;
  fusion_modules.py:97-134, 290-333 (EGACA).  Backward: SURVEY.md Appendix A.2.

Design notes (MI355X-first):
  * activations live in HBM as NHWC fp32; ``torch.cat`` is never materialised (two-source
    conv tiles), activations/residual adds ride in the conv epilogues;
  * parameters and gradients live in two flat arenas (``ParamArena``); the nn.Module's
    parameters are views into them, so clip+AdamW and the DDP all-reduce are single
    launches over one buffer;
  * EGACA's image branch (LN -> 1x1 -> dw3x3 -> GELU) does not depend on t: it is computed
    once per sweep direction instead of 2T times, its gradient is accumulated over the sweep;
  * EGACA's beta/gamma are folded into conv3/conv5's packed weights (scaled branch + residual
    in the conv epilogue) and un-folded once after BPTT (``refid_fold_back``);
  * dead work is skipped: ``encoders_backward[2].down`` (its output is discarded by the
    reference, arch :179) and the T copies of backward states.
"""
import os
from collections import OrderedDict

import torch

from . import ops
from ._lib import RefidHipError


def param_shapes(img_chn, ev_chn=2, out_chn=3, base=32, num_residual_blocks=2, num_block=1, num_encoders=3):
    """State-dict inventory (names, shapes, registration order) of the reference network:
    XXNet_final_attenfusion_arch.py:90-128 + the sub-module ctors it calls."""
    b = base
    sh = OrderedDict()

    def conv(name, co, ci, k, bias=True):
        sh[name + ".weight"] = (co, ci, k, k)
        if bias:
            sh[name + ".bias"] = (co,)

    conv("head.conv2d", b, ev_chn, 5)
    enc_in = [b * 2 ** i for i in range(num_encoders)]          # arch:49-56
    enc_out = [b * 2 ** (i + 1) for i in range(num_encoders)]

    def evr(prefix, ci, co, fuse, atten):
        conv(prefix + ".conv.conv2d", co, ci, 3)
        if atten:
            a = prefix + ".atten_fuse"
            sh[a + ".beta"] = (1, ci, 1, 1)
            sh[a + ".gamma"] = (1, co, 1, 1)
            conv(a + ".conv1", ci, ci, 1)
            sh[a + ".conv2.weight"] = (ci, 1, 3, 3); sh[a + ".conv2.bias"] = (ci,)
            conv(a + ".conv1_e", ci, ci, 1)
            sh[a + ".conv2_e.weight"] = (ci, 1, 3, 3); sh[a + ".conv2_e.bias"] = (ci,)
            conv(a + ".conv3", ci, 2 * ci, 1)
            conv(a + ".se_1.1", ci // 2, ci, 1); conv(a + ".se_1.3", ci, ci // 2, 1)
            conv(a + ".se_2.1", ci // 2, ci, 1); conv(a + ".se_2.3", ci, ci // 2, 1)
            conv(a + ".conv4", 2 * ci, ci, 1)
            conv(a + ".conv5", co, 2 * ci, 1)
            conv(a + ".conv_y_side", co, ci, 1)
            for n in ("norm1", "norm1_e", "norm2"):
                sh[f"{a}.{n}.weight"] = (ci,); sh[f"{a}.{n}.bias"] = (ci,)
        t = prefix + ".recurrent_block.forward_trunk.main"
        conv(t + ".0", co, 2 * co, 3)
        for k in range(num_block):                      # make_layer(ResidualBlockNoBN, num_block): rsm:719-726,760-773
            conv(t + f".2.{k}.conv1", co, co, 3)
            conv(t + f".2.{k}.conv2", co, co, 3)
        if fuse:
            conv(prefix + ".fuse_two_dir.conv2d", co, 2 * co, 1)
        sh[prefix + ".down.weight"] = (co, co, 4, 4)

    for i in range(num_encoders):
        evr(f"encoders_backward.{i}", enc_in[i], enc_out[i], False, i == 1)
    for i in range(num_encoders):
        evr(f"encoders_forward.{i}", enc_in[i], enc_out[i], True, i == 1)
    conv("head_img.conv2d", b, img_chn, 5)
    for i in range(num_encoders):
        p = f"img_encoders.{i}"
        conv(p + ".identity", enc_out[i], enc_in[i], 1)
        conv(p + ".conv_1", enc_out[i], enc_in[i], 3)
        conv(p + ".conv_2", enc_out[i], enc_out[i], 3)
        sh[p + ".down.weight"] = (enc_out[i], enc_out[i], 4, 4)
    cmax = b * 2 ** num_encoders
    for i in range(num_residual_blocks):
        conv(f"resblocks.{i}.conv1", cmax, cmax, 3)
        conv(f"resblocks.{i}.conv2", cmax, cmax, 3)
    for j, ci in enumerate(reversed(enc_out)):
        p = f"decoders.{j}"
        sh[p + ".transposed_conv2d.weight"] = (ci, ci // 2, 2, 2)
        sh[p + ".transposed_conv2d.bias"] = (ci // 2,)
        t = p + ".forward_trunk.main"
        conv(t + ".0", ci // 2, ci, 3)
        conv(t + ".2.0.conv1", ci // 2, ci // 2, 3)      # (the decoders' trunks always hold ONE block: rsm:375-384 does not pass
        conv(t + ".2.0.conv2", ci // 2, ci // 2, 3)      #  num_block on)
    conv("pred.conv2d", out_chn, b, 3)
    return sh


class ParamArena:
    """Flat fp32 parameter / gradient arenas; every tensor starts on a 16-byte boundary."""

    def __init__(self, shapes, device):
        self.shapes = shapes
        self.offsets = OrderedDict()
        off = 0
        for k, s in shapes.items():
            n = 1
            for d in s:
                n *= d
            self.offsets[k] = (off, n)
            off += (n + 3) // 4 * 4
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.pack_epoch = 0                   # bumped by every batched repack: stamps ConvOp's lazily packed fallback layouts

    def p(self, k):
        o, n = self.offsets[k]
        return self.flat_p[o:o + n].view(self.shapes[k])

    def g(self, k):
        o, n = self.offsets[k]
        return self.flat_g[o:o + n].view(self.shapes[k])


def _pad4(c):
    return (c + 3) // 4 * 4


# 3x3 stride-1 convolutions (88 % of the FLOPs) run on the fused Winograd F(2x2,3x3) tile
# (csrc/conv_wino.hip) when they are wide enough for its 64-channel tile; REFID_WINOGRAD=0
# forces the direct implicit-GEMM tile everywhere.
USE_WINOGRAD = os.environ.get("REFID_WINOGRAD", "1") != "0"
USE_POINTWISE = os.environ.get("REFID_POINTWISE", "1") != "0"     # register-operand tile for 1x1 convs
# weight-gradient kernels on a side HIP stream (REFID_OVERLAP_WGRAD=0: everything on one stream)
# Default since the end of round 5: OFF -- the weight gradients run on the main stream, in 8-step (large batches) or 24-step
# (small batches) groups.  The side stream dates from rounds 1-2, when the dependent input-gradient chain left the chip idle
# between its kernels; with today's kernels a B=8 step is 100 % busy on one stream, and the second compute stream only
# costs: B=8 412.0-413.8 ms with it vs 406.1-406.9 without (three alternating runs on one box), B=1 108.9 vs 103.5 (there it
# is also the sixth stream on four hardware queues).  REFID_OVERLAP_WGRAD=1 turns it back on (tests/test_hip_streams.py keeps
# its write-after-read hazard check alive).
_OVL_ENV = os.environ.get("REFID_OVERLAP_WGRAD", "0")
OVERLAP_WGRAD = _OVL_ENV not in ("0", "auto")


def overlap_wgrad():
    return bool(OVERLAP_WGRAD)
# EGACA forward as 6 launches (LayerNorm prologues, squeeze-excite + scale inside conv3, GELU second output) instead of
# 12; REFID_EGACA_FUSED=0: one kernel per reference op
EGACA_FUSED = os.environ.get("REFID_EGACA_FUSED", "1") != "0"


# The forward pass as a WAVEFRONT over HIP streams: EvR level i of step t needs level i-1 of step t and its own state of the
# previous step; the bottleneck / decoders / pred of step t need step t's encoder outputs and their own states of the previous
# step.  So level 0 (t+2), level 1 (t+1), level 2 (t) and the decoders (t-1) are independent kernel chains: each gets its own
# stream, joined by one cross-stream dependency per level and step.  The ramp-up / tail of one chain's launches and the small
# grids of the deep levels overlap the other chains (B=1: 124 -> 112 ms/step; B=8: < 1 %).  BPTT stays on one stream + the
# weight-gradient side stream: the same wavefront over BPTT measured SLOWER (B=1 120 ms, B=8 544 ms: the weight-gradient
# stream then has to wait for every chain).  REFID_PIPELINE=0: one chain; 1: always; default "auto": only while the batch
# leaves the chip room (B H W <= PIPELINE_MAX_PIX: B <= 4 at 256 x 256).  Every stream beyond the device's four hardware
# queues shares a queue with another one: with the three chain streams next to the weight-gradient stream, the input
# prefetch stream and RCCL's stream, a B=8 step lost 2 % -- and 4.5 % (18 ms) with a process group up -- to copies and
# barrier packets queued in front of an unrelated stream's kernels (round 5: tools/diag_gradsync.py, DESIGN.md section 5;
# REFID_FORCE_GRADSYNC=1: 429.7 vs 428.6 ms plain once the chains are off, 462 vs 442 with them), while the chains
# themselves bought < 1 % at B=8 (B=1: 120.3 -> 112.5 ms, B=4: 248.3 -> 245.8).
_PIPE_ENV = os.environ.get("REFID_PIPELINE", "auto")
PIPELINE = "auto" if _PIPE_ENV == "auto" else _PIPE_ENV != "0"
PIPELINE_MAX_PIX = int(os.environ.get("REFID_PIPELINE_MAX_PIX", str(4 * 256 * 256)))


def use_pipeline(b, h, w):
    return PIPELINE if isinstance(PIPELINE, bool) else b * h * w <= PIPELINE_MAX_PIX
PACK_BATCH = os.environ.get("REFID_PACK_BATCH", "1") != "0"       # all weight packings of a step in one launch
# fp32 Winograd packings of convs that run on their Winograd x six planes: packed on demand instead of every step (ConvOp)
LAZY_FALLBACK_PACKS = os.environ.get("REFID_LAZY_PACKS", "1") != "0"
CONVT_PW = os.environ.get("REFID_CONVT_PW", "1") != "0"           # ConvTranspose2d forward on the pointwise tile
CONVT_PWS = os.environ.get("REFID_CONVT_PWS", "1") != "0"         # ... its weight gradient as a streaming patch GEMM (algo 8)
# LayerNorm / depthwise / bias gradient sums of a BPTT half queued and issued grouped by destination (ops.rows_sum_defer)
ROWS_DEFER = os.environ.get("REFID_ROWS_DEFER", "0") != "0"
# the element-wise slab-reduction stages at the end of BPTT as one launch per kernel family (finish_wgrads)
FINISH_BATCH = os.environ.get("REFID_FINISH_BATCH", "1") != "0"
# The skip sums the reference forms right after a conv (b0 = e + x_blocks[2], decoder inputs z + e_blocks[.], arch:16-17,
# 199-203,211) and their BPTT counterparts (g_di + g_hd, g_b0 + g_skip) leave with the PRODUCING tile as a second output
# (refid_conv_desc.out2 = out + add2) instead of a separate add kernel each.  0: one add kernel per sum.
FUSE_SUMS = os.environ.get("REFID_FUSE_SUMS", "1") != "0"
# Linearity of the convolutions (round 4).  Three input sums of the reference have a term that does not depend on the time
# step:  C3(a_t + x_blocks[1])  (level-2 first conv, rsm:278-281),  C1([s_t | S_b])  (fuse_two_dir, rsm:291-293; S_b is the
# FINAL backward state for every t -- arch:181) and  pred(z_t + head)  (arch:215).  W(a_t + c) + bias = W a_t + (W c + bias):
# the second term is computed ONCE per sweep and rides in as the per-step conv's residual, so the sum tensor is never built
# (no add kernel) and the per-step fuse conv reads half its K.  In BPTT the gradient of the constant operand is
# W^T (sum_t g_t) and its weight-gradient share is (sum_t g_t) (x) c: one input-gradient and one weight-gradient launch on
# the summed gradient (ops.sum_n: one pass over the kept per-step tensors) instead of one accumulation per step.  The same
# deferred sum replaces the per-step `+=` into the image branch's gradients.  Results differ from the one-GEMM form only in
# summation order (parity tests unchanged).  REFID_LINEAR_SPLIT=0: the reference's op sequence.
LINEAR_SPLIT = os.environ.get("REFID_LINEAR_SPLIT", "1") != "0"


class _SideStreams:
    def __init__(self):
        self._s = {}
        self.pending = []                    # deferred weight-gradient launches: (op, g, a, b)

    def get(self, device):
        if device.type != "cuda":
            return None
        s = self._s.get(device.index)
        if s is None:
            s = self._s[device.index] = torch.cuda.Stream(device=device)
        return s

    def join(self, device):
        """Make the current stream wait for everything issued on the side stream."""
        flush_wgrads(device)
        s = self._s.get(device.index)
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)


WGRAD_STREAM = _SideStreams()
DEC_STREAM = _SideStreams()                  # the decoder chain of the forward-sweep pipeline (PIPELINE)
LV_STREAMS = (_SideStreams(), _SideStreams(), _SideStreams())   # EvR levels 1, 2 (and 3: num_encoders = 4); level 0 runs on the caller's stream
WGRAD_BATCH = int(os.environ.get("REFID_WGRAD_BATCH", "8"))     # deferred launches per cross-stream dependency
# The weight gradients of up to this many consecutive time steps of one conv are ONE launch (the weights are shared over
# T: their partial-sum slabs -- 134-537 MB of read-modify-write per launch at B=8 -- are then touched once per group
# instead of once per step; 3x3 and 4x4/stride-2 convs; 1 = off)
# Split-bf16 direct 3x3 tile (refid_conv2d algo 4) in the fp32 modes: 0 = never (Winograd fp32 tile), 6 / 3 = that many
# bf16 products per fp32 product.  compute_dtype "bf16x3" sets 3; "bf16" uses the tile with one product wherever it
# beats the LDS-staged bf16 tile (single-source convs and every input gradient).  6 is an experiment switch: measured
# equal to the Winograd tile at best (the bf16 matrix pipe is power limited with real data: DESIGN.md).
MFMA_SPLIT = int(os.environ.get("REFID_MFMA_SPLIT", "0"))
# Winograd-domain GEMMs of the 3x3 convs with more than 32 output channels on the bf16 matrix cores, six exact-split
# bf16 products per fp32 product (refid_conv2d algo 5, csrc/conv_wino6.hip): same error class as the fp32 Winograd tile at
# 2.67x fewer matrix-pipe cycles.  0 = fp32 Winograd tile everywhere.
WINO6 = os.environ.get("REFID_WINO6", "1") != "0"
# ... and, round 6, as THREE fp16 products on two-plane operands (refid_conv_desc.mfma_terms = 3): half the MFMAs and two
# thirds of the U bytes of the six-bf16-product form for a per-product error of ~2^-22 instead of ~2^-24 (still below the fp32
# accumulation's own error; fp16's range is handled by exact power-of-two scales, csrc/conv_wino6.hip).  REFID_WINO_F16=0: the
# six-product form (A/B switch).
_WF16 = os.environ.get("REFID_WINO_F16", "1")
WINO_F16_FWD = _WF16 in ("1", "fwd")                  # ("fwd" / "dgrad": only the forward convs / only the input gradients)
WINO_F16_DGRAD = _WF16 in ("1", "dgrad")
WINO6_MIN_CO = int(os.environ.get("REFID_WINO6_MIN_CO", "32"))
WINO6_THIN = os.environ.get("REFID_WINO6_THIN", "1") != "0"        # pred's forward (32 -> 3) on the 32-channel Winograd x six form
# conv_down (4x4 / stride 2) and its input gradient have no Winograd form; on the split tile with six bf16 products per
# fp32 product they run 1.6-2x faster than on the fp32 MFMA tile at the same distance from the float64 result (the operand
# split is exact: tests/test_hip_conv.py::test_split_tile_conv_down_*).  0 = keep them on the fp32 MFMA tile.
# (round 6: 19 = THREE fp16 products on two-plane operands scaled by exact powers of two -- the weights per tensor, the
#  activations per workgroup and online along K, csrc/conv_split.hip --: half of 6's MFMAs in the same error class; the default.
#  6 = six bf16 products on exact three-plane operands, round 2-5's form.)
DOWN_SPLIT = int(os.environ.get("REFID_DOWN_SPLIT", "19"))
# 1x1 convolutions on the pointwise tile's six-product form (refid_conv2d algo 3, mfma_terms 6).  Measured
# (tools/bench_pw6.py, profiles/r03_pw6_bench.txt): 1.03-1.26x the fp32-MFMA form when the launch is repeated on warm
# operands, NO gain inside the train step (676 launches: 39.8 vs 39.2 ms; the squeeze-excite fused conv3 is slower, 116
# vs 90 us) -- on cold operands these tiles wait for HBM, not for the matrix pipe, and the six-product form runs 3 instead of
# 4 waves per SIMD (144 registers).  Off; the form stays reachable through the C ABI and is tested.
# six bf16 products on the pointwise tile: "1" every 1x1 layer (round 3: flat in the step), "2" only where the reduction runs over
# 128 channels or more (those layers are bound by the fp32 matrix pipe), "0" never
PW6 = int(os.environ.get("REFID_PW6", "0") or 0)
PW6_MIN_K = 128
# smallest output-channel count whose 3x3 weight gradient goes to the Winograd tile (64 x 32 channel tiles)
WGRAD_WINO_MIN_CO = int(os.environ.get("REFID_WGRAD_WINO_MIN_CO", "32"))
# Time steps per weight-gradient launch (the ABI takes up to 24): all T steps of a sweep in ONE launch write the partial-sum slabs
# once instead of read-modify-writing them per group.  Round 4 (weight gradients on a side stream): 8, because a later start of
# the weight-gradient kernels cost more overlap than the slab passes saved (B=8 460.5 vs 458.3 ms at 24 vs 8).  Round 5: the
# weight gradients run on the MAIN stream at every batch size, a launch is a link of the one chain, and 24 wins everywhere:
# B=8 400.2 / 399.9 vs 404.9 / 403.5 ms (alternating, one box; 12 steps: 405.0 / 402.8), B=1 99.6 -> 98.0 ms.
_WG_ENV = os.environ.get("REFID_WGRAD_GROUP")
WGRAD_GROUP = max(1, min(24, int(_WG_ENV))) if _WG_ENV else (8 if OVERLAP_WGRAD else 24)
WGRAD_GROUP_SMALL = max(1, min(24, int(_WG_ENV))) if _WG_ENV else 24
# What a waiting weight-gradient call keeps alive: the (gradient, input) tensors of its step, which BPTT would otherwise have
# released -- measured 128 GB (all 23 steps per launch) against 90 GB (8 per launch) at B=8, 256 x 256: ~4.9 KB per pixel and
# time step.  Engine._set_wgrad_groups shrinks the group when that would not fit into half of the HBM still free when BPTT
# starts (a larger crop / T / batch then runs in smaller groups instead of running out of memory).
WGRAD_KEEP_BYTES_PER_PIXEL_STEP = 4900
# Round 5: 3x3 weight gradients on the Winograd 2x4-tile form (refid_wgrad_desc.algo = 5: F(3,2) x F(3,4), 24 instead of 32
# fp32 MFMAs per 8 output pixels, packed transforms) wherever the 2x2-tile form (algo 1) was used and the output has at
# least WGRAD_F4_MIN_HW rows and columns (below that a 4 x 16-pixel K tile is mostly zero padding and the 2x2 form's smaller
# transform error is free).  REFID_WGRAD_F4=0 is the A/B switch back to algo 1.
# 1x1 weight gradients on the streaming form (csrc/wgrad_pws.hip: LDS-DMA ring, every tensor read once at 128 x 128 channels),
# and grouped over the time steps like the 3x3 ones (the older 1x1 tiles cannot group).  REFID_PWS_WGRAD=0: the older tiles.
PWS_WGRAD = os.environ.get("REFID_PWS_WGRAD", "1") != "0"
WGRAD_F4 = os.environ.get("REFID_WGRAD_F4", "1") != "0"
WGRAD_F4_MIN_HW = int(os.environ.get("REFID_WGRAD_F4_MIN_HW", "16"))
# conv_down's weight gradient (4x4 stride 2) on the same kernel through the input's four parity phases (algo 7: 12 instead of
# 16 fp32 MFMA-units per output pixel).  REFID_WGRAD_DOWN_F4=0: the direct tile.
WGRAD_DOWN_F4 = os.environ.get("REFID_WGRAD_DOWN_F4", "1") != "0"


def wgrad_group_cap(n, pixels, device, free_bytes=None):
    """Largest group size <= n whose waiting operands (WGRAD_KEEP_BYTES_PER_PIXEL_STEP per pixel and step) fit into half of
    the free HBM: the device's free memory plus what torch's caching allocator holds unallocated."""
    if n <= 1 or pixels <= 0:
        return n
    if free_bytes is None:
        if torch.device(device).type != "cuda":
            return n
        free_bytes = torch.cuda.mem_get_info(device)[0] + (torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    per_step = pixels * WGRAD_KEEP_BYTES_PER_PIXEL_STEP
    return max(1, min(n, int(free_bytes // 2 // per_step)))


def flush_wgrads(device):
    """Launch the deferred weight-gradient kernels on the side stream, after everything enqueued so far on the
    current stream (their operands' producers)."""
    pend = WGRAD_STREAM.pending
    if not pend:
        return
    side = WGRAD_STREAM.get(device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for op, g, a, b, bias, i_base in pend:
            op._wgrad(g, a, b, bias, i_base)
    pend.clear()


FILL_MIN_WG = int(os.environ.get("REFID_FILL_MIN_WG", "256"))     # workgroups from which conv_down takes the split tile


def _fills_gpu(n, ho, wo, cout, classes):
    """Does the split tile's smallest grid (4 x 32 pixel tiles x 64 channels) give each of the 256 CUs a workgroup?
    Same policy switch as the split-K of the other tiles (refid_conv_desc.split_k): decided by the total grid under
    'auto', by the per-sample geometry (as if 8 samples) otherwise, so that a sample's bits do not depend on the batch."""
    if ops.WINO_SPLIT != 2:
        n = 8
    return n * -(-ho // 4) * -(-wo // 32) * -(-cout // 64) * classes >= FILL_MIN_WG


_LIM4 = (2 ** 31 - 1) // 4                    # fp32 elements below 2 GiB


def _batch_step(n, *tensors):
    """Largest sub-batch whose tensors all stay below 2 GiB."""
    per = max(t.stride(0) * 4 for t in tensors if t is not None)
    if per >= 2 ** 31 - 1:
        raise RefidHipError(f"conv: one sample's tensor has {per} bytes; the tiles' 32-bit offsets end at 2 GiB (tile the frame: "
                            "refid_amd.tiling)")
    return max(1, (2 ** 31 - 2) // per)


class ConvOp:
    """One convolution of the network: geometry + packed weights + the three kernels."""
    default_split = 0                         # set by Engine.__init__ for the convs it builds (compute_dtype "bf16x3")

    def __init__(self, arena, name, kind="conv", need_dgrad=True, scale_name=None, bf16=False, split=0):
        self.arena, self.name, self.kind = arena, name, kind
        self.bf16 = bf16
        self.split = 0                        # product terms of the split-bf16 tile (algo 4) this conv may use
        self.s_f16 = False                    # ... as three fp16 products (terms 19: the packings hold two fp16 planes)
        self.w = arena.p(name + ".weight")
        self.gw = arena.g(name + ".weight")
        self.has_bias = (name + ".bias") in arena.shapes
        self.b = arena.p(name + ".bias") if self.has_bias else None
        self.gb = arena.g(name + ".bias") if self.has_bias else None
        self.scale = arena.p(scale_name).view(-1) if scale_name else None
        self.scale_name = scale_name
        # a conv whose packed weights carry a folded per-row scale (EGACA's beta / gamma) accumulates the gradient of
        # the FOLDED weights; it must never be mixed with what the arena already holds (gradient accumulation over
        # several backward calls), so it goes to a private buffer that Engine.backward zeroes and refid_fold_back
        # un-folds into the arena (Engine._bind_fold_scratch)
        self.gw_arena, self.gb_arena = self.gw, self.gb
        s = self.w.shape
        if kind == "convT":
            self.ci, self.co, self.k = s[0], s[1], 2
        else:
            self.co, self.ci, self.k = s[0], s[1], s[2]
        self.need_dgrad = need_dgrad
        k = self.k
        if kind == "conv":
            self.stride, self.pad, self.mode = 1, k // 2, 0
            self.f_geo = (k, k, 1, 0)
            self.f_role, self.f_rows = ops.ROLE_FWD, self.co
            self.d_geo = (k, k, 1, 0)
            self.d_role, self.d_rows = ops.ROLE_DGRAD, self.ci
        elif kind == "down":
            self.stride, self.pad, self.mode = 2, 1, 0
            self.f_geo = (4, 4, 2, 0)
            self.f_role, self.f_rows = ops.ROLE_FWD, self.co
            self.d_geo = (4, 4, 2, 2)
            self.d_role, self.d_rows = ops.ROLE_DOWN_DGRAD, self.ci
        elif kind == "convT":
            self.stride, self.pad, self.mode = 1, 0, 1
            self.f_geo = (1, 1, 1, 1)
            self.f_role, self.f_rows = ops.ROLE_CONVT, 4 * self.co
            self.d_geo = (2, 2, 2, 0)
            self.d_role, self.d_rows = ops.ROLE_CONVT_DGRAD, self.ci
        else:
            raise ValueError(kind)
        kh, kw, st, md = self.f_geo
        self.f_kc = ops.conv_kc(kh, kw, st, md)
        self.f_bn = ops.conv_bn(kh, kw, st, md, self.f_rows)
        self.f_algo = self.d_algo = 0
        pointwise = USE_POINTWISE and kind == "conv" and k == 1 and self.ci % 16 == 0 and self.co % 16 == 0
        if bf16 and pointwise:
            # the 1x1 layers are bandwidth bound on fp32 tensors either way: the register-operand fp32 tile (no LDS,
            # no conversion pass) beats the LDS-staged bf16 tile on them (39.5 -> 34 ms / step), at full precision
            bf16 = self.bf16 = False
        if bf16:
            # bf16 MFMA operands on the direct tile (fp32 accumulate / epilogue / tensors); no Winograd:
            # its transforms would amplify the operand rounding, and bf16 MFMA is 16x the fp32 rate anyway
            self.f_algo = self.d_algo = 2
            self.f_kc *= 2
        elif USE_WINOGRAD and kind == "conv" and k == 3:
            if self.co >= 16:                  # pred (3 channels) stays on the direct tile
                self.f_algo, self.f_role, self.f_kc, self.f_bn = 1, ops.ROLE_WINO_FWD, 8, 64
            self.d_algo, self.d_role = 1, ops.ROLE_WINO_DGRAD
        elif pointwise:
            self.f_algo = self.d_algo = 3
            self.f_kc, self.f_bn = 8, 32
        elif kind == "convT" and USE_POINTWISE and CONVT_PW and self.ci % 16 == 0 and self.co % 4 == 0 and 4 * self.co > 32:
            # ConvTranspose2d(2,2) forward = a 1x1 GEMM over 4 Co columns: the register-operand pointwise tile with a
            # pixel-shuffle store (the LDS-staged direct tile ran it at 0.24 of the fp32 pipe with 0.78 LDS bank conflicts)
            self.f_algo = 3
            self.f_kc, self.f_bn = 8, 32
        self.f_pad = -(-self.f_rows // self.f_bn) * self.f_bn
        dev = self.w.device
        pdt = torch.bfloat16 if bf16 else torch.float32
        self.wp = torch.empty(ops.packed_weight_floats(self.f_role, self.f_bn, self.f_kc, k, k, self.co, self.ci),
                              dtype=pdt, device=dev)
        self.wd = None
        self.d_bn_cache = {}
        if need_dgrad:
            kh, kw, st, md = self.d_geo
            self.d_kc = ops.conv_kc(kh, kw, st, md)
            # a dgrad may be issued for a row range (two-source convs): tile width follows the range
            self.d_bn = ops.conv_bn(kh, kw, st, md, self.d_rows if self.d_rows <= 128 else 128)
            if kind == "conv" and self.ci == 2 * self.co and self.k in (1, 3):
                self.d_bn = ops.conv_bn(kh, kw, st, md, self.co)     # issued as two halves of co rows
            if kind == "convT" and self.f_algo == 3 and (2 * self.co) % 8 == 0:
                # its input gradient (2x2 stride 2: non-overlapping patches) as ONE GEMM with K = 4 Co on the pointwise tile: a
                # pixel's patch is two contiguous runs of 2 Co floats (the direct tile fetched 3.3-4.4x its bytes, 0.29 of HBM)
                self.d_algo, self.d_role = 3, ops.ROLE_CONVT_DGRAD_PW
            if self.d_algo == 1:
                self.d_kc, self.d_bn = 8, 64
            if self.d_algo == 3:
                self.d_kc, self.d_bn = 8, 32
            if bf16:
                self.d_kc *= 2
            self.d_pad = -(-self.d_rows // self.d_bn) * self.d_bn
            self.wd = torch.empty(ops.packed_weight_floats(self.d_role, self.d_bn, self.d_kc, k, k, self.co, self.ci),
                                  dtype=pdt, device=dev)
        # pointwise tile, six bf16 products per fp32 product: second packing (three bf16 planes, 16-channel chunks)
        self.wpp6 = self.wdp6 = None
        if PW6 and self.kind == "conv" and self.f_algo == 3 and self.ci % 16 == 0 and self.co % 16 == 0:
            if self.co > 32 and (PW6 == 1 or self.ci >= PW6_MIN_K):
                self.wpp6 = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_FWD, 32, 1, 1, self.co, self.ci, 3) // 2,
                                        dtype=torch.bfloat16, device=dev)
            if need_dgrad and self.ci > 32 and (PW6 == 1 or self.co >= PW6_MIN_K):
                self.wdp6 = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_DGRAD, 32, 1, 1, self.co, self.ci, 3) // 2,
                                        dtype=torch.bfloat16, device=dev)
        # Winograd x six bf16 products (algo 5): third packing -- three bf16 planes of U = G g G^T.  From 32 output channels
        # on (round 4: the tile's 32-channel form; REFID_WINO6_MIN_CO=33 restores round 3's choice for an A/B)
        self.wp6 = self.wd6 = None
        # ... and thin outputs (pred, 32 -> 3): the direct fp32 tile pads them to 32 GEMM columns and is bound by the fp32
        # matrix pipe (100 us per launch at B=8); on the 32-channel Winograd x six form the padding costs cheap bf16 MFMAs
        thin6 = WINO6_THIN and kind == "conv" and k == 3 and self.co <= 4 and self.ci % 16 == 0 and USE_WINOGRAD
        # the packings' form: two fp16 planes + header (mfma_terms 3) or three bf16 planes (0)
        self.w6_f16_f, self.w6_f16_d = WINO_F16_FWD, WINO_F16_DGRAD
        if WINO6 and not bf16 and ((self.f_algo == 1 and self.co >= WINO6_MIN_CO) or thin6) and self.ci % 4 == 0 and \
                ops.packed_weight_wino6_bytes(ops.ROLE_WINO_FWD, self.co, self.ci) < 2 ** 31 - 1:
            self.wp6 = torch.empty(ops.packed_weight_wino6_bytes(ops.ROLE_WINO_FWD, self.co, self.ci, self.w6_f16_f) // 2,
                                   dtype=torch.bfloat16, device=dev)
        if WINO6 and not bf16 and need_dgrad and self.d_algo == 1 and self.ci >= WINO6_MIN_CO and self.co % 4 == 0 and \
                ops.packed_weight_wino6_bytes(ops.ROLE_WINO_DGRAD, self.co, self.ci) < 2 ** 31 - 1:
            self.wd6 = torch.empty(ops.packed_weight_wino6_bytes(ops.ROLE_WINO_DGRAD, self.co, self.ci, self.w6_f16_d) // 2,
                                   dtype=torch.bfloat16, device=dev)
        # split-bf16 direct tile (algo 4): second packing next to the default one
        terms = 1 if bf16 else (split or ConvOp.default_split or MFMA_SPLIT)
        self.wps = self.wds = None
        if terms and kind == "conv" and k == 3 and self.ci % 8 == 0 and self.co >= 16:
            self.split = terms
            planes = {1: 1, 3: 2, 6: 3}[terms]
            self.s_planes = planes
            self.sf_bn = ops.conv_bn(3, 3, 1, 0, self.co)
            self.sf_pad = -(-self.co // self.sf_bn) * self.sf_bn
            self.wps = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_FWD, self.sf_bn, 3, 3, self.co, self.ci, planes) // 2,
                                   dtype=torch.bfloat16, device=dev)
            if need_dgrad and self.co % 8 == 0:
                self.sd_bn = self.d_bn if self.d_algo != 1 else ops.conv_bn(3, 3, 1, 0, min(self.ci, 128))
                if self.ci == 2 * self.co:
                    self.sd_bn = ops.conv_bn(3, 3, 1, 0, self.co)
                self.sd_pad = -(-self.ci // self.sd_bn) * self.sd_bn
                self.wds = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_DGRAD, self.sd_bn, 3, 3, self.co, self.ci, planes) // 2,
                                       dtype=torch.bfloat16, device=dev)
        if kind == "down" and self.ci % 8 == 0 and self.co % 8 == 0 and (bf16 or DOWN_SPLIT or ConvOp.default_split):
            self.split = terms = 1 if bf16 else (ConvOp.default_split or DOWN_SPLIT)
            self.s_planes = planes = {1: 1, 3: 2, 6: 3, ops.TERMS_F16X3: 2}[terms]
            self.s_f16 = f16 = terms == ops.TERMS_F16X3
            if not bf16:                      # (plain bf16 operands: the LDS-staged tile is as fast on the forward conv)
                self.sf_bn = ops.conv_bn(4, 4, 2, 0, self.co)
                self.sf_pad = -(-self.co // self.sf_bn) * self.sf_bn
                self.wps = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_FWD, self.sf_bn, 4, 4, self.co, self.ci, planes, f16) // 2,
                                       dtype=torch.bfloat16, device=dev)
            if need_dgrad:
                self.sd_bn = ops.conv_bn(4, 4, 2, 2, self.ci)
                self.sd_pad = -(-self.ci // self.sd_bn) * self.sd_bn
                self.wds = torch.empty(ops.packed_weight_split_bytes(ops.ROLE_DOWN_DGRAD, self.sd_bn, 4, 4, self.co, self.ci,
                                                                     planes, f16) // 2, dtype=torch.bfloat16, device=dev)
        self.b_eff = self.b
        if self.scale is not None and self.has_bias:
            self.b_eff = torch.empty_like(self.b)
        # A conv with Winograd x six planes runs on them wherever its shapes allow (fwd / dgrad below): its fp32 Winograd
        # packing (wp / wd, algo 1) is a FALLBACK layout (two sources whose first is not a multiple of 16 channels, input-
        # gradient row ranges below 32 rows) -- 159 MB of the model's 463 MB of packed weights that the batched repack of
        # every optimiser step does not write; whoever needs one packs it on demand (_fallback_wp / _fallback_wd), stamped
        # with the arena's pack epoch.  The one-by-one repack() writes everything.
        self.wp_lazy = LAZY_FALLBACK_PACKS and self.wp6 is not None and self.f_algo == 1
        self.wd_lazy = LAZY_FALLBACK_PACKS and self.wd6 is not None and self.d_algo == 1
        self._wp_epoch = self._wd_epoch = -1
        # weight-gradient partial sums of the T recurrent steps accumulate in a private slab buffer
        # and are reduced into the parameter gradient once per step (finish_wgrad)
        self.wslab = None
        self.w_calls = 0
        self.w_last = None
        self.w_pend = []                      # weight-gradient calls waiting for their group
        self.w_algo, self.w_bias = 0, True
        # time steps per weight-gradient launch: 1 = launch immediately (every op by default: an op called once per
        # backward -- the image branch, every EvhinetEngine op -- must not wait for finish_wgrad); Engine sets
        # min(WGRAD_GROUP, T) on the convs the T recurrent steps share (Engine._set_wgrad_groups)
        self.w_group = 1

    def plan_repack(self, plan):
        """The same packings as repack(), as entries of an ops.PackPlan (one launch for the whole model)."""
        k = self.k
        if not self.wp_lazy:
            plan.add_pack(self.w, self.f_role, self.f_bn, self.f_kc, k, k, self.co, self.ci, self.wp, oscale=self.scale, bf16=self.bf16)
        if self.wd is not None and not self.wd_lazy:
            plan.add_pack(self.w, self.d_role, self.d_bn, self.d_kc, k, k, self.co, self.ci, self.wd, oscale=self.scale, bf16=self.bf16)
        if self.wpp6 is not None:
            plan.add_split(self.w, ops.ROLE_FWD, 32, 1, 1, self.co, self.ci, 3, self.wpp6, oscale=self.scale)
        if self.wdp6 is not None:
            plan.add_split(self.w, ops.ROLE_DGRAD, 32, 1, 1, self.co, self.ci, 3, self.wdp6, oscale=self.scale)
        if self.wp6 is not None:
            plan.add_wino6(self.w, ops.ROLE_WINO_FWD, self.co, self.ci, self.wp6, oscale=self.scale, f16=self.w6_f16_f)
        if self.wd6 is not None:
            plan.add_wino6(self.w, ops.ROLE_WINO_DGRAD, self.co, self.ci, self.wd6, oscale=self.scale, f16=self.w6_f16_d)
        if self.wps is not None:
            plan.add_split(self.w, ops.ROLE_FWD, self.sf_bn, k, k, self.co, self.ci, self.s_planes, self.wps, oscale=self.scale, f16=self.s_f16)
        if self.wds is not None:
            plan.add_split(self.w, ops.ROLE_DOWN_DGRAD if self.kind == "down" else ops.ROLE_DGRAD, self.sd_bn, k, k, self.co,
                           self.ci, self.s_planes, self.wds, oscale=self.scale, f16=self.s_f16)
        if self.scale is not None and self.has_bias:
            plan.add_mul_vec(self.b, self.scale, self.b_eff)

    def _fallback_wp(self):
        """The fp32 Winograd forward packing, written now if the batched repack left it out (wp_lazy)."""
        if self.wp_lazy and self._wp_epoch != self.arena.pack_epoch:
            ops.pack_conv_weights(self.w, self.f_role, self.f_bn, self.f_kc, self.k, self.k, self.co, self.ci, out=self.wp,
                                  oscale=self.scale)
            self._wp_epoch = self.arena.pack_epoch
        return self.wp

    def _fallback_wd(self):
        if self.wd_lazy and self._wd_epoch != self.arena.pack_epoch:
            ops.pack_conv_weights(self.w, self.d_role, self.d_bn, self.d_kc, self.k, self.k, self.co, self.ci, out=self.wd,
                                  oscale=self.scale)
            self._wd_epoch = self.arena.pack_epoch
        return self.wd

    def pack_fallbacks(self):
        """Bring the lazily packed layouts up to date (tests that read wp / wd directly)."""
        self._fallback_wp()
        if self.wd is not None:
            self._fallback_wd()

    def repack(self):
        k = self.k
        pack = ops.pack_conv_weights_bf16 if self.bf16 else ops.pack_conv_weights
        pack(self.w, self.f_role, self.f_bn, self.f_kc, k, k, self.co, self.ci, out=self.wp, oscale=self.scale)
        if self.wd is not None:
            pack(self.w, self.d_role, self.d_bn, self.d_kc, k, k, self.co, self.ci, out=self.wd, oscale=self.scale)
        self._wp_epoch = self._wd_epoch = self.arena.pack_epoch
        if self.wpp6 is not None:
            ops.pack_conv_weights_split(self.w, ops.ROLE_FWD, 32, 1, 1, self.co, self.ci, planes=3, out=self.wpp6, oscale=self.scale)
        if self.wdp6 is not None:
            ops.pack_conv_weights_split(self.w, ops.ROLE_DGRAD, 32, 1, 1, self.co, self.ci, planes=3, out=self.wdp6, oscale=self.scale)
        if self.wp6 is not None:
            ops.pack_conv_weights_wino6(self.w, ops.ROLE_WINO_FWD, self.co, self.ci, out=self.wp6, oscale=self.scale, f16=self.w6_f16_f)
        if self.wd6 is not None:
            ops.pack_conv_weights_wino6(self.w, ops.ROLE_WINO_DGRAD, self.co, self.ci, out=self.wd6, oscale=self.scale, f16=self.w6_f16_d)
        if self.wps is not None:
            ops.pack_conv_weights_split(self.w, ops.ROLE_FWD, self.sf_bn, k, k, self.co, self.ci, planes=self.s_planes,
                                        out=self.wps, oscale=self.scale, f16=self.s_f16)
        if self.wds is not None:
            ops.pack_conv_weights_split(self.w, ops.ROLE_DOWN_DGRAD if self.kind == "down" else ops.ROLE_DGRAD, self.sd_bn,
                                        k, k, self.co, self.ci, planes=self.s_planes, out=self.wds, oscale=self.scale, f16=self.s_f16)
        if self.scale is not None and self.has_bias:
            ops.mul_vec(self.b, self.scale, out=self.b_eff)

    # ---- forward ---------------------------------------------------------------------------
    def fwd(self, a, b=None, res=None, slope_pre=1.0, slope_post=1.0, out=None, pw=None, bias=True, plus=None):
        """out = post(pre(conv([a|b]) + bias) + res).  bias=False leaves the bias out (a conv is linear: where one operand of
        an input sum does not depend on the time step, W (a_t + c) + bias is issued as W a_t + (W c + bias) with the second
        term computed once and passed as `res` -- Engine.forward).  plus: a tensor of the output's shape; returns
        (out, out + plus) -- the skip sum the next layer reads leaves with this tile (refid_conv_desc.out2)."""
        n, h, w, _ = a.shape
        bv = self.b_eff if bias else None
        if self.kind == "conv":
            ho, wo, oc = h, w, self.co
        elif self.kind == "down":
            ho, wo, oc = h // 2, w // 2, self.co
        else:
            ho, wo, oc = 2 * h, 2 * w, self.co
        if out is None:
            out = torch.empty((n, ho, wo, _pad4(oc)), dtype=torch.float32, device=a.device)
            if _pad4(oc) != oc:
                out.zero_()
                out = out[..., :oc]
        o2 = torch.empty_like(out) if plus is not None else None
        two = dict(add2=plus, out2=o2) if plus is not None else {}
        if n * a.stride(0) >= _LIM4 or n * out.stride(0) >= _LIM4 or (b is not None and n * b.stride(0) >= _LIM4):
            step = _batch_step(n, a, b, res, out)
            # the tiles address their tensors with 32-bit byte offsets (hardware range-checked buffer loads): a batch whose
            # tensors reach 2 GiB is issued in sub-batches -- samples are independent (no cross-sample op on the path)
            if pw is not None:
                raise RefidHipError(f"{self.name}: fused pointwise extras cannot be issued in sub-batches (tensor >= 2 GiB)")
            for i in range(0, n, step):
                j = min(n, i + step)
                self.fwd(a[i:j], None if b is None else b[i:j], None if res is None else res[i:j], slope_pre, slope_post,
                         out[i:j], bias=bias)
            return out if plus is None else (out, ops.add(out, plus, out=o2))
        kh, kw, st, md = self.f_geo
        if self.wps is not None and (self.split > 1 or b is None) and a.shape[3] % 8 == 0 and pw is None and \
                (self.kind == "conv" or (b is None and _fills_gpu(n, ho, wo, self.co, 1))):
            # (one product = plain bf16 operands: only where this tile beats the LDS-staged one -- single-source convs;
            #  conv_down: only when the grid gives every CU a workgroup -- the fp32 MFMA tile has a split-K form for less)
            ops.conv2d(a, self.wps, out, kh=kh, kw=kw, stride=st, pad=1, mode=0, cout=self.co, cout_pad=self.sf_pad, in_b=b,
                       bias=bv, res=res, slope_pre=slope_pre, slope_post=slope_post, algo=4, terms=self.split, **two)
            return out if plus is None else (out, o2)
        if self.wp6 is not None and self.split == 0 and (b is None or a.shape[3] % 16 == 0):
            ops.conv2d(a, self.wp6, out, kh=3, kw=3, stride=1, pad=1, mode=0, cout=self.f_rows, cout_pad=-(-self.f_rows // 64) * 64, in_b=b,
                       bias=bv, res=res, slope_pre=slope_pre, slope_post=slope_post, algo=5, terms=3 if self.w6_f16_f else 0, **two)
            return out if plus is None else (out, o2)
        if self.wpp6 is not None and a.shape[3] % 16 == 0 and (b is None or b.shape[3] % 16 == 0):
            ops.conv2d(a, self.wpp6, out, kh=1, kw=1, stride=1, pad=0, mode=0, cout=self.f_rows, cout_pad=self.f_pad, in_b=b,
                       bias=bv, res=res, slope_pre=slope_pre, slope_post=slope_post, algo=3, pw=pw, terms=6)
            return out if plus is None else (out, ops.add(out, plus, out=o2))
        if self.f_algo == 3:                                      # (the pointwise tile has no second output)
            two = {}
        ops.conv2d(a, self._fallback_wp(), out, kh=kh, kw=kw, stride=st, pad=self.pad, mode=md, cout=self.f_rows,
                   cout_pad=self.f_pad, in_b=b, bias=bv, res=res, slope_pre=slope_pre, slope_post=slope_post,
                   algo=self.f_algo, pw=pw, **two)
        if plus is not None and not two:
            ops.add(out, plus, out=o2)
        return out if plus is None else (out, o2)

    def can_fwd_from(self):
        """Can fwd_from() address a block of input channels of this conv directly?  (fp32 pointwise tile, chunk-major packing)"""
        return self.kind == "conv" and self.k == 1 and self.f_algo == 3 and self.wpp6 is None and not self.bf16

    def fwd_from(self, a, k_base):
        """conv over the input channels k_base .. k_base + C_a only (+ bias): the packing is [chunk][row][kc], so the block's
        weights are the packed array from chunk k_base / kc on, and `a` is the single source (the time-independent half of
        fuse_two_dir, rsm:291-293, needs no stand-in zero tensor for the other half)."""
        if not self.can_fwd_from() or k_base % self.f_kc or a.shape[3] % self.f_kc:
            raise RefidHipError(f"{self.name}: fwd_from needs the fp32 pointwise tile and chunk-aligned channel blocks")
        n, h, w, _ = a.shape
        out = torch.empty((n, h, w, self.co), dtype=torch.float32, device=a.device)
        ops.conv2d(a, self.wp[(k_base // self.f_kc) * self.f_pad * self.f_kc:], out, kh=1, kw=1, stride=1, pad=0, mode=0,
                   cout=self.f_rows, cout_pad=self.f_pad, bias=self.b_eff, algo=3)
        return out

    # ---- input gradient ----------------------------------------------------------------------
    def dgrad(self, g, rows=None, res=None, mask=None, slope_mask=1.0, out=None, plus=None, gelu_mask=False):
        """Input gradient (+ res, masked).  plus: returns (out, out + plus) -- see fwd.  gelu_mask: multiply by GELU'(mask)
        instead of the leaky-step derivative (pointwise tile; elsewhere a separate gelu_bwd launch does it)."""
        if self.wd is None:
            raise RefidHipError(f"{self.name}: dgrad weights were not requested")
        base, cnt = rows if rows is not None else (0, self.d_rows)
        n, h, w, _ = g.shape
        if self.kind == "conv":
            ho, wo = h, w
        elif self.kind == "down":
            ho, wo = 2 * h, 2 * w
        else:
            ho, wo = h // 2, w // 2
        if out is None:
            out = torch.empty((n, ho, wo, cnt), dtype=torch.float32, device=g.device)
        o2 = torch.empty_like(out) if plus is not None else None
        two = dict(add2=plus, out2=o2) if plus is not None else {}
        if n * g.stride(0) >= _LIM4 or n * out.stride(0) >= _LIM4:      # (see fwd)
            step = _batch_step(n, g, res, mask, out)
            for i in range(0, n, step):
                j = min(n, i + step)
                self.dgrad(g[i:j], rows, None if res is None else res[i:j], None if mask is None else mask[i:j], slope_mask,
                           out[i:j], gelu_mask=gelu_mask)
            return out if plus is None else (out, ops.add(out, plus, out=o2))
        kh, kw, st, md = self.d_geo
        pad = self.pad if self.kind == "conv" else (1 if self.kind == "down" else 0)
        if self.kind == "conv":
            pad = self.k - 1 - self.pad
        # GELU' rides only in the fp32 pointwise tile (refid_conv_desc.mask_mode = 1); every other tile would silently apply
        # the leaky-step mask instead, so a request that cannot be honoured is an error, never a wrong gradient
        gelu_ok = self.kind == "conv" and self.d_algo == 3 and self.wds is None and not (self.wd6 is not None and self.split == 0 and cnt >= WINO6_MIN_CO) \
            and not (self.wdp6 is not None and cnt > 32 and g.shape[3] % 16 == 0)
        if gelu_mask and not gelu_ok:
            raise RefidHipError(f"{self.name}: gelu_mask needs the fp32 pointwise input-gradient tile (d_algo {self.d_algo}); "
                                f"apply ops.gelu_bwd separately")
        if self.wds is not None and (self.kind == "conv" or _fills_gpu(n, h, w, cnt, 4)):
            ops.conv2d(g, self.wds, out, kh=kh, kw=kw, stride=st, pad=1, mode=md, cout=cnt, cout_pad=self.sd_pad, co_base=base,
                       res=res, mask=mask, slope_mask=slope_mask, algo=4, terms=self.split, **two)
            return out if plus is None else (out, o2)
        if self.wd6 is not None and self.split == 0 and cnt >= WINO6_MIN_CO:
            ops.conv2d(g, self.wd6, out, kh=3, kw=3, stride=1, pad=1, mode=0, cout=cnt, cout_pad=self.d_pad, co_base=base,
                       res=res, mask=mask, slope_mask=slope_mask, algo=5, terms=3 if self.w6_f16_d else 0, **two)
            return out if plus is None else (out, o2)
        if self.wdp6 is not None and cnt > 32 and g.shape[3] % 16 == 0:
            ops.conv2d(g, self.wdp6, out, kh=1, kw=1, stride=1, pad=0, mode=0, cout=cnt, cout_pad=self.d_pad, co_base=base,
                       res=res, mask=mask, slope_mask=slope_mask, algo=3, terms=6)
            return out if plus is None else (out, ops.add(out, plus, out=o2))
        fused_two = bool(two)                                    # was the second output handed to the tile?
        if self.d_algo == 3 and self.kind == "convT":            # the patch GEMM: dense pixels; second output supported
            if g.stride(2) != g.shape[3]:
                g = g.contiguous()
        elif self.d_algo == 3:                                    # (the 1x1 layers keep their separate skip-sum launch)
            two, fused_two = {}, False
            if gelu_mask:
                two = dict(mask_mode=1)
        ops.conv2d(g, self._fallback_wd(), out, kh=kh, kw=kw, stride=st, pad=pad, mode=md, cout=cnt, cout_pad=self.d_pad,
                   co_base=base, res=res, mask=mask, slope_mask=slope_mask, algo=self.d_algo, **two)
        if plus is not None and not fused_two:
            ops.add(out, plus, out=o2)
        return out if plus is None else (out, o2)

    # ---- weight / bias gradient ----------------------------------------------------------------
    def wgrad(self, g, a, b=None, bias=True, i_base=0):
        """g: gradient w.r.t. this conv's (pre-epilogue) output; (a|b): its input sources.  bias=False: no bias-gradient
        contribution from this call (the extra call of a linearity split: the per-step calls already hold sum g).
        i_base > 0: `a` holds the input channels i_base .. i_base + C_a of the conv (the time-independent half of a
        linearity split): one-shot launch straight into that column block of the gradient, no bias share.

        Weight gradients are off BPTT's critical path (only input gradients feed the next step): the calls of the T steps
        wait in w_pend and go out as ONE launch per sweep on the main stream (_wgrad_issue / _launch_group); with
        REFID_OVERLAP_WGRAD=1 they are issued on a side stream instead, as in rounds 1-4."""
        side = WGRAD_STREAM.get(g.device) if overlap_wgrad() else None
        if side is None:
            return self._wgrad(g, a, b, bias, i_base)
        for t in (g, a, b):
            if t is not None:
                t.record_stream(side)                       # allocator must not recycle them early
        # deferred: the launch happens at the next flush_wgrads() -- ONE cross-stream dependency per batch instead
        # of one event record + wait per weight-gradient call (~1500 per step; each left a ~7 us bubble)
        WGRAD_STREAM.pending.append((self, g, a, b, bias, i_base))
        if len(WGRAD_STREAM.pending) >= WGRAD_BATCH:
            flush_wgrads(g.device)

    def _wgrad(self, g, a, b=None, bias=True, i_base=0):
        if i_base:
            if self.kind != "conv" or b is not None or bias:
                raise RefidHipError(f"{self.name}: a column-block weight gradient is one source, no bias, stride-1 conv")
            ops.conv2d_wgrad(g, a, self.gw, kh=self.k, kw=self.k, stride=1, pad=self.pad, i_base=i_base, i_total=self.ci,
                             algo=0, phase=0)
            return
        if self.kind == "convT":
            # roles swapped (refid_hip.h): "g" := layer input (low res), "src" := output gradient: a 2x2 stride-2 weight
            # gradient in the IOHW layout; grouped over the time steps and reduced once per backward like the others
            # (round 5; it was a one-shot call per step: 69 partial-product + 69 reduction launches per step)
            if self.has_bias and bias:
                ops.colsum(g, self.gb)
            g, a, b, bias = a, g, None, False
            if CONVT_PWS and PWS_WGRAD and not self.bf16 and a.stride(2) == a.shape[3] and g.shape[2] % 32 == 0 and \
                    self._pws_plan_ok(self.ci, 2 * self.co, 2 * self.co, 4 * self.co):
                # non-overlapping patches: ONE streaming 1x1 weight gradient with K = (dy, dx, co) over the even / odd rows of
                # the output gradient (refid_wgrad_desc.algo 8); the direct 2x2 tile sat at 0.30 of the fp32 pipe
                return self._wgrad_issue(g, a, None, False, 8)
            return self._wgrad_issue(g, a, None, False, 0)
        gb = self.gb if bias else None
        algo = 1 if (USE_WINOGRAD and self.kind == "conv" and self.k == 3 and self.co >= WGRAD_WINO_MIN_CO and self.ci >= 32) else 0
        if algo == 1 and b is not None and a.shape[3] % 32 != 0:
            algo = 0          # the Winograd weight-gradient tile picks the source per 32-channel tile (base 24, 40, 48 ...)
        if algo == 1 and WGRAD_F4 and min(g.shape[1], g.shape[2]) >= WGRAD_F4_MIN_HW and g.shape[3] % 4 == 0 and \
                a.shape[3] % 4 == 0 and (b is None or b.shape[3] % 4 == 0):
            algo = 5          # Winograd over 2x4 tiles: 0.75x the fp32 MFMAs of algo 1, packed transforms
        if self.kind == "down" and WGRAD_DOWN_F4 and USE_WINOGRAD and self.co >= 32 and self.ci >= 32 and a.shape[3] % 32 == 0 and \
                (b is None or b.shape[3] % 32 == 0) and a.shape[1] % 2 == 0 and a.shape[2] % 2 == 0 and \
                min(g.shape[1], g.shape[2]) >= WGRAD_F4_MIN_HW and g.shape[3] % 4 == 0:
            algo = 7          # the 2x4-tile Winograd kernel on the four parity phases of the input
        if self.bf16 and self.kind == "conv" and self.k == 3 and self.co > 32 and self.ci > 32:
            algo = 2          # bf16 matrix-core operands, fp32 accumulation (compute_dtype: bf16)
        return self._wgrad_issue(g, a, b, bias, algo)

    def _wgrad_issue(self, g, a, b, bias, algo):
        """Queue (grouped time steps) or launch the partial products of one call with the chosen algorithm."""
        gb = self.gb if bias else None
        pws = PWS_WGRAD and self.kind == "conv" and self.k == 1 and algo == 0 and self._pws_ok(a, b)
        if self.w_group > 1 and ((self.kind == "conv" and self.k == 3 and self.ci > 4) or self.kind in ("down", "convT") or pws):
            # same source split, algorithm and bias mode as the waiting calls (the first recurrent step has no second source yet)?
            if self.w_pend and ((self.w_pend[0][2] is None) != (b is None) or self.w_algo != algo or self.w_bias != bias):
                self._launch_group()
            self.w_pend.append((g, a, b))
            self.w_algo, self.w_bias = algo, bias
            if len(self.w_pend) >= self.w_group:
                self._launch_group()
            return
        self._slab_layout(algo)
        self.wslab = ops.conv2d_wgrad(g, a, self.gw, in_b=b, db=gb, algo=algo, phase=1 if self.w_calls == 0 else 2,
                                      slabs=self.wslab, **self._wg_geo())
        self.w_calls += 1
        self.w_last = (g, a, b, algo)

    def _wg_geo(self):
        """Geometry keywords of this op's refid_conv2d_wgrad calls (ConvTranspose2d: the 2x2 stride-2 conv of the swapped roles)."""
        if self.kind == "convT":
            return dict(kh=2, kw=2, stride=2, pad=0, i_total=self.co)
        return dict(kh=self.k, kw=self.k, stride=self.stride, pad=self.pad, i_total=self.ci)

    def _pws_ok(self, a, b):
        """Mirror of refid_wgrad_pws_ok (csrc/wgrad_pws.hip): does this 1x1 weight gradient take the streaming form?  (Only it
        can take several time steps per launch.)"""
        return self._pws_plan_ok(self.co, a.shape[3], b.shape[3] if b is not None else 0, self.ci)

    @staticmethod
    def _pws_plan_ok(co, ca, cb, ci):
        if co < 64 or co % 32 or ca % 32 or cb % 32 or ci % 32:
            return False
        wi = 4 if ci >= 128 else (2 if ci >= 64 else 1)
        while cb and wi > 1 and ca % (32 * wi):
            wi //= 2
        return not (co >= 128 and wi == 1 and cb and ca % 64)

    def _slab_layout(self, algo):
        """The partial-sum slabs persist over the calls of one backward pass, and their layout belongs to the algorithm
        ([split][16 xi][co][ci] for the Winograd tiles, [split][tap][co][ci] for the direct ones).  The algorithm is a
        function of the op's geometry -- except for a two-source call whose split is not a multiple of the Winograd
        tile's 32 channels -- so a change between two calls is rare; when it happens, what has accumulated is reduced
        into the parameter gradient (phase 3 ACCUMULATES into dw) and the slabs start over."""
        if self.w_calls and self.w_last is not None and self.w_last[3] != algo:
            g, a, b, old = self.w_last
            ops.conv2d_wgrad(g, a, self.gw, in_b=b, db=self.gb if self.kind != "convT" else None, algo=old, phase=3,
                             slabs=self.wslab, **self._wg_geo())
            self.w_calls = 0

    def _launch_group(self):
        (g, a, b), more = self.w_pend[0], self.w_pend[1:]
        self._slab_layout(self.w_algo)
        self.wslab = ops.conv2d_wgrad(g, a, self.gw, in_b=b, db=self.gb if self.w_bias else None, algo=self.w_algo,
                                      phase=1 if self.w_calls == 0 else 2, slabs=self.wslab, more=more, **self._wg_geo())
        self.w_calls += 1
        self.w_last = (g, a, b, self.w_algo)
        self.w_pend = []

    def finish_wgrad(self, batched=False):
        """Reduce the accumulated slabs into the parameter gradient (once per step, after BPTT).  batched: the element-wise
        stage is only queued (phase 4); the caller ends its loop over the ops with ops.wgrad_finish_flush()
        (finish_wgrads below)."""
        side = WGRAD_STREAM.get(self.w.device) if overlap_wgrad() else None
        if side is not None:
            flush_wgrads(self.w.device)                    # this op's launches may still be deferred
        if self.w_calls == 0 and not self.w_pend:
            return
        if side is not None:
            with torch.cuda.stream(side):
                self._finish_wgrad()
        else:
            self._finish_wgrad(batched)

    def _finish_wgrad(self, batched=False):
        if self.w_pend:
            self._launch_group()                           # the last, possibly shorter, group
        g, a, b, algo = self.w_last
        ops.conv2d_wgrad(g, a, self.gw, in_b=b, db=self.gb if self.kind != "convT" else None, algo=algo,
                         phase=4 if batched else 3, slabs=self.wslab, **self._wg_geo())
        self.w_calls = 0
        self.w_last = None


def finish_wgrads(op_list):
    """finish_wgrad() of every op, with the ~130 small element-wise reduction stages of a step issued as one launch per
    kernel family (refid_wgrad_desc.phase 4 + refid_wgrad_finish_flush; REFID_FINISH_BATCH=0: one launch per op).  The ops'
    gradient tensors are distinct, so the queued stages are independent."""
    batched = FINISH_BATCH and not overlap_wgrad()
    try:
        for o in op_list:
            o.finish_wgrad(batched)
    finally:
        try:
            if batched:
                ops.wgrad_finish_flush()                   # (also after an error: nothing stays queued in the library)
        finally:
            ops.rows_sum_flush()                           # the per-channel sums queued since rows_sum_defer() (no-op otherwise;
                                                           #  its own `finally`: a failing slab flush must not leave them queued)


class _Trunk:
    """ConvResidualBlocks (rsm:719-726): conv3x3 + LeakyReLU(.1), then num_block ResidualBlockNoBN (rsm:755-758; every shipped
    YAML: one)."""

    def __init__(self, arena, prefix, bf16=False):
        self.c0 = ConvOp(arena, prefix + ".0", bf16=bf16)
        self.blocks = []
        k = 0
        while f"{prefix}.2.{k}.conv1.weight" in arena.shapes:
            self.blocks.append((ConvOp(arena, f"{prefix}.2.{k}.conv1", bf16=bf16), ConvOp(arena, f"{prefix}.2.{k}.conv2", bf16=bf16)))
            k += 1
        self.c1, self.c2 = self.blocks[0]
        self.C = self.c0.co

    def ops(self):
        return [self.c0] + [c for blk in self.blocks for c in blk]


class _Egaca:
    def __init__(self, arena, a, bf16=False):
        self.a = a
        P, G = arena.p, arena.g
        self.conv1 = ConvOp(arena, a + ".conv1", bf16=bf16)
        self.conv1_e = ConvOp(arena, a + ".conv1_e", bf16=bf16)
        self.conv3 = ConvOp(arena, a + ".conv3", scale_name=a + ".beta", bf16=bf16)
        self.conv4 = ConvOp(arena, a + ".conv4", bf16=bf16)
        self.conv5 = ConvOp(arena, a + ".conv5", scale_name=a + ".gamma", bf16=bf16)
        self.side = ConvOp(arena, a + ".conv_y_side", bf16=bf16)
        self.c = self.conv1.ci
        # conv_y_side(y) + gamma * conv5(f4) (fm:331) as ONE 1x1 conv over the concatenated operand [f4 | y]: the two packed
        # weights are chunk-major ([chunk][row][8]) with the same row padding, so conv5's chunks followed by side's ARE the
        # packing of [gamma * W5 | W_side]; the `side` tensor (a write + a read of the widest EGACA tensor per call) and its
        # launch disappear from the forward pass.  Backward keeps the two convs (their gradients are separate tensors anyway).
        self.wp_cat = self.b_cat = None
        c5, sd = self.conv5, self.side
        if LINEAR_SPLIT and c5.f_algo == 3 and sd.f_algo == 3 and not c5.bf16 and (c5.f_pad, c5.f_kc, c5.f_bn) == (sd.f_pad, sd.f_kc, sd.f_bn) \
                and c5.ci % 8 == 0 and c5.has_bias and sd.has_bias:
            n5 = c5.wp.numel()
            self.wp_cat = torch.empty(n5 + sd.wp.numel(), dtype=torch.float32, device=c5.wp.device)
            c5.wp, sd.wp = self.wp_cat[:n5], self.wp_cat[n5:]          # (before the pack plan is built: it packs into these views)
            self.b_cat = torch.empty_like(sd.b)
        self.names = {n: (P(f"{a}.{n}"), G(f"{a}.{n}")) for n in (
            "norm1.weight", "norm1.bias", "norm1_e.weight", "norm1_e.bias", "norm2.weight", "norm2.bias",
            "conv2.weight", "conv2.bias", "conv2_e.weight", "conv2_e.bias",
            "se_1.1.weight", "se_1.1.bias", "se_1.3.weight", "se_1.3.bias", "beta", "gamma")}

    def ops(self):
        return [self.conv1, self.conv1_e, self.conv3, self.conv4, self.conv5, self.side]

    def p(self, n):
        return self.names[n][0]

    def g(self, n):
        return self.names[n][1]


class _EvrLevel:
    def __init__(self, arena, prefix, level, fuse, dead_down=False, bf16=False):
        self.level = level
        self.conv = ConvOp(arena, prefix + ".conv.conv2d", bf16=bf16) if level != 1 else None
        self.att = _Egaca(arena, prefix + ".atten_fuse", bf16=bf16) if level == 1 else None
        self.trunk = _Trunk(arena, prefix + ".recurrent_block.forward_trunk.main", bf16=bf16)
        self.fuse = ConvOp(arena, prefix + ".fuse_two_dir.conv2d", bf16=bf16) if fuse else None
        self.down = None if dead_down else ConvOp(arena, prefix + ".down", kind="down", bf16=bf16)
        self.C = self.trunk.C
        self.q_const = self.p_fuse = self.zero_s = None     # per-forward constants of the linearity split (Engine.forward)

    def ops(self):
        r = self.trunk.ops()
        for o in (self.conv, self.fuse, self.down):
            if o is not None:
                r.append(o)
        if self.att is not None:
            r += self.att.ops()
        return r


class Engine:
    """Forward + BPTT backward of FinalBidirectionAttenfusion on one GPU."""

    def __init__(self, img_chn, ev_chn=2, out_chn=3, base=32, num_residual_blocks=2, device="cuda",
                 compute_dtype="fp32", num_block=1, num_encoders=3):
        if num_encoders not in (2, 3, 4):
            # (level 1 is the EGACA level, so two at least; more than four levels would need more forward-wavefront streams and
            #  H, W multiples of 32: nothing in the reference uses them)
            raise ValueError("num_encoders must be 2, 3 or 4 (every options/*.yml of the reference: 3; the reference ctor's default: 4)")
        self.NL = NL = num_encoders
        if base not in (8, 16, 32, 64):
            # EGACA's LayerNorm / depthwise / squeeze-excite kernels take 2*base in {16, 32, 64, 128} channels; every
            # options/*.yml of the reference uses 32
            raise ValueError("base_num_channels must be 8, 16, 32 or 64 (the reference's configs use 32)")
        if compute_dtype not in ("fp32", "bf16", "bf16x3"):
            raise ValueError(f"compute_dtype must be 'fp32', 'bf16x3' or 'bf16', got {compute_dtype!r}")
        # "bf16x3": fp32 tensors, fp32 accumulation; the 3x3 forward / input-gradient convs multiply on the bf16 matrix
        # cores with every operand split in two bf16 numbers and three products per fp32 product (2^-16 relative per
        # product, 64x finer than TF32); weight gradients and everything else as in "fp32".  Explicit opt-in.
        ConvOp.default_split = 3 if compute_dtype == "bf16x3" else 0
        # "bf16": BASELINE config 3 -- conv forward / input-gradient operands in bf16 on the matrix cores,
        # fp32 accumulation, fp32 master weights, activations, weight gradients, loss, grad-norm, optimizer
        self.compute_dtype = compute_dtype
        bf = compute_dtype == "bf16"
        self.img_chn, self.ev_chn, self.out_chn, self.base = img_chn, ev_chn, out_chn, base
        self.nres = num_residual_blocks
        self.device = torch.device(device)
        self.shapes = param_shapes(img_chn, ev_chn, out_chn, base, num_residual_blocks, num_block, NL)
        self.arena = A = ParamArena(self.shapes, self.device)
        self.head_ev = ConvOp(A, "head.conv2d", need_dgrad=False, bf16=bf)
        self.head_img = ConvOp(A, "head_img.conv2d", need_dgrad=False, bf16=bf)
        self.enc_b = [_EvrLevel(A, f"encoders_backward.{i}", i, False, dead_down=(i == NL - 1), bf16=bf) for i in range(NL)]
        self.enc_f = [_EvrLevel(A, f"encoders_forward.{i}", i, True, bf16=bf) for i in range(NL)]
        self.img = []
        for i in range(NL):
            p = f"img_encoders.{i}"
            self.img.append(dict(identity=ConvOp(A, p + ".identity", bf16=bf), conv_1=ConvOp(A, p + ".conv_1", bf16=bf),
                                 conv_2=ConvOp(A, p + ".conv_2", bf16=bf),
                                 down=ConvOp(A, p + ".down", kind="down", bf16=bf)))
        self.res = [(ConvOp(A, f"resblocks.{i}.conv1", bf16=bf), ConvOp(A, f"resblocks.{i}.conv2", bf16=bf))
                    for i in range(self.nres)]
        self.dec = []
        for j in range(NL):
            p = f"decoders.{j}"
            self.dec.append(dict(t2=ConvOp(A, p + ".transposed_conv2d", kind="convT", bf16=bf),
                                 trunk=_Trunk(A, p + ".forward_trunk.main", bf16=bf)))
        self.pred = ConvOp(A, "pred.conv2d", bf16=bf)
        ConvOp.default_split = 0                   # (a construction-time context, not a setting other engines inherit)
        self.all_ops = [self.head_ev, self.head_img, self.pred]
        for lv in self.enc_b + self.enc_f:
            self.all_ops += lv.ops()
        for e in self.img:
            self.all_ops += list(e.values())
        for c1, c2 in self.res:
            self.all_ops += [c1, c2]
        for d in self.dec:
            self.all_ops += [d["t2"]] + d["trunk"].ops()
        self.early_ops = [self.pred]
        for lv in self.enc_f:
            self.early_ops += lv.ops()
        for c1, c2 in self.res:
            self.early_ops += [c1, c2]
        for d in self.dec:
            self.early_ops += [d["t2"]] + d["trunk"].ops()
        self.packed_version = -1
        self.param_version = 0
        self._pack_plan = None
        self.ctx = None
        self._bind_fold_scratch()
        # convs whose weights the T (or 2T) recurrent steps share: their weight gradients may wait for a group of steps
        img_ops = {id(o) for e in self.img for o in e.values()} | {id(self.head_img), id(self.head_ev)}
        self.recurrent_ops = [o for o in self.all_ops if id(o) not in img_ops]

    def _set_wgrad_groups(self, T, small=False, pixels=0):
        """Group size of the deferred weight-gradient launches: min(WGRAD_GROUP, T) on the recurrent convs (a group
        never outlives a sweep; small batches: WGRAD_GROUP_SMALL), 1 elsewhere; capped by the HBM that is still free
        (`pixels` = B H W of the step: a waiting call keeps its step's tensors alive)."""
        n = max(1, min(WGRAD_GROUP_SMALL if small else WGRAD_GROUP, T))
        n = wgrad_group_cap(n, pixels, self.device)
        for o in self.recurrent_ops:
            o.w_group = n

    def _bind_fold_scratch(self):
        self.folded_ops = [o for o in self.all_ops if o.scale is not None]
        n = sum((o.gw_arena.numel() + 3) // 4 * 4 + (o.gb_arena.numel() + 3) // 4 * 4 for o in self.folded_ops)
        self.fold_scratch = torch.zeros(max(n, 4), dtype=torch.float32, device=self.device)
        off = 0
        for o in self.folded_ops:
            for attr, ref in (("gw", o.gw_arena), ("gb", o.gb_arena)):
                k = ref.numel()
                setattr(o, attr, self.fold_scratch[off:off + k].view(ref.shape))
                off += (k + 3) // 4 * 4

    # -------------------------------------------------------------------------------------------
    def mark_params_changed(self):
        self.param_version += 1

    def repack(self):
        """Packed copies of the weights, once per optimiser step: ONE launch for all ~220 packings (ops.PackPlan; built on
        first use -- the parameter arena and the packed buffers never move).  REFID_PACK_BATCH=0: one launch per packing."""
        if self.packed_version == self.param_version:
            return
        if PACK_BATCH:
            if self._pack_plan is None:
                plan = ops.PackPlan(self.all_ops[0].w.device)
                for o in self.all_ops:
                    o.plan_repack(plan)
                self._pack_plan = plan.build()
            self._pack_plan.run()
            self.arena.pack_epoch += 1                     # the lazily packed fallback layouts are stale now
        else:
            for o in self.all_ops:
                o.repack()
        for lv in self.enc_b + self.enc_f:
            if lv.att is not None and lv.att.wp_cat is not None:
                ops.add(lv.att.conv5.b_eff, lv.att.side.b, out=lv.att.b_cat)      # bias of the merged conv5 + conv_y_side
        self.packed_version = self.param_version

    # -------------------------------------------------------------------------------------------
    # EGACA
    # -------------------------------------------------------------------------------------------
    def _egaca_img_path(self, A, img):
        """xi = GELU(dw3x3(conv1(LN1(img)))): t-independent, once per sweep (fm:300,303-305)."""
        n, h, w, c = img.shape
        if EGACA_FUSED and 32 < c <= 64 and c % 8 == 0 and A.conv1.f_algo == 3:
            ln_i = torch.empty_like(img)                      # (kept: conv1's weight gradient reads it)
            c1i = A.conv1.fwd(img, pw=dict(ln_gamma=A.p("norm1.weight"), ln_beta=A.p("norm1.bias"), ln_out=ln_i))
        else:
            ln_i = ops.layernorm2d_fwd(img, A.p("norm1.weight"), A.p("norm1.bias"))
            c1i = A.conv1.fwd(ln_i)
        dwi, xi = ops.dwconv3x3_gelu_fwd(c1i, A.p("conv2.weight"), A.p("conv2.bias"))
        return dict(ln_i=ln_i, c1i=c1i, dwi=dwi, xi=xi, gxi=None)

    def _egaca_fwd(self, A, ev, img, ip, st):
        n, h, w, c = ev.shape
        # (every conv3 workgroup re-reduces the pool partials of its sample: bounded to 256 rows, i.e. 256x256 pixels)
        if EGACA_FUSED and (h * w) % 128 == 0 and 32 < c <= 64 and c % 8 == 0 and A.conv1_e.f_algo == 3 and \
                ops.dwconv_pool_parts(h, w, c) <= 256:
            return self._egaca_fwd_fused(A, ev, img, ip, st)
        ln_e = ops.layernorm2d_fwd(ev, A.p("norm1_e.weight"), A.p("norm1_e.bias"))
        c1e = A.conv1_e.fwd(ln_e)
        dwe, xe, pool = ops.dwconv3x3_gelu_fwd(c1e, A.p("conv2_e.weight"), A.p("conv2_e.bias"), want_pool=True)
        m, z1, s = ops.se_fwd(pool, 1.0 / (h * w), A.p("se_1.1.weight"), A.p("se_1.1.bias"),
                              A.p("se_1.3.weight"), A.p("se_1.3.bias"))
        xs = ops.scale_cat(ip["xi"], xe, s)
        evimg = ops.add(ev, img)
        y = A.conv3.fwd(xs, res=evimg)                    # ev + img + beta * conv3(.)   (fm:319)
        ln2 = ops.layernorm2d_fwd(y, A.p("norm2.weight"), A.p("norm2.bias"))
        c4 = A.conv4.fwd(ln2)
        f4 = ops.gelu_fwd(c4)
        side = A.side.fwd(y)
        out = A.conv5.fwd(f4, res=side)                   # conv_y_side(y) + gamma * conv5(.) (fm:331)
        if st is not None:
            st["eg"] = dict(ev=ev, ln_e=ln_e, c1e=c1e, dwe=dwe, xe=xe, m=m, z1=z1, s=s, xs=xs, y=y, ln2=ln2,
                            c4=c4, f4=f4)
        return out

    def _egaca_fwd_fused(self, A, ev, img, ip, st):
        """fusion_modules.py:290-333 in six launches (csrc/conv_pw.hip, refid_pw_extras):
          1  c1e = conv1_e(LN1e(ev))                          LayerNorm2d in the conv's prologue
          2  dwe, xe, pool partials = GELU(dw3x3(c1e))
          3  y = ev + img + beta * conv3([xi*s | xe*s])      s = sigmoid(W2 relu(W1 mean(xe) + b1) + b2) computed per
                                                              workgroup from the pool partials and applied to the operand
          4  c4 = conv4(LN2(y)), f4 = GELU(c4)                LayerNorm prologue + GELU second output
          5  side = conv_y_side(y)
          6  out = side + gamma * conv5(f4)
        In training the tensors the (unchanged) backward reads -- ln_e, xs, m / z1 / s, ln2, c4 -- are side outputs."""
        n, h, w, c = ev.shape
        save = st is not None
        dev = ev.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)          # noqa: E731
        ln_e = new(n, h, w, c) if save else None
        c1e = A.conv1_e.fwd(ev, pw=dict(ln_gamma=A.p("norm1_e.weight"), ln_beta=A.p("norm1_e.bias"), ln_out=ln_e))
        dwe, xe, pool = ops.dwconv3x3_gelu_fwd(c1e, A.p("conv2_e.weight"), A.p("conv2_e.bias"), want_pool=True)
        m = z1 = s = xs = None
        if save:
            m, z1, s, xs = new(n, c), new(n, c // 2), new(n, c), new(n, h, w, 2 * c)
        y = A.conv3.fwd(ip["xi"], xe, res=ev,
                        pw=dict(pool=pool, hw=h * w, se_w1=A.p("se_1.1.weight"), se_b1=A.p("se_1.1.bias"),
                                se_w2=A.p("se_1.3.weight"), se_b2=A.p("se_1.3.bias"), se_m=m, se_z1=z1, se_s=s,
                                xs_out=xs, res2=img))
        ln2 = new(n, h, w, c) if save else None
        f4 = new(n, h, w, A.conv4.co)
        c4 = A.conv4.fwd(y, pw=dict(ln_gamma=A.p("norm2.weight"), ln_beta=A.p("norm2.bias"), ln_out=ln2, out2=f4))
        if A.wp_cat is not None:
            out = new(n, h, w, A.conv5.co)
            ops.conv2d(f4, A.wp_cat, out, kh=1, kw=1, stride=1, pad=0, mode=0, cout=A.conv5.co, cout_pad=A.conv5.f_pad, in_b=y,
                       bias=A.b_cat, algo=3)
        else:
            side = A.side.fwd(y)
            out = A.conv5.fwd(f4, res=side)
        if save:
            st["eg"] = dict(ev=ev, ln_e=ln_e, c1e=c1e, dwe=dwe, xe=xe, m=m, z1=z1, s=s, xs=xs, y=y, ln2=ln2, c4=c4, f4=f4)
        return out

    def _egaca_bwd(self, A, g_u, img_grad, ip, st, gy_keep=None):
        """Returns gradient w.r.t. the event input; adds the image-input gradient into img_grad -- or, with gy_keep, appends
        it to that list (summed once per sweep: Engine.backward_early / backward_late)."""
        e = st["eg"]
        A.conv5.wgrad(g_u, e["f4"])
        if A.conv5.d_algo == 3 and A.conv5.wdp6 is None and LINEAR_SPLIT:
            g_c4 = A.conv5.dgrad(g_u, mask=e["c4"], gelu_mask=True)      # GELU' rides in the tile's mask epilogue
        else:
            g_f4 = A.conv5.dgrad(g_u)
            g_c4 = ops.gelu_bwd(g_f4, e["c4"], out=g_f4)
        A.conv4.wgrad(g_c4, e["ln2"])
        g_ln2 = A.conv4.dgrad(g_c4)
        A.side.wgrad(g_u, e["y"])
        g_y = A.side.dgrad(g_u)
        ops.layernorm2d_bwd(g_ln2, e["y"], A.p("norm2.weight"), g_y, A.g("norm2.weight"), A.g("norm2.bias"),
                            accumulate=True)
        A.conv3.wgrad(g_y, e["xs"])
        g_xs = A.conv3.dgrad(g_y)
        gs = ops.egaca_gs_reduce(g_xs, ip["xi"], e["xe"])
        gm = ops.se_bwd(gs, e["s"], e["z1"], e["m"], A.p("se_1.1.weight"), A.p("se_1.3.weight"),
                        A.g("se_1.1.weight"), A.g("se_1.1.bias"), A.g("se_1.3.weight"), A.g("se_1.3.bias"))
        first = ip["gxi"] is None
        if first:
            ip["gxi"] = torch.empty_like(ip["xi"])
        g_dwe = ops.egaca_bwd_elem(g_xs, e["s"], gm, e["dwe"], ip["gxi"], accumulate_xi=not first)
        g_c1e = ops.dwconv3x3_bwd(g_dwe, e["c1e"], A.p("conv2_e.weight"), A.g("conv2_e.weight"), A.g("conv2_e.bias"))
        A.conv1_e.wgrad(g_c1e, e["ln_e"])
        g_lne = A.conv1_e.dgrad(g_c1e)
        if gy_keep is not None:
            gy_keep.append(g_y)                           # y = ev + img + ...: image gets g_y (deferred sum)
        else:
            ops.add(img_grad, g_y, out=img_grad)
        # NOT in place: conv3's weight-gradient kernel may still be reading g_y on the side stream
        g_ev = ops.layernorm2d_bwd(g_lne, e["ev"], A.p("norm1_e.weight"), torch.empty_like(g_y),
                                   A.g("norm1_e.weight"), A.g("norm1_e.bias"), res=g_y)
        return g_ev                                       # = g_y + LN1e-backward: gradient w.r.t. ev

    def _egaca_img_bwd(self, A, img, img_grad, ip):
        if ip["gxi"] is None:
            return
        g_dwi = ops.gelu_bwd(ip["gxi"], ip["dwi"])
        g_c1i = ops.dwconv3x3_bwd(g_dwi, ip["c1i"], A.p("conv2.weight"), A.g("conv2.weight"), A.g("conv2.bias"))
        A.conv1.wgrad(g_c1i, ip["ln_i"])
        g_lni = A.conv1.dgrad(g_c1i)
        ops.layernorm2d_bwd(g_lni, img, A.p("norm1.weight"), img_grad, A.g("norm1.weight"), A.g("norm1.bias"),
                            accumulate=True)

    def _egaca_fold_back(self, A):
        for conv, nm in ((A.conv3, "beta"), (A.conv5, "gamma")):
            ops.fold_back(conv.w, conv.b, conv.scale, conv.gw, conv.gb, conv.gw_arena, conv.gb_arena,
                          A.g(nm).view(-1))

    # -------------------------------------------------------------------------------------------
    # trunk = EvR hidden-state update (rsm:659-678, 719-726, 755-758)
    # -------------------------------------------------------------------------------------------
    @staticmethod
    def _trunk_fwd(T, u, h, st, plus=None):
        """plus: also returns s + plus (the next decoder's input sum, written by the last conv's tile)."""
        v = T.c0.fwd(u, h, slope_pre=0.1)
        xs, rs, x = [], [], v                            # block k: x_{k+1} = x_k + conv2(relu(conv1(x_k))), x_0 = v
        for k, (c1, c2) in enumerate(T.blocks):
            r = c1.fwd(x, slope_pre=0.0)
            xs.append(x); rs.append(r)
            x = c2.fwd(r, res=x, plus=plus if k == len(T.blocks) - 1 else None)
        s, sp = x if plus is not None else (x, None)
        if st is not None:
            st.update(u=u, h=h, v=v, r=rs[0], xs=xs, rs=rs, s=s)
        return s if plus is None else (s, sp)

    @staticmethod
    def _trunk_bwd(T, g_s, st, mask_u=None, slope_u=1.0):
        u, h, v = st["u"], st["h"], st["v"]
        C = T.C
        g = g_s
        for k in range(len(T.blocks) - 1, -1, -1):       # x_{k+1} = x_k + conv2(relu(conv1(x_k)))
            c1, c2 = T.blocks[k]
            x, r = st["xs"][k], st["rs"][k]
            c2.wgrad(g, r)
            g_r = c2.dgrad(g, mask=r, slope_mask=0.0)
            c1.wgrad(g_r, x)
            # x_0 = v is LeakyReLU(.1) of main.0's output: its derivative mask rides on the last tile; x_k (k > 0) is linear
            g = c1.dgrad(g_r, res=g, mask=v, slope_mask=0.1) if k == 0 else c1.dgrad(g_r, res=g)
        g_v = g
        T.c0.wgrad(g_v, u, h)
        g_u = T.c0.dgrad(g_v, rows=(0, C), mask=mask_u, slope_mask=slope_u)
        g_h = T.c0.dgrad(g_v, rows=(C, C)) if h is not None else None
        return g_u, g_h

    # -------------------------------------------------------------------------------------------
    # EvR level (rsm:270-296)
    # -------------------------------------------------------------------------------------------
    def _evr_fwd(self, L, a, xb, h_prev, Sb, ip, st, plus=None):
        """Returns (level output, new state) -- and, with plus, (level output + plus) as a third element."""
        i = L.level
        if i == 0:
            src = a
            u = L.conv.fwd(a, slope_pre=0.04)                 # LeakyReLU(.2) twice (rsm:81-82,284-285)
        elif i >= 2:
            if L.q_const is not None:                         # C3(a + x_blocks[i-1]) + bias = C3(a) + q_const
                src = a
                u = L.conv.fwd(a, res=L.q_const, slope_post=0.04, bias=False)
            else:
                src = ops.add(a, xb[i - 1])
                u = L.conv.fwd(src, slope_pre=0.04)
        else:
            src = a
            u = self._egaca_fwd(L.att, a, xb[0], ip, st)
        s = self._trunk_fwd(L.trunk, u, h_prev, st)
        f = s
        if L.fuse is not None:
            if L.p_fuse is not None:                          # C1([s | S_b]) + bias = C1_s(s) + p_fuse
                f = L.fuse.fwd(s, res=L.p_fuse, slope_post=0.2, bias=False)
            else:
                f = L.fuse.fwd(s, Sb, slope_pre=0.2)
        o = op = None
        if L.down is not None:
            o = L.down.fwd(f, plus=plus)
            if plus is not None:
                o, op = o
        if st is not None:
            st.update(src=src, f=f, Sb=Sb)
        return (o, s) if plus is None else (o, s, op)

    # -------------------------------------------------------------------------------------------
    def forward(self, x, event, save=True):
        """x: (B,img_chn,H,W) or (B,2,3,H,W); event: (B,T,ev_chn,H,W); returns (B,T,out_chn,H,W)."""
        if x.dim() == 5:
            x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3], x.shape[4])   # arch:140-141
        if x.dtype != torch.float32 or event.dtype != torch.float32 or not x.is_cuda or not event.is_cuda:
            raise RefidHipError("forward: float32 CUDA tensors required")
        B, T, nb, H, W = event.shape
        NL = self.NL
        if H % (1 << NL) or W % (1 << NL):      # reference: a shape error at the first skip sum
            raise RuntimeError(f"H and W must be multiples of {1 << NL}, got {H}x{W}")
        if x.shape != (B, self.img_chn, H, W) or nb != self.ev_chn:
            raise RuntimeError(f"unexpected input shapes x={tuple(x.shape)} event={tuple(event.shape)}")
        self.repack()
        dev = x.device
        x = x.contiguous()
        event = event.contiguous()
        x_in = ops.nchw_to_nhwc(x, _pad4(self.img_chn))
        ev_in = ops.nchw_to_nhwc_tb(event, _pad4(self.ev_chn))            # time-major: step t = samples t B .. (t+1) B
        head = self.head_img.fwd(x_in, slope_pre=0.2)                      # arch:147-148
        e_all = self.head_ev.fwd(ev_in, slope_pre=0.2)                     # arch:149
        xb, img_saved = [], []
        g = head
        for i in range(NL):                                                # rsm:41-49
            E = self.img[i]
            c1 = E["conv_1"].fwd(g, slope_pre=0.2)
            c2 = E["conv_2"].fwd(c1, slope_pre=0.2)
            sm = E["identity"].fwd(g, res=c2)
            o = E["down"].fwd(sm)
            img_saved.append((g, c1, c2, sm))
            xb.append(o)
            g = o
        ip_b = self._egaca_img_path(self.enc_b[1].att, xb[0])
        ip_f = self._egaca_img_path(self.enc_f[1].att, xb[0])

        # linearity split: the time-independent halves of the level-2 first conv and of pred (the fuse convs' follow the
        # backward sweep, which produces their operand)
        lin = LINEAR_SPLIT
        for L in self.enc_b[2:] + self.enc_f[2:]:                          # levels >= 2: C3(a + x_blocks[level - 1])
            L.q_const = L.conv.fwd(xb[L.level - 1]) if lin else None       # C3(x_blocks[level - 1]) + bias
        for L in self.enc_f:
            L.p_fuse = None
        q_pred = None
        if lin:
            q_pred = torch.zeros((B, H, W, _pad4(self.out_chn)), dtype=torch.float32, device=dev)
            self.pred.fwd(head, out=q_pred[..., :self.out_chn])            # pred(head) + bias

        main = torch.cuda.current_stream()
        pipe = use_pipeline(B, H, W)
        lvs = [main] + ([LV_STREAMS[k].get(dev) for k in range(NL - 1)] if pipe else [main] * (NL - 1))
        for s_ in lvs[1:]:
            if s_ is not main:
                s_.wait_stream(main)                   # image branch, event head: everything issued so far

        def level(L, i, cur, h_prev, Sb_i, ip, st, plus=None):
            """EvR level i on its stream, after the producer of `cur` (level i-1 of the same step)."""
            if lvs[i] is main and i == 0:
                return self._evr_fwd(L, cur, xb, h_prev, Sb_i, ip, st, plus)
            if lvs[i] is not lvs[i - 1]:
                lvs[i].wait_stream(lvs[i - 1])
                cur.record_stream(lvs[i])
            with torch.cuda.stream(lvs[i]):
                return self._evr_fwd(L, cur, xb, h_prev, Sb_i, ip, st, plus)

        hb = [None] * NL
        steps_b = []
        for t in range(T - 1, -1, -1):                                     # arch:172-181
            cur = e_all[t * B:(t + 1) * B]
            sts = []
            for i in range(NL):
                st = {} if save else None
                cur, hb[i] = level(self.enc_b[i], i, cur, hb[i], None, ip_b, st)
                sts.append(st)
            steps_b.append((t, sts))
        Sb = hb                                                            # aliasing: final states only
        if lin:
            # fuse_two_dir's constant half, C1_b(S_b,i) + bias, on the stream that produced S_b,i and runs level i next
            for i, L in enumerate(self.enc_f):
                with torch.cuda.stream(lvs[i]):
                    if L.fuse.can_fwd_from() and L.C % L.fuse.f_kc == 0 and _pad4(L.fuse.co) == L.fuse.co:
                        L.zero_s = None
                        L.p_fuse = L.fuse.fwd_from(Sb[i], L.C)             # the S_b block of the packed weights, one source
                    else:
                        if L.zero_s is None or L.zero_s.shape != Sb[i].shape:
                            L.zero_s = torch.zeros_like(Sb[i])             # (stands in for the s_t half; cached across steps)
                        L.p_fuse = L.fuse.fwd(L.zero_s, Sb[i])

        out = torch.empty((B, T, self.out_chn, H, W), dtype=torch.float32, device=dev)
        # pred's NHWC outputs of all T steps (time-major); converted to the (B,T,C,H,W) stack in one launch after the loop
        out4 = torch.zeros((T * B, H, W, _pad4(self.out_chn)), dtype=torch.float32, device=dev)
        hf = [None] * NL
        hd = [None] * NL
        steps_f = []
        dstream = DEC_STREAM.get(dev) if pipe else None
        if dstream is not None:
            dstream.wait_stream(main)                  # xb, head ...: everything issued so far

        fuse = FUSE_SUMS

        def decode(t, eb, sts, b0_in):
            bs = []
            z = eb[NL - 1]
            di = None
            for i, (c1, c2) in enumerate(self.res):                        # arch:199-203, rsm:488-503
                b0 = (b0_in if b0_in is not None else ops.add(z, xb[NL - 1])) if i == 0 else z
                b1 = c1.fwd(b0, slope_pre=0.0)
                if fuse and i == self.nres - 1:                            # decoder 0's input sum leaves with this tile
                    z, di = c2.fwd(b1, res=b0, slope_post=0.0, plus=eb[NL - 1])
                else:
                    z = c2.fwd(b1, res=b0, slope_post=0.0)
                bs.append((b0, b1, z))
            ds = []
            for j in range(NL):                                            # arch:210-212, rsm:386-408
                D = self.dec[j]
                if di is None:
                    di = ops.add(z, eb[NL - 1 - j])
                q = D["t2"].fwd(di)
                dst = {} if save else None
                if save:
                    dst["di"] = di
                di = None
                if fuse and j < NL - 1:                                    # the next decoder's input sum
                    z, di = self._trunk_fwd(D["trunk"], q, hd[j], dst, plus=eb[NL - 2 - j])
                else:
                    z = self._trunk_fwd(D["trunk"], q, hd[j], dst)
                hd[j] = z
                ds.append(dst)
            if q_pred is not None:                                         # pred(z + head) = pred(z) + q_pred
                pi = z
                self.pred.fwd(z, res=q_pred[..., :self.out_chn], out=out4[t * B:(t + 1) * B, ..., :self.out_chn], bias=False)
            else:
                pi = ops.add(z, head)
                self.pred.fwd(pi, out=out4[t * B:(t + 1) * B, ..., :self.out_chn])   # arch:215 (no activation)
            if save:
                steps_f.append(dict(lv=sts, bs=bs, ds=ds, pi=pi))

        for t in range(T):                                                 # arch:185-216
            cur = e_all[t * B:(t + 1) * B]
            sts, eb = [], []
            b0_in = None
            for i in range(NL):
                st = {} if save else None
                if fuse and i == NL - 1:               # b0 = e_blocks[-1] + x_blocks[-1] (arch:199-203) from the down tile
                    cur, hf[i], b0_in = level(self.enc_f[i], i, cur, hf[i], Sb[i], ip_f, st, plus=xb[NL - 1])
                else:
                    cur, hf[i] = level(self.enc_f[i], i, cur, hf[i], Sb[i], ip_f, st)
                sts.append(st)
                eb.append(cur)
            if dstream is None:
                decode(t, eb, sts, b0_in)
            else:
                for i in range(NL):
                    dstream.wait_stream(lvs[i])        # step t's encoder outputs
                    eb[i].record_stream(dstream)       # (allocated on another stream's pool)
                if b0_in is not None:
                    b0_in.record_stream(dstream)
                with torch.cuda.stream(dstream):
                    decode(t, eb, sts, b0_in)
        for s_ in lvs[1:] + ([dstream] if dstream is not None else []):
            if s_ is not main:
                main.wait_stream(s_)
        ops.nhwc_to_nchw_tb(out4[..., :self.out_chn], self.out_chn, out)    # arch:218 (torch.stack)
        if save:
            self.ctx = dict(B=B, T=T, H=H, W=W, x_in=x_in, ev_in=ev_in, head=head, e_all=e_all, xb=xb,
                            img_saved=img_saved, ip_b=ip_b, ip_f=ip_f, steps_b=steps_b, steps_f=steps_f, Sb=Sb, lin=lin)
        else:
            self.ctx = None
        return out

    # -------------------------------------------------------------------------------------------
    def zero_grad(self):
        self.arena.flat_g.zero_()

    def backward(self, gout, grad_sync=None):
        """BPTT.  gout: (B,T,out_chn,H,W) gradient of the loss w.r.t. forward()'s result.
        Parameter gradients are ACCUMULATED into the arena (call zero_grad() first).
        grad_sync(phase): optional hook, called with "early" once the forward-sweep / decoder /
        bottleneck / pred gradients are final (overlaps the rest of BPTT) and with "late" at the end."""
        st = self.backward_early(gout)
        if grad_sync is not None:
            grad_sync("early")
        self.backward_late(st)
        if grad_sync is not None:
            grad_sync("late")

    def backward_early(self, gout):
        """First half of BPTT (the forward sweep, walked t = T-1 .. 0): afterwards the gradients of the forward-sweep
        encoders, bottleneck, decoders and pred are final.  Returns the state backward_late() continues from.  (Two
        halves so that a data-parallel job can start its first all-reduce in between -- and so that each half can be
        captured in its own hipGraph with the collective issued eagerly between the replays.)"""
        c = self.ctx
        if c is None:
            raise RefidHipError("backward: no saved forward (call forward(save=True) first)")
        WGRAD_STREAM.pending.clear()          # leftovers of a backward that raised must never be launched
        for o in self.all_ops:                # ... nor may its half-filled slabs be added to (phase 2) or reduced
            o.w_calls, o.w_last, o.w_pend = 0, None, []
        self.fold_scratch.zero_()             # folded-weight gradients of THIS backward only (see ConvOp.__init__)
        self.ctx = None
        if ROWS_DEFER:
            ops.rows_sum_defer()              # per-channel gradient sums of this half: queued until finish_wgrads
        B, T, H, W = c["B"], c["T"], c["H"], c["W"]
        self._set_wgrad_groups(T, use_pipeline(B, H, W), B * H * W)
        dev = gout.device
        gout = gout.contiguous()
        xb, head, e_all, Sb = c["xb"], c["head"], c["e_all"], c["Sb"]
        zeros = lambda t: torch.zeros(t.shape, dtype=torch.float32, device=dev)  # noqa: E731
        lin = c["lin"]
        # pred has no activation (arch:215), so head's share sum_t dgrad(g_t) is dgrad(sum_t g_t): one launch
        g4_all = ops.nchw_to_nhwc_tb(gout, _pad4(self.out_chn))           # all T steps' output gradients, time-major, one launch
        g4sum = ops.nchw_tsum_to_nhwc(gout, _pad4(self.out_chn))
        g_head = self.pred.dgrad(g4sum)
        # gradients of the image branch's x_blocks: with the linearity split they are assembled once per sweep from the kept
        # per-step tensors (ops.sum_n / one input-gradient launch on the summed gradient), never accumulated step by step
        NL = self.NL
        g_xb = [None] * NL if lin else [zeros(t) for t in xb]
        g_Sb = [None] * NL
        g_e = torch.empty_like(e_all)
        # per-step gradients the deferred sums read (gu: per level >= 2, the constant operand x_blocks[level - 1]'s share)
        keep = dict(gy=[], gu={i: [] for i in range(2, NL)}, gb0=[], gf=tuple([] for _ in range(NL)))

        # ---------------- forward sweep, t = T-1 .. 0 -------------------------------------------
        g_hf = [None] * NL
        g_hd = [None] * NL
        for t in range(T - 1, -1, -1):
            S = c["steps_f"][t]
            g4 = g4_all[t * B:(t + 1) * B]
            self.pred.wgrad(g4, S["pi"])
            g_sd = self.pred.dgrad(g4, res=g_hd[NL - 1])      # + the last decoder's state gradient, fused
            g_skip = [None] * NL
            for j in range(NL - 1, -1, -1):
                D, dst = self.dec[j], S["ds"][j]
                g_q, g_hd[j] = self._trunk_bwd(D["trunk"], g_sd, dst)
                D["t2"].wgrad(g_q, dst["di"])
                if j > 0 and g_hd[j - 1] is not None and FUSE_SUMS:
                    g_di, g_sd = D["t2"].dgrad(g_q, plus=g_hd[j - 1])       # + the previous decoder's state gradient
                else:
                    g_di = D["t2"].dgrad(g_q)
                    if j > 0:
                        g_sd = g_di if g_hd[j - 1] is None else ops.add(g_di, g_hd[j - 1])
                g_skip[NL - 1 - j] = g_di
            # bottleneck
            g_z = g_skip[NL - 1]                              # decoder 0's di = z + e_blocks[-1]
            for i in range(self.nres - 1, -1, -1):
                c1, c2 = self.res[i]
                b0, b1, z = S["bs"][i]
                gz = ops.act_bwd(g_z, z, 0.0)
                c2.wgrad(gz, b1)
                g_b1 = c2.dgrad(gz, mask=b1, slope_mask=0.0)
                c1.wgrad(g_b1, b0)
                if i > 0:
                    g_z = c1.dgrad(g_b1, res=gz)              # b0 is the previous block's output
                elif FUSE_SUMS:
                    g_b0, g_o = c1.dgrad(g_b1, res=gz, plus=g_skip[NL - 1])     # the last level's output gradient leaves with the tile
                else:
                    g_b0 = c1.dgrad(g_b1, res=gz)
                    g_o = ops.add(g_b0, g_skip[NL - 1])
            if lin:
                keep["gb0"].append(g_b0)
            else:
                ops.add(g_xb[NL - 1], g_b0, out=g_xb[NL - 1])
            for i in range(NL - 1, -1, -1):
                L, st = self.enc_f[i], S["lv"][i]
                C = L.C
                L.down.wgrad(g_o, st["f"])
                g_f = L.down.dgrad(g_o, mask=st["f"], slope_mask=0.2)
                if lin:
                    L.fuse.wgrad(g_f, st["s"])                # the S_b half: once, on sum_t g_f (below)
                    keep["gf"][i].append(g_f)
                else:
                    L.fuse.wgrad(g_f, st["s"], st["Sb"])
                    if g_Sb[i] is None:
                        g_Sb[i] = L.fuse.dgrad(g_f, rows=(C, C))
                    else:
                        L.fuse.dgrad(g_f, rows=(C, C), res=g_Sb[i], out=g_Sb[i])
                g_s = L.fuse.dgrad(g_f, rows=(0, C), res=g_hf[i])
                g_o = self._evr_first_bwd(L, g_s, st, g_hf, g_xb, g_e, t, B, c["ip_f"], g_skip, True, keep if lin else None)

        if lin:
            # the time-independent operands' shares, from the summed per-step gradients
            self.pred.wgrad(g4sum, head, bias=False)                      # (sum_t g4) (x) head; bias: the per-step calls
            g_xb[NL - 1] = ops.sum_n(keep["gb0"])
            for i, L in enumerate(self.enc_f):
                gsum = ops.sum_n(keep["gf"][i])
                if L.zero_s is None:
                    L.fuse.wgrad(gsum, Sb[i], bias=False, i_base=L.C)     # (sum_t g_f) (x) S_b, into columns C .. 2C
                else:
                    L.fuse.wgrad(gsum, L.zero_s, Sb[i], bias=False)
                g_Sb[i] = L.fuse.dgrad(gsum, rows=(L.C, L.C))             # dL/dS_b = W_b^T sum_t g_f
            # (top level first: x_blocks[NL-1]'s gradient already holds the bottleneck's share, x_blocks[i-1] for i < NL-1... each
            #  level >= 2 adds its constant operand's share to g_xb[level - 1]: the first writer of that slot overwrites)
            for i in range(2, NL):
                self._lin_level2_tail(self.enc_f[i], keep["gu"][i], xb, g_xb, first=g_xb[i - 1] is None)
            g_xb[0] = ops.sum_n(keep["gy"])
        # forward-sweep, bottleneck, decoder and pred weights are final from here on -- except the
        # folded EGACA convs, un-folded now so the early bucket is complete
        self._egaca_img_bwd(self.enc_f[1].att, xb[0], g_xb[0], c["ip_f"])
        finish_wgrads(self.early_ops)
        WGRAD_STREAM.join(dev)
        self._egaca_fold_back(self.enc_f[1].att)
        return dict(c=c, B=B, dev=dev, g_xb=g_xb, g_Sb=g_Sb, g_e=g_e, g_head=g_head)

    def backward_late(self, state):
        """Second half of BPTT: the backward sweep (walked t = 0 .. T-1), event head and image branch."""
        c, B, dev, g_xb, g_Sb, g_e, g_head = (state[k] for k in ("c", "B", "dev", "g_xb", "g_Sb", "g_e", "g_head"))
        xb, head, e_all = c["xb"], c["head"], c["e_all"]
        if ROWS_DEFER:
            ops.rows_sum_defer()
        # ---------------- backward sweep (executed t = T-1..0), BPTT in reverse: t = 0 .. T-1 ----
        NL = self.NL
        g_hb = [None] * NL
        lin = c["lin"]
        keep = dict(gy=[], gu={i: [] for i in range(2, NL)})
        for t, sts in reversed(c["steps_b"]):
            g_o = None
            for i in range(NL - 1, -1, -1):
                L, st = self.enc_b[i], sts[i]
                carry = g_hb[i] if g_hb[i] is not None else g_Sb[i]     # t == 0: dL/dS_b,i
                if i == NL - 1:                                         # (its conv_down is dead: arch:181 discards the output)
                    g_s = carry
                else:
                    L.down.wgrad(g_o, st["s"])
                    g_s = L.down.dgrad(g_o, res=carry)
                g_o = self._evr_first_bwd(L, g_s, st, g_hb, g_xb, g_e, t, B, c["ip_b"], None, False, keep if lin else None)
        if lin:
            for i in range(2, NL):
                self._lin_level2_tail(self.enc_b[i], keep["gu"][i], xb, g_xb, first=False)
            ops.sum_n([g_xb[0]] + keep["gy"], out=g_xb[0])

        # ---------------- t-independent tails ----------------------------------------------------
        self._egaca_img_bwd(self.enc_b[1].att, xb[0], g_xb[0], c["ip_b"])
        gz_e = ops.act_bwd(g_e, e_all, 0.2, out=g_e)
        self.head_ev.wgrad(gz_e, c["ev_in"])
        g = g_xb[NL - 1]
        for i in range(NL - 1, -1, -1):
            E = self.img[i]
            gin, c1, c2, sm = c["img_saved"][i]
            E["down"].wgrad(g, sm)
            g_sm = E["down"].dgrad(g)
            E["identity"].wgrad(g_sm, gin)
            gz2 = ops.act_bwd(g_sm, c2, 0.2)
            E["conv_2"].wgrad(gz2, c1)
            g_c1 = E["conv_2"].dgrad(gz2, mask=c1, slope_mask=0.2)
            E["conv_1"].wgrad(g_c1, gin)
            acc = g_xb[i - 1] if i > 0 else g_head
            t1 = E["identity"].dgrad(g_sm, res=acc)
            if i > 0:
                g = E["conv_1"].dgrad(g_c1, res=t1)
            else:
                g = E["conv_1"].dgrad(g_c1, res=t1, mask=head, slope_mask=0.2)
        self.head_img.wgrad(g, c["x_in"])
        finish_wgrads(self.all_ops)
        WGRAD_STREAM.join(dev)
        self._egaca_fold_back(self.enc_b[1].att)

    def _lin_level2_tail(self, L, gu, xb, g_xb, first):
        """First conv of a level >= 2, linearity split: the share of the constant operand x_blocks[level - 1] -- its
        weight-gradient term (sum_t g_u) (x) x_blocks[level - 1] and its input gradient W^T sum_t g_u -- from ONE summed gradient
        per sweep."""
        k = L.level - 1
        gsum = ops.sum_n(gu)
        L.conv.wgrad(gsum, xb[k], bias=False)
        if first:
            g_xb[k] = L.conv.dgrad(gsum)
        else:
            L.conv.dgrad(gsum, res=g_xb[k], out=g_xb[k])

    def _evr_first_bwd(self, L, g_s, st, g_h, g_xb, g_e, t, B, ip, g_skip, first_writer, keep=None):
        """Trunk + first op of an EvR level; returns the gradient w.r.t. the level's input
        (already including the decoder skip gradient when g_skip is given).  keep: the linearity split's lists of per-step
        gradients (None / the split switched off: accumulate step by step as the reference's autograd does)."""
        i = L.level
        lin = keep is not None
        if i == 1:
            g_u, g_h[i] = self._trunk_bwd(L.trunk, g_s, st)
            g_in = self._egaca_bwd(L.att, g_u, None if lin else g_xb[0], ip, st, keep["gy"] if lin else None)
            if g_skip is not None:
                g_in = ops.add(g_in, g_skip[0], out=g_in)
            return g_in
        g_u, g_h[i] = self._trunk_bwd(L.trunk, g_s, st, mask_u=st["u"], slope_u=0.04)
        L.conv.wgrad(g_u, st["src"])
        if i >= 2:
            if lin:
                keep["gu"][i].append(g_u)                     # x_blocks[i-1]'s share: _lin_level2_tail
                return L.conv.dgrad(g_u, res=g_skip[i - 1] if g_skip is not None else None)
            g_a2 = L.conv.dgrad(g_u)
            ops.add(g_xb[i - 1], g_a2, out=g_xb[i - 1])
            if g_skip is not None:
                g_a2 = ops.add(g_a2, g_skip[i - 1], out=g_a2)
            return g_a2
        # level 0: the input is e_t = head(event_t); collect its gradient (both sweeps) for the
        # event head's weight gradient.  The forward-sweep BPTT runs first and writes every slice.
        sl = g_e[t * B:(t + 1) * B]
        if first_writer:
            L.conv.dgrad(g_u, out=sl)
        else:
            L.conv.dgrad(g_u, res=sl, out=sl)
        return None
