"""Thin torch-tensor wrappers over the C ABI (include/refid_hip.h).

Tensors are NHWC fp32 on the GPU: shape (N, H, W, C) with stride(-1) == 1; the pixel
pitch may exceed C (channel-slice views of wider buffers).  torch is used here only
for device memory and streams; all arithmetic happens in librefid_hip.so.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, WgradDesc, check, lib

ROLE_FWD, ROLE_DGRAD, ROLE_CONVT, ROLE_CONVT_DGRAD, ROLE_DOWN_DGRAD = range(5)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _nhwc(t, name):
    """(ptr, ld) of an NHWC view; validates layout."""
    if t is None:
        return None, 0
    if t.dtype != torch.float32 or not t.is_cuda:
        raise _lib.RefidHipError(f"{name}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")
    if t.dim() != 4:
        raise _lib.RefidHipError(f"{name}: expected NHWC 4-D tensor, got shape {tuple(t.shape)}")
    n, h, w, c = t.shape
    ld = t.stride(2)
    ok = t.stride(3) == 1 and ld % 4 == 0 and ld >= c
    ok = ok and (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld)
    if not ok or t.data_ptr() % 16 != 0:
        raise _lib.RefidHipError(f"{name}: not a dense-pixel NHWC view (shape {tuple(t.shape)}, "
                                 f"strides {t.stride()}, ptr%16={t.data_ptr() % 16})")
    return t.data_ptr(), ld


def conv_kc(kh, kw, stride, mode=0):
    return lib().refid_conv_kc(kh, kw, stride, mode)


def conv_bn(kh, kw, stride, mode, cout):
    return lib().refid_conv_bn(kh, kw, stride, mode, cout)


def pack_conv_weights(w, role, bn, kc, kh, kw, o, i):
    """Pack a reference-layout weight (OIHW, or IOHW for ConvTranspose2d) for the conv tile."""
    L = lib()
    nfl = L.refid_packed_weight_floats(role, o, i, kh, kw, kc, bn)
    if nfl == 0:
        raise _lib.RefidHipError("pack_conv_weights: bad geometry")
    w = w.contiguous()
    out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    check(L.refid_pack_conv_weights(w.data_ptr(), out.data_ptr(), role, o, i, kh, kw, kc, bn, _stream()),
          "refid_pack_conv_weights")
    return out


def conv2d(in_a, w_packed, out, *, kh, kw, stride=1, pad=0, mode=0, cout, cout_pad, co_base=0,
           in_b=None, bias=None, res=None, mask=None, slope_pre=1.0, slope_post=1.0, slope_mask=1.0):
    """out = mask(post(pre(conv([in_a|in_b]) + bias) + res)); see refid_conv_desc."""
    d = ConvDesc()
    d.in_a, d.ld_a = _nhwc(in_a, "in_a")
    d.c_a = in_a.shape[3]
    if in_b is not None:
        d.in_b, d.ld_b = _nhwc(in_b, "in_b")
        d.c_b = in_b.shape[3]
        if in_b.shape[:3] != in_a.shape[:3]:
            raise _lib.RefidHipError("conv2d: in_a / in_b pixel grids differ")
    d.w_packed = w_packed.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.out, d.ld_out = _nhwc(out, "out")
    d.res, d.ld_res = _nhwc(res, "res")
    d.mask, d.ld_mask = _nhwc(mask, "mask")
    d.n, d.h, d.w = in_a.shape[0], in_a.shape[1], in_a.shape[2]
    if mode == 0:
        d.ho, d.wo = out.shape[1], out.shape[2]
        chan_ok = out.shape[3] == cout
    else:
        d.ho, d.wo = d.h, d.w
        if out.shape[1] != 2 * d.h or out.shape[2] != 2 * d.w:
            raise _lib.RefidHipError("conv2d: mode 1/2 output must be (2h, 2w)")
        chan_ok = out.shape[3] == (cout // 4 if mode == 1 else cout)
    if not chan_ok or out.shape[0] != d.n:
        raise _lib.RefidHipError(f"conv2d: output shape {tuple(out.shape)} inconsistent with cout={cout} mode={mode}")
    for t, nm in ((res, "res"), (mask, "mask")):
        if t is not None and t.shape != out.shape:
            raise _lib.RefidHipError(f"conv2d: {nm} shape {tuple(t.shape)} != out shape {tuple(out.shape)}")
    d.cout, d.cout_pad, d.co_base = cout, cout_pad, co_base
    d.kh, d.kw, d.stride, d.pad, d.mode = kh, kw, stride, pad, mode
    d.slope_pre, d.slope_post, d.slope_mask = slope_pre, slope_post, slope_mask
    check(lib().refid_conv2d(C.byref(d), _stream()), "refid_conv2d")
    return out


_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only scratch buffer per device (split-K slabs).  Stream-ordered re-use."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(g, in_a, dw, *, kh, kw, stride=1, pad=0, in_b=None, db=None, i_base=0, i_total=None):
    """dw (+)= wgrad, db (+)= sum g; dw in the reference layout (c_o, i_total, kh, kw)."""
    d = WgradDesc()
    d.g, d.ld_g = _nhwc(g, "g")
    d.c_o = g.shape[3]
    d.in_a, d.ld_a = _nhwc(in_a, "in_a")
    d.c_a = in_a.shape[3]
    if in_b is not None:
        d.in_b, d.ld_b = _nhwc(in_b, "in_b")
        d.c_b = in_b.shape[3]
    d.n, d.h, d.w = in_a.shape[0], in_a.shape[1], in_a.shape[2]
    d.ho, d.wo = g.shape[1], g.shape[2]
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.i_base = i_base
    d.i_total = i_total if i_total is not None else d.c_a + d.c_b
    if not dw.is_contiguous() or dw.numel() != d.c_o * d.i_total * kh * kw:
        raise _lib.RefidHipError(f"wgrad: dw shape {tuple(dw.shape)} does not match ({d.c_o},{d.i_total},{kh},{kw})")
    d.dw = dw.data_ptr()
    d.db = db.data_ptr() if db is not None else None
    nbytes = lib().refid_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise _lib.RefidHipError("wgrad: unsupported geometry")
    ws = _workspace(nbytes, g.device)
    d.slabs = ws.data_ptr()
    check(lib().refid_conv2d_wgrad(C.byref(d), _stream()), "refid_conv2d_wgrad")


def nchw_to_nhwc(src, c_pad=None):
    """(N,C,H,W) contiguous -> (N,H,W,c_pad) with zero channel padding."""
    src = src.contiguous()
    n, c, h, w = src.shape
    c_pad = c_pad or ((c + 3) // 4) * 4
    dst = torch.empty((n, h, w, c_pad), dtype=torch.float32, device=src.device)
    check(lib().refid_nchw_to_nhwc(src.data_ptr(), dst.data_ptr(), n, c, h, w, c_pad, _stream()), "refid_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src, c, dst, dst_batch_stride=None):
    """First c channels of NHWC src -> NCHW dst (dst may be a (B,T,...) stack slice)."""
    ptr, ld = _nhwc(src, "src")
    n, h, w, _ = src.shape
    if dst_batch_stride is None:
        dst_batch_stride = c * h * w
    check(lib().refid_nhwc_to_nchw(ptr, ld, dst.data_ptr(), dst_batch_stride, n, c, h, w, _stream()),
          "refid_nhwc_to_nchw")
    return dst


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    if not (a.is_contiguous() and b.is_contiguous() and out.is_contiguous()) or a.shape != b.shape:
        raise _lib.RefidHipError("add: contiguous same-shape tensors required")
    check(lib().refid_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "refid_add")
    return out


def act_bwd(g, y, slope, out=None, accumulate=False):
    """out (+)= g * (y > 0 ? 1 : slope)."""
    if out is None:
        out = torch.empty_like(g)
        accumulate = False
    if not (g.is_contiguous() and y.is_contiguous() and out.is_contiguous()) or g.shape != y.shape:
        raise _lib.RefidHipError("act_bwd: contiguous same-shape tensors required")
    check(lib().refid_act_bwd(g.data_ptr(), y.data_ptr(), out.data_ptr(), slope, int(accumulate), g.numel(), _stream()),
          "refid_act_bwd")
    return out
