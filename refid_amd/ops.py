"""Thin torch-tensor wrappers over the C ABI (include/refid_hip.h).

Tensors are NHWC fp32 on the GPU: shape (N, H, W, C) with stride(-1) == 1; the pixel
pitch may exceed C (channel-slice views of wider buffers).  torch is used here only
for device memory and streams; all arithmetic happens in librefid_hip.so.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, WgradDesc, check, lib

ROLE_FWD, ROLE_DGRAD, ROLE_CONVT, ROLE_CONVT_DGRAD, ROLE_DOWN_DGRAD, ROLE_WINO_FWD, ROLE_WINO_DGRAD, ROLE_CONVT_DGRAD_PW = range(8)


# split-K policy for small grids (refid_conv_desc.split_k): REFID_SPLITK = auto (default: by total grid size) |
# sample (by per-sample geometry: a sample's bits do not depend on the batch it is in) | 0 (never)
WINO_SPLIT = {"0": 0, "s": 1}.get(os.environ.get("REFID_SPLITK", os.environ.get("REFID_WINO_SPLITK", "auto"))[:1], 2)

# Winograd x six / x three tile selection (refid_conv_desc.wino_tile): 0 = by problem size, 1 = the 64-channel tile wherever
# cout > 32, 4 = the 32-channel tile everywhere (A/B switches)
WINO_TILE = int(os.environ.get("REFID_WINO_TILE", "0"))

# bench.py's roofline leg: when PROFILE is a list, conv2d()/conv2d_wgrad() bracket each launch with
# events on the launch stream and append (kernel, algorithmic_flops, start_event, end_event).
PROFILE = None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_REQUIRE_CUDA = True       # (host-logic tests run the wrappers on CPU tensors against a recording library handle)


def _nhwc(t, name):
    """(ptr, ld) of an NHWC view; validates layout."""
    if t is None:
        return None, 0
    if t.dtype != torch.float32 or (_REQUIRE_CUDA and not t.is_cuda):
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


def packed_weight_floats(role, bn, kc, kh, kw, o, i):
    return lib().refid_packed_weight_floats(role, o, i, kh, kw, kc, bn)


def pack_conv_weights(w, role, bn, kc, kh, kw, o, i, out=None, oscale=None):
    """Pack a reference-layout weight (OIHW, or IOHW for ConvTranspose2d) for the conv tile.

    oscale: optional per-output-channel factor folded into the packed copy (FWD/DGRAD)."""
    L = lib()
    nfl = L.refid_packed_weight_floats(role, o, i, kh, kw, kc, bn)
    if nfl == 0:
        raise _lib.RefidHipError("pack_conv_weights: bad geometry")
    if not w.is_contiguous():
        raise _lib.RefidHipError("pack_conv_weights: weight must be contiguous")
    if out is None:
        out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    elif out.numel() != nfl:
        raise _lib.RefidHipError("pack_conv_weights: out has the wrong size")
    if oscale is None:
        check(L.refid_pack_conv_weights(w.data_ptr(), out.data_ptr(), role, o, i, kh, kw, kc, bn, _stream()),
              "refid_pack_conv_weights")
    else:
        check(L.refid_pack_conv_weights_scaled(w.data_ptr(), oscale.data_ptr(), out.data_ptr(), role, o, i, kh, kw,
                                               kc, bn, _stream()), "refid_pack_conv_weights_scaled")
    return out


def pack_conv_weights_bf16(w, role, bn, kc, kh, kw, o, i, out=None, oscale=None):
    """bf16 packed weights for conv2d(algo=2); kc = 2 * conv_kc(...)."""
    L = lib()
    n = L.refid_packed_weight_floats(role, o, i, kh, kw, kc, bn)
    if n == 0:
        raise _lib.RefidHipError("pack_conv_weights_bf16: bad geometry")
    if out is None:
        out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    check(L.refid_pack_conv_weights_bf16(w.data_ptr(), oscale.data_ptr() if oscale is not None else None,
                                         out.data_ptr(), role, o, i, kh, kw, kc, bn, _stream()),
          "refid_pack_conv_weights_bf16")
    return out


TERMS_F16X3 = 19          # refid_conv_desc.mfma_terms of the split tile's three-fp16-product form (16 + 3)


def packed_weight_split_bytes(role, bn, kh, kw, o, i, planes, f16=False):
    """f16: two fp16 planes behind a 64-byte header (conv2d algo 4 with terms = TERMS_F16X3); `planes` is then ignored."""
    if f16:
        return lib().refid_packed_weight_split_f16_bytes(role, o, i, kh, kw, bn)
    return lib().refid_packed_weight_split_bytes(role, o, i, kh, kw, bn, planes)


def pack_conv_weights_split(w, role, bn, kh, kw, o, i, planes=3, out=None, oscale=None, f16=False):
    """Split-bf16 packed weights for conv2d(algo=4): `planes` bf16 numbers per weight (3 = exact, 2 = 2^-17); f16: two fp16
    planes of w 2^eW (22 bits), the scale exponent found by a reduction over the tensor and kept in the packing's header."""
    L = lib()
    nb = packed_weight_split_bytes(role, bn, kh, kw, o, i, planes, f16)
    if nb == 0:
        raise _lib.RefidHipError("pack_conv_weights_split: bad geometry")
    if not w.is_contiguous():
        raise _lib.RefidHipError("pack_conv_weights_split: weight must be contiguous")
    if out is None:
        out = torch.empty(nb // 2, dtype=torch.bfloat16, device=w.device)
    elif out.numel() * out.element_size() != nb:
        raise _lib.RefidHipError("pack_conv_weights_split: out has the wrong size")
    if f16:
        check(L.refid_pack_conv_weights_split_f16(w.data_ptr(), oscale.data_ptr() if oscale is not None else None,
                                                  out.data_ptr(), role, o, i, kh, kw, bn, _stream()),
              "refid_pack_conv_weights_split_f16")
        return out
    check(L.refid_pack_conv_weights_split(w.data_ptr(), oscale.data_ptr() if oscale is not None else None,
                                          out.data_ptr(), role, o, i, kh, kw, bn, planes, _stream()),
          "refid_pack_conv_weights_split")
    return out


def packed_weight_wino6_bytes(role, o, i, f16=False):
    """Bytes of the Winograd-domain packing for conv2d(algo=5): three bf16 planes (terms 0 / 6), or -- f16 -- a 64-byte header
    + two fp16 planes (terms 3)."""
    L = lib()
    return L.refid_packed_weight_wino3h_bytes(role, o, i, 64) if f16 else L.refid_packed_weight_wino6_bytes(role, o, i, 64)


def pack_conv_weights_wino6(w, role, o, i, out=None, oscale=None, f16=False):
    """Winograd-domain weights as three bf16 planes for conv2d(algo=5); role = ROLE_WINO_FWD / ROLE_WINO_DGRAD.
    f16: the two-fp16-plane packing of conv2d(algo=5, terms=3) (scaled by a per-tensor power of two kept in its header)."""
    L = lib()
    nb = packed_weight_wino6_bytes(role, o, i, f16)
    if nb == 0:
        raise _lib.RefidHipError("pack_conv_weights_wino6: bad geometry")
    if not w.is_contiguous():
        raise _lib.RefidHipError("pack_conv_weights_wino6: weight must be contiguous")
    if out is None:
        out = torch.empty(nb // 2, dtype=torch.bfloat16, device=w.device)
    elif out.numel() * out.element_size() != nb:
        raise _lib.RefidHipError("pack_conv_weights_wino6: out has the wrong size")
    fn = L.refid_pack_conv_weights_wino3h if f16 else L.refid_pack_conv_weights_wino6
    check(fn(w.data_ptr(), oscale.data_ptr() if oscale is not None else None, out.data_ptr(), role, o, i, 64, _stream()),
          "refid_pack_conv_weights_wino3h" if f16 else "refid_pack_conv_weights_wino6")
    return out


def _pw_extras(pw, out):
    """refid_pw_extras from a dict of tensors / scalars (see include/refid_hip.h); returns (struct, keep-alive list)."""
    x = _lib.PwExtras()
    if pw.get("ln_gamma") is not None:
        x.ln_gamma, x.ln_beta, x.ln_eps = _c(pw["ln_gamma"], "ln_gamma"), _c(pw["ln_beta"], "ln_beta"), pw.get("ln_eps", 1e-6)
        if pw.get("ln_out") is not None:
            x.ln_out, x.ld_ln_out = _nhwc(pw["ln_out"], "ln_out")
    if pw.get("pool") is not None:
        pool = pw["pool"]
        x.pool, x.pool_parts, x.se_c = _c(pool, "pool"), pool.shape[1], pool.shape[2]
        x.hw = pw["hw"]
        x.inv_hw = 1.0 / pw["hw"]
        x.se_w1, x.se_b1, x.se_w2, x.se_b2 = (_c(pw[k], k) for k in ("se_w1", "se_b1", "se_w2", "se_b2"))
        if pw.get("se_s") is not None:
            x.se_m, x.se_z1, x.se_s = _c(pw["se_m"], "se_m"), _c(pw["se_z1"], "se_z1"), _c(pw["se_s"], "se_s")
        if pw.get("xs_out") is not None:
            x.xs_out, x.ld_xs_out = _nhwc(pw["xs_out"], "xs_out")
    if pw.get("res2") is not None:
        if pw["res2"].shape != out.shape:
            raise _lib.RefidHipError("conv2d: res2 shape differs from the output's")
        x.res2, x.ld_res2 = _nhwc(pw["res2"], "res2")
    if pw.get("out2") is not None:
        if pw["out2"].shape != out.shape:
            raise _lib.RefidHipError("conv2d: out2 shape differs from the output's")
        x.out2, x.ld_out2 = _nhwc(pw["out2"], "out2")
    return x


def conv2d(in_a, w_packed, out, *, kh, kw, stride=1, pad=0, mode=0, cout, cout_pad, co_base=0,
           in_b=None, bias=None, res=None, mask=None, slope_pre=1.0, slope_post=1.0, slope_mask=1.0, algo=0, pw=None,
           terms=0, add2=None, out2=None, mask_mode=0):
    """out = mask(post(pre(conv([in_a|in_b]) + bias) + res)); see refid_conv_desc.  pw: dict of the pointwise tile's
    EGACA fusions (refid_pw_extras); terms: algo 4's product count (0 / 6, or 3); algo 3: 0 = fp32 MFMA, 6 = six bf16 products
    (w_packed from pack_conv_weights_split with kh = kw = 1).  add2 / out2: second output out2 = out + add2 (not algo 3)."""
    if DESC_CACHE and pw is None and PROFILE is None:
        return _conv2d_cached(in_a, w_packed, out, kh, kw, stride, pad, mode, cout, cout_pad, co_base, in_b, bias, res, mask,
                              slope_pre, slope_post, slope_mask, algo, terms, add2, out2, mask_mode)
    d = ConvDesc()
    d.mfma_terms = terms
    if pw is not None:
        if algo != 3:
            raise _lib.RefidHipError("conv2d: pw fusions need the pointwise tile (algo 3)")
        pwx = _pw_extras(pw, out)
        d.pw = C.pointer(pwx)
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
    if out2 is not None:
        if add2 is None or add2.shape != out.shape or out2.shape != out.shape:
            raise _lib.RefidHipError("conv2d: out2 needs add2, both of the output's shape")
        d.add2, d.ld_add2 = _nhwc(add2, "add2")
        d.out2, d.ld_out2 = _nhwc(out2, "out2")
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
    d.mask_mode = mask_mode
    d.algo = algo
    d.wino_tile = WINO_TILE
    if WINO_SPLIT and algo in (0, 1, 2, 4, 5):          # (algo 4: no workspace; the policy decides its tile shape)
        d.split_k = WINO_SPLIT
        need = lib().refid_conv_workspace_bytes(C.byref(d))
        if need:
            ws = _workspace(need, in_a.device, "conv")
            d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    if PROFILE is None:
        check(lib().refid_conv2d(C.byref(d), _stream()), "refid_conv2d")
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().refid_conv2d(C.byref(d), _stream()), "refid_conv2d")
    e1.record()
    taps = {0: kh * kw, 1: 1, 2: 16}[mode]
    flops = 2.0 * d.n * d.ho * d.wo * cout * (d.c_a + d.c_b) * taps
    name = "conv_igemm_kernel<" + lib().refid_conv_tile_name(kh, kw, stride, mode, cout).decode() + \
        (", true> [bf16 operands]" if algo == 2 else ">")
    if algo == 1:
        name = "conv_wino_kernel<1, 2>" if cout <= 32 else "conv_wino_kernel<2, 1>"
    if algo == 5:
        name = "conv_wino6_kernel<2>" if cout > 32 else "conv_wino6_kernel<1>"     # 64- / 32-channel workgroup tile
        if terms == 3:
            name = name[:-1] + ", true>"                                           # three fp16 products
    if algo == 4:
        name = "conv_split_kernel<%s>" % ("3, fp16" if terms == TERMS_F16X3 else (terms or 6))
    if algo == 3:
        name = "conv_pw_kernel<1, 8>" if cout <= 32 else ("conv_pw_kernel<2, 4, six>" if terms == 6 else "conv_pw_kernel<2, 4>")
    opix = d.n * out.shape[1] * out.shape[2]
    nbytes = 4.0 * (d.n * d.h * d.w * (d.c_a + d.c_b) + opix * out.shape[3] * (1 + (res is not None) + (mask is not None))
                    + cout * (d.c_a + d.c_b) * taps)
    if out2 is not None:                                   # the second output and its addend are algorithmic bytes too
        nbytes += 8.0 * opix * out.shape[3]
    if pw is not None:                                     # the pointwise tile's fused side outputs / second residual (EGACA)
        for k in ("ln_out", "xs_out", "out2", "res2"):
            if pw.get(k) is not None:
                nbytes += 4.0 * pw[k].numel()
    PROFILE.append((name, flops, e0, e1,
                    (d.n, d.h, d.w, d.c_a, d.c_b, cout, int(res is not None), int(mask is not None), int(bias is not None)),
                    nbytes))
    return out


# ---- descriptor cache (REFID_DESC_CACHE=1; off by default until it has been timed on a GPU box) -------------------------------
# At B=1 a train step is ~4800 launches of ~18 us of GPU time each, and the Python side of ONE conv2d call above -- ~35 ctypes
# field stores, five layout checks, the workspace query -- costs ~16 us: the host is part of what the step waits for (the hipGraph
# replay of the same step is 3.5 ms faster).  Everything in a descriptor except the pointers is a function of the call's scalar
# arguments and of the operands' shapes and strides: the first call with a given signature goes through conv2d's full path with a
# recording library handle, later calls copy the recorded descriptor, refresh the pointers (re-checking their alignment) and the
# workspace, and launch.  tests/test_host_logic.py compares the descriptor BYTES of the two paths call by call.
DESC_CACHE = os.environ.get("REFID_DESC_CACHE", "0") == "1"
_DESC_CACHE = {}
_DESC_PTRS = ("in_a", "in_b", "w_packed", "bias", "out", "res", "mask", "add2", "out2")


def _sig(t):
    return None if t is None else (t.shape, t.stride())


def _conv2d_cached(in_a, w_packed, out, kh, kw, stride, pad, mode, cout, cout_pad, co_base, in_b, bias, res, mask,
                   slope_pre, slope_post, slope_mask, algo, terms, add2, out2, mask_mode):
    global lib, DESC_CACHE
    key = (kh, kw, stride, pad, mode, cout, cout_pad, co_base, slope_pre, slope_post, slope_mask, algo, terms, mask_mode,
           WINO_TILE, WINO_SPLIT, _sig(in_a), _sig(in_b), _sig(out), _sig(res), _sig(mask), _sig(add2), _sig(out2),
           bias is not None, in_a.dtype, out.dtype, in_a.device)
    ent = _DESC_CACHE.get(key)
    if ent is None:
        # first call with this signature: the validating path, with the launch intercepted to keep its descriptor
        real, seen = lib(), []

        class _Recorder:
            def __getattr__(self, name):
                if name == "refid_conv2d":
                    def launch(dref, st):
                        seen.append(ConvDesc.from_buffer_copy(C.string_at(C.addressof(dref._obj), C.sizeof(ConvDesc))))
                        return real.refid_conv2d(dref, st)
                    return launch
                return getattr(real, name)

        keep_lib, keep_flag = lib, DESC_CACHE
        rec = _Recorder()
        lib, DESC_CACHE = (lambda: rec), False
        try:
            conv2d(in_a, w_packed, out, kh=kh, kw=kw, stride=stride, pad=pad, mode=mode, cout=cout, cout_pad=cout_pad,
                   co_base=co_base, in_b=in_b, bias=bias, res=res, mask=mask, slope_pre=slope_pre, slope_post=slope_post,
                   slope_mask=slope_mask, algo=algo, terms=terms, add2=add2, out2=out2, mask_mode=mask_mode)
        finally:
            lib, DESC_CACHE = keep_lib, keep_flag
        _DESC_CACHE[key] = (seen[0], int(seen[0].ws_bytes))
        return out
    d, ws_bytes = ConvDesc.from_buffer_copy(ent[0]), ent[1]     # (a private copy: the recorded descriptor is shared by threads)
    if _REQUIRE_CUDA and not (in_a.is_cuda and out.is_cuda):
        raise _lib.RefidHipError("conv2d: expected CUDA tensors")
    ptrs = (in_a.data_ptr(), None if in_b is None else in_b.data_ptr(), w_packed.data_ptr(),
            None if bias is None else bias.data_ptr(), out.data_ptr(), None if res is None else res.data_ptr(),
            None if mask is None else mask.data_ptr(),
            None if (add2 is None or out2 is None) else add2.data_ptr(),      # (as the validating path: add2 only with out2)
            None if out2 is None else out2.data_ptr())
    for nm, p in zip(_DESC_PTRS, ptrs):
        if p is not None and p % 16 != 0 and nm not in ("w_packed", "bias"):
            raise _lib.RefidHipError(f"{nm}: not a dense-pixel NHWC view (ptr%16={p % 16})")
        setattr(d, nm, p)
    if ws_bytes:
        ws = _workspace(ws_bytes, in_a.device, "conv")     # (per launching stream; grow-only)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    check(lib().refid_conv2d(C.byref(d), _stream()), "refid_conv2d")
    return out


_ws_cache = {}
# Bumped whenever a scratch buffer a captured hipGraph may have baked in (split-K workspaces, persistent weight-gradient
# slabs) is replaced by a larger one: train.py re-captures its graphs when the epoch moved (the old buffer is freed).
ALLOC_EPOCH = 0


def _workspace(nbytes, device, kind="wgrad"):
    """Grow-only scratch buffer per (device, stream, kind): the C ABI never allocates, the caller lends it
    scratch that is private to the launching stream (stream-ordered re-use)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, kind)
    global ALLOC_EPOCH
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            ALLOC_EPOCH += 1
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(g, in_a, dw, *, kh, kw, stride=1, pad=0, in_b=None, db=None, i_base=0, i_total=None, algo=0,
                 phase=0, slabs=None, more=()):
    """phase 0: one-shot.  phase 1/2: partial products into the caller's persistent `slabs`
    (overwrite / add); phase 3: reduce `slabs` into dw/db (see refid_wgrad_desc); phase 4: as 3 with the element-wise
    stage queued until wgrad_finish_flush().  Returns `slabs`."""
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
    d.i_total = i_total if i_total is not None else dw.shape[1]
    d.o_real = dw.shape[0]
    d.algo = algo
    d.phase = phase
    # more: up to 23 further (g, in_a, in_b) triples (refid_wgrad_desc.groups <= 24) -- other time steps of the same conv, added
    # in the same launch
    if more:
        d.groups = 1 + len(more)
        for k, (g2, a2, b2) in enumerate(more):
            pg, ldg = _nhwc(g2, "g (group)")
            pa, lda = _nhwc(a2, "in_a (group)")
            pb, ldb = _nhwc(b2, "in_b (group)") if b2 is not None else (None, d.ld_b)
            if g2.shape != g.shape or a2.shape != in_a.shape or (b2 is None) != (in_b is None) or \
                    (ldg, lda, ldb) != (d.ld_g, d.ld_a, d.ld_b):
                raise _lib.RefidHipError("wgrad: grouped time steps must share shapes, pitches and the source split")
            d.g_more[k], d.in_a_more[k], d.in_b_more[k] = pg, pa, pb
    if not dw.is_contiguous() or dw.dim() != 4 or dw.shape[1] != d.i_total or dw.shape[2:] != (kh, kw) \
            or d.o_real > d.c_o:
        raise _lib.RefidHipError(f"wgrad: dw shape {tuple(dw.shape)} does not match g/in channels "
                                 f"({d.c_o},{d.i_total},{kh},{kw})")
    d.dw = dw.data_ptr()
    d.db = db.data_ptr() if db is not None else None
    nbytes = lib().refid_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise _lib.RefidHipError("wgrad: unsupported geometry")
    if phase == 0:
        ws = _workspace(nbytes, g.device)
    else:
        if slabs is None or (phase == 1 and slabs.numel() * 4 < nbytes):
            # phase 1 overwrites: a larger batch / crop than the first step's simply gets a larger buffer
            if slabs is not None:
                global ALLOC_EPOCH
                ALLOC_EPOCH += 1
            slabs = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.device)
        elif slabs.numel() * 4 < nbytes:
            raise _lib.RefidHipError("wgrad: persistent slab buffer too small for this geometry (phase 2/3 must "
                                     "see the geometry of the phase-1 call)")
        ws = slabs
    d.slabs = ws.data_ptr()
    if PROFILE is None or phase >= 3:
        check(lib().refid_conv2d_wgrad(C.byref(d), _stream()), "refid_conv2d_wgrad")
        return slabs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().refid_conv2d_wgrad(C.byref(d), _stream()), "refid_conv2d_wgrad")
    e1.record()
    ngrp = max(1, d.groups)
    flops = 2.0 * ngrp * d.n * d.ho * d.wo * d.o_real * min(d.c_a + d.c_b, d.i_total) * kh * kw
    nbytes = 4.0 * ngrp * (d.n * d.h * d.w * (d.c_a + d.c_b) + d.n * d.ho * d.wo * d.c_o)
    PROFILE.append((("wgrad_wino_kernel" if algo == 1 else "wgrad_wino24_kernel" if algo == 5 else "wgrad_wino24_down_kernel" if algo == 7 else "wgrad_wino4_kernel" if algo == 6 else "wgrad_wino6_kernel" if algo == 3 else "wgrad_wino_dma_kernel" if algo == 4 else ("wgrad_bf16_kernel" if algo == 2 else
                                                           f"wgrad_kernel<{kh}x{kw}s{stride}>")), flops, e0, e1,
                    (d.n, d.h, d.w, d.c_a, d.c_b, d.c_o, 0, 0, int(db is not None)), nbytes))
    return slabs


_ROWS_KEEP = None          # partial-row buffers of the queued per-channel sums (alive until rows_sum_flush)


def rows_sum_defer():
    """From here to rows_sum_flush() the per-channel parameter-gradient sums of layernorm2d_bwd / dwconv3x3_bwd / colsum are
    queued (refid_rows_sum_defer); the caller must not read those gradients in between."""
    global _ROWS_KEEP
    check(lib().refid_rows_sum_defer(1), "refid_rows_sum_defer")
    _ROWS_KEEP = []


def rows_sum_flush():
    """Issue the queued sums (one launch per ~160, grouped by destination, call order within one) and end the deferral."""
    global _ROWS_KEEP
    if _ROWS_KEEP is None:
        return
    try:
        check(lib().refid_rows_sum_flush(_stream()), "refid_rows_sum_flush")
    finally:
        _ROWS_KEEP = None


def _keep_rows(parts):
    if _ROWS_KEEP is not None:
        _ROWS_KEEP.append(parts)


def wgrad_finish_flush():
    """Issue the element-wise slab-reduction stages queued by conv2d_wgrad(..., phase=4) calls: one launch per kernel
    family on the current stream (refid_wgrad_finish_flush)."""
    check(lib().refid_wgrad_finish_flush(_stream()), "refid_wgrad_finish_flush")


def nchw_to_nhwc(src, c_pad=None, out=None):
    """(N,C,H,W) with dense (C,H,W) planes (any batch stride) -> (N,H,W,c_pad), zero channel padding."""
    n, c, h, w = src.shape
    if src.stride(3) != 1 or src.stride(2) != w or src.stride(1) != h * w:
        raise _lib.RefidHipError("nchw_to_nhwc: per-sample (C,H,W) block must be dense")
    c_pad = c_pad or ((c + 3) // 4) * 4
    dst = out if out is not None else torch.empty((n, h, w, c_pad), dtype=torch.float32, device=src.device)
    check(lib().refid_nchw_to_nhwc(src.data_ptr(), src.stride(0) if n > 1 else c * h * w, dst.data_ptr(), n, c, h, w,
                                   c_pad, _stream()), "refid_nchw_to_nhwc")
    return dst


def nchw_to_nhwc_tb(src, c_pad=None):
    """(B,T,C,H,W) stack with dense (C,H,W) blocks -> (T*B,H,W,c_pad), TIME-major (sample t*B + b), one launch."""
    b, t, c, h, w = src.shape
    if src.stride(4) != 1 or src.stride(3) != w or src.stride(2) != h * w:
        raise _lib.RefidHipError("nchw_to_nhwc_tb: per-(sample, step) (C,H,W) block must be dense")
    c_pad = c_pad or ((c + 3) // 4) * 4
    dst = torch.empty((t * b, h, w, c_pad), dtype=torch.float32, device=src.device)
    check(lib().refid_nchw_to_nhwc_tb(src.data_ptr(), src.stride(0), src.stride(1), dst.data_ptr(), b, t, c, h, w, c_pad,
                                      _stream()), "refid_nchw_to_nhwc_tb")
    return dst


def nhwc_to_nchw_tb(src, c, dst):
    """First c channels of the time-major NHWC stack src (T*B,H,W,ld) -> dst (B,T,c,H,W), one launch."""
    ptr, ld = _nhwc(src, "src")
    b, t = dst.shape[0], dst.shape[1]
    n, h, w, _ = src.shape
    if n != b * t or dst.shape[2:] != (c, h, w) or dst.stride(4) != 1 or dst.stride(3) != w or dst.stride(2) != h * w:
        raise _lib.RefidHipError("nhwc_to_nchw_tb: dst must be (B,T,c,H,W) with dense (c,H,W) blocks matching src")
    check(lib().refid_nhwc_to_nchw_tb(ptr, ld, dst.data_ptr(), dst.stride(0), dst.stride(1), b, t, c, h, w, _stream()),
          "refid_nhwc_to_nchw_tb")
    return dst


def nchw_tsum_to_nhwc(src, c_pad=None):
    """(N,T,C,H,W) contiguous -> sum over T as (N,H,W,c_pad)."""
    n, t, c, h, w = src.shape
    if not src.is_contiguous():
        raise _lib.RefidHipError("nchw_tsum_to_nhwc: contiguous (N,T,C,H,W) tensor required")
    c_pad = c_pad or ((c + 3) // 4) * 4
    dst = torch.empty((n, h, w, c_pad), dtype=torch.float32, device=src.device)
    check(lib().refid_nchw_tsum_to_nhwc(src.data_ptr(), t * c * h * w, c * h * w, t, dst.data_ptr(), n, c, h, w, c_pad,
                                        _stream()), "refid_nchw_tsum_to_nhwc")
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


SUM_MAX = 48


def sum_n(tensors, out=None):
    """out = tensors[0] + tensors[1] + ... (index order; any number of same-shape contiguous tensors; out may be
    tensors[0]).  More than SUM_MAX inputs are summed in passes."""
    ts = list(tensors)
    if not ts:
        raise _lib.RefidHipError("sum_n: no inputs")
    if out is None:
        out = torch.empty_like(ts[0])
    for t in ts + [out]:
        if not t.is_contiguous() or t.shape != ts[0].shape or t.dtype != torch.float32:
            raise _lib.RefidHipError("sum_n: contiguous same-shape float32 tensors required")
    while True:
        part, ts = ts[:SUM_MAX], ts[SUM_MAX:]
        ptrs = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
        check(lib().refid_sum_n(ptrs, len(part), out.data_ptr(), out.numel(), _stream()), "refid_sum_n")
        if not ts:
            return out
        ts = [out] + ts


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


# ---------------------------------------------------------------------------------------------
# EGACA pieces / train-step tail.  Plain contiguous NHWC tensors unless a pitch is mentioned.
# ---------------------------------------------------------------------------------------------
def _c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise _lib.RefidHipError(f"{name}: contiguous CUDA float32 tensor required")
    return t.data_ptr()


def layernorm2d_fwd(x, w, b, out=None, eps=1e-6):
    px, ld = _nhwc(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    po, ldo = _nhwc(out, "out")
    npix = x.shape[0] * x.shape[1] * x.shape[2]
    check(lib().refid_layernorm2d_fwd(px, ld, _c(w, "w"), _c(b, "b"), po, ldo, npix, x.shape[3], eps, _stream()),
          "refid_layernorm2d_fwd")
    return out


def layernorm2d_bwd(g, x, w, gx, dw, db, accumulate=False, res=None, eps=1e-6):
    """gx = (res or 0) + LayerNorm2d backward; accumulate=True is res=gx (in place)."""
    pg, ldg = _nhwc(g, "g")
    px, ldx = _nhwc(x, "x")
    pgx, ldgx = _nhwc(gx, "gx")
    if accumulate and res is None:
        res = gx
    pr, ldr = _nhwc(res, "res") if res is not None else (None, 0)
    npix = x.shape[0] * x.shape[1] * x.shape[2]
    c = x.shape[3]
    nb = lib().refid_layernorm2d_bwd_parts(npix, c)
    if nb <= 0:
        raise _lib.RefidHipError(f"layernorm2d_bwd: unsupported channel count {c}")
    parts = torch.empty((nb, 2 * c), dtype=torch.float32, device=x.device)      # fixed-order partial sums of dw / db
    check(lib().refid_layernorm2d_bwd(pg, ldg, px, ldx, _c(w, "w"), pgx, ldgx, pr, ldr, _c(dw, "dw"),
                                      _c(db, "db"), parts.data_ptr(), npix, c, eps, _stream()), "refid_layernorm2d_bwd")
    _keep_rows(parts)
    return gx


def dwconv_pool_parts(h, w, c):
    return lib().refid_dwconv_pool_parts(h, w, c)


def dwconv3x3_gelu_fwd(x, w, b, want_pool=False):
    """Returns (pre, act) or (pre, act, pool_parts) with pool_parts (n, parts, c) per-workgroup sums of act."""
    px, ld = _nhwc(x, "x")
    n, h, wd, c = x.shape
    pre = torch.empty((n, h, wd, c), dtype=torch.float32, device=x.device)
    act = torch.empty_like(pre)
    pool = None
    if want_pool:
        parts = lib().refid_dwconv_pool_parts(h, wd, c)
        if parts <= 0:
            raise _lib.RefidHipError(f"dwconv3x3: unsupported channel count {c}")
        pool = torch.empty((n, parts, c), dtype=torch.float32, device=x.device)
    check(lib().refid_dwconv3x3_gelu_fwd(px, ld, _c(w, "w"), _c(b, "b"), pre.data_ptr(), act.data_ptr(),
                                         pool.data_ptr() if pool is not None else None, n, h, wd, c, _stream()),
          "refid_dwconv3x3_gelu_fwd")
    return (pre, act, pool) if want_pool else (pre, act)


def dwconv3x3_bwd(gd, x, w, dw, db):
    px, ld = _nhwc(x, "x")
    n, h, wd, c = x.shape
    gin = torch.empty((n, h, wd, c), dtype=torch.float32, device=x.device)
    nparts = lib().refid_dwconv3x3_bwd_parts(h, wd, c)
    if nparts <= 0:
        raise _lib.RefidHipError(f"dwconv3x3_bwd: unsupported channel count {c}")
    parts = torch.empty((n * nparts, 10 * c), dtype=torch.float32, device=x.device)
    check(lib().refid_dwconv3x3_bwd(_c(gd, "gd"), px, ld, _c(w, "w"), gin.data_ptr(), _c(dw, "dw"), _c(db, "db"),
                                    parts.data_ptr(), n, h, wd, c, _stream()), "refid_dwconv3x3_bwd")
    _keep_rows(parts)
    return gin


def se_fwd(pool, inv_hw, w1, b1, w2, b2):
    """pool: (n, parts, c) partial sums (or (n, c))."""
    if pool.dim() == 2:
        pool = pool.unsqueeze(1)
    n, parts, c = pool.shape
    m = torch.empty((n, c), dtype=torch.float32, device=pool.device)
    z1 = torch.empty((n, c // 2), dtype=torch.float32, device=pool.device)
    s = torch.empty_like(m)
    check(lib().refid_se_fwd(_c(pool, "pool"), parts, inv_hw, _c(w1, "w1"), _c(b1, "b1"), _c(w2, "w2"), _c(b2, "b2"),
                             m.data_ptr(), z1.data_ptr(), s.data_ptr(), n, c, _stream()), "refid_se_fwd")
    return m, z1, s


def se_bwd(gs, s, z1, m, w1, w2, dw1, db1, dw2, db2):
    n, c = gs.shape
    gm = torch.empty_like(gs)
    scratch = torch.empty((n, c + c // 2), dtype=torch.float32, device=gs.device)
    check(lib().refid_se_bwd(_c(gs, "gs"), _c(s, "s"), _c(z1, "z1"), _c(m, "m"), _c(w1, "w1"), _c(w2, "w2"),
                             gm.data_ptr(), _c(dw1, "dw1"), _c(db1, "db1"), _c(dw2, "dw2"), _c(db2, "db2"),
                             scratch.data_ptr(), n, c, _stream()), "refid_se_bwd")
    return gm


def scale_cat(xi, xe, s):
    n, h, w, c = xi.shape
    out = torch.empty((n, h, w, 2 * c), dtype=torch.float32, device=xi.device)
    check(lib().refid_scale_cat(_c(xi, "xi"), _c(xe, "xe"), _c(s, "s"), out.data_ptr(), n, h * w, c, _stream()),
          "refid_scale_cat")
    return out


def egaca_gs_reduce(gxs, xi, xe):
    n, h, w, c = xi.shape
    gs = torch.empty((n, c), dtype=torch.float32, device=xi.device)
    nb = lib().refid_egaca_gs_reduce_parts(h * w, c)
    if nb <= 0:
        raise _lib.RefidHipError(f"egaca_gs_reduce: unsupported channel count {c}")
    parts = torch.empty((n, nb, c), dtype=torch.float32, device=xi.device)
    check(lib().refid_egaca_gs_reduce(_c(gxs, "gxs"), _c(xi, "xi"), _c(xe, "xe"), gs.data_ptr(), parts.data_ptr(), n,
                                      h * w, c, _stream()), "refid_egaca_gs_reduce")
    return gs


def egaca_bwd_elem(gxs, s, gm, dwe, gxi, accumulate_xi):
    n, h, w, c = dwe.shape
    gdwe = torch.empty_like(dwe)
    check(lib().refid_egaca_bwd_elem(_c(gxs, "gxs"), _c(s, "s"), _c(gm, "gm"), 1.0 / (h * w), _c(dwe, "dwe"),
                                     gdwe.data_ptr(), _c(gxi, "gxi"), int(accumulate_xi), n, h * w, c, _stream()),
          "refid_egaca_bwd_elem")
    return gdwe


def gelu_fwd(x):
    out = torch.empty_like(x)
    check(lib().refid_gelu_fwd(_c(x, "x"), out.data_ptr(), x.numel(), _stream()), "refid_gelu_fwd")
    return out


def gelu_bwd(g, x, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(lib().refid_gelu_bwd(_c(g, "g"), _c(x, "x"), out.data_ptr(), x.numel(), _stream()), "refid_gelu_bwd")
    return out


def colsum(g, db):
    pg, ld = _nhwc(g, "g")
    npix = g.shape[0] * g.shape[1] * g.shape[2]
    nb = lib().refid_colsum_parts(npix, g.shape[3])
    if nb <= 0:
        raise _lib.RefidHipError(f"colsum: unsupported channel count {g.shape[3]}")
    parts = torch.empty((nb, g.shape[3]), dtype=torch.float32, device=g.device)
    check(lib().refid_colsum(pg, ld, _c(db, "db"), parts.data_ptr(), npix, g.shape[3], _stream()), "refid_colsum")
    _keep_rows(parts)


class PackPlan:
    """All weight packings of a model as ONE launch (refid_pack_batch).  add_* mirror pack_conv_weights[_bf16 / _split /
    _wino6] and mul_vec with preallocated outputs; the tensors must stay alive and in place (parameter arena, packed
    buffers).  run() after every optimiser step."""

    def __init__(self, device):
        self.device = device
        self.esz = lib().refid_pack_entry_bytes()
        self.items = []
        self.keep = []
        self.table = None
        self.nblocks = 0

    def _add(self, kind, w, oscale, dst, role=0, o=0, i=0, kh=1, kw=1, kc=8, bn=32, planes=0):
        if self.table is not None:
            raise _lib.RefidHipError("PackPlan: already built")
        for t in (w, oscale, dst):
            if t is not None and not t.is_contiguous():
                raise _lib.RefidHipError("PackPlan: tensors must be contiguous")
        self.items.append((kind, w, oscale, dst, role, o, i, kh, kw, kc, bn, planes))
        self.keep += [w, oscale, dst]

    def add_pack(self, w, role, bn, kc, kh, kw, o, i, out, oscale=None, bf16=False):
        self._add(0, w, oscale, out, role, o, i, kh, kw, kc, bn, 1 if bf16 else 0)

    def add_split(self, w, role, bn, kh, kw, o, i, planes, out, oscale=None, f16=False):
        if f16:
            self._add(6, w, oscale, out, role, o, i, kh, kw, 8, bn, 2)
        else:
            self._add(2 if kh == 1 and kw == 1 else 1, w, oscale, out, role, o, i, kh, kw, 8, bn, planes)

    def add_wino6(self, w, role, o, i, out, oscale=None, f16=False):
        self._add(5 if f16 else 3, w, oscale, out, role, o, i, 3, 3, 16, 64, 2 if f16 else 3)

    def add_mul_vec(self, a, b, out):
        self._add(4, a, b, out, o=a.numel())

    def build(self):
        L = lib()
        buf = (C.c_char * (self.esz * len(self.items)))()
        blk = 0
        for n, (kind, w, oscale, dst, role, o, i, kh, kw, kc, bn, planes) in enumerate(self.items):
            nb = L.refid_pack_entry_fill(C.addressof(buf) + n * self.esz, kind, w.data_ptr(),
                                         oscale.data_ptr() if oscale is not None else None, dst.data_ptr(), role, o, i, kh, kw,
                                         kc, bn, planes, blk)
            if nb <= 0:
                raise _lib.RefidHipError(f"refid_pack_entry_fill (record {n}, kind {kind}): " + L.refid_last_error().decode())
            blk += nb
        # every record filled, first blocks consecutive from 0 (the kernel's binary search relies on it)
        if L.refid_pack_table_check(C.addressof(buf), len(self.items)) != blk:
            raise _lib.RefidHipError("refid_pack_table_check: " + L.refid_last_error().decode())
        self.nblocks = blk
        self.table = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).to(self.device)
        return self

    def run(self):
        if any(it[0] in (5, 6) for it in self.items):       # the fp16 packings' scale exponents (max |w| per tensor) first
            check(lib().refid_pack_batch_prepass(self.table.data_ptr(), len(self.items), _stream()), "refid_pack_batch_prepass")
        check(lib().refid_pack_batch(self.table.data_ptr(), len(self.items), self.nblocks, _stream()), "refid_pack_batch")


def mul_vec(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib().refid_mul_vec(_c(a, "a"), _c(b, "b"), out.data_ptr(), a.numel(), _stream()), "refid_mul_vec")
    return out


def fold_back(w, b, scale, gw_folded, gb_folded, gw, gb, dscale):
    """Un-fold the gradient of (scale[r]*W[r,:], scale[r]*b[r]) accumulated in gw_folded/gb_folded:
    dscale[r] += <W[r,:],Gf[r,:]> + b[r]*gbf[r];  gw[r,:] += scale[r]*Gf[r,:];  gb[r] += scale[r]*gbf[r]."""
    rows = scale.numel()
    k = w.numel() // rows
    check(lib().refid_fold_back(_c(w, "w"), _c(b, "b"), _c(scale, "scale"), _c(gw_folded, "gw_folded"),
                                _c(gb_folded, "gb_folded"), _c(gw, "gw"), _c(gb, "gb"), _c(dscale, "dscale"), rows, k,
                                _stream()), "refid_fold_back")


def charbonnier(pred, gt, grad=None, eps=1e-12, grad_scale=None):
    """Returns the device double holding sum sqrt((pred-gt)^2+eps); grad (optional) gets d/dpred of the MEAN.
    Two-stage, deterministic: buf[0] = the sum, buf[1:] = the per-workgroup partials."""
    n = pred.numel()
    buf = torch.empty(1 + lib().refid_charbonnier_parts(n), dtype=torch.float64, device=pred.device)
    if grad_scale is None:
        grad_scale = 1.0 / n
    check(lib().refid_charbonnier(_c(pred, "pred"), _c(gt, "gt"), grad.data_ptr() if grad is not None else None,
                                  buf.data_ptr(), buf[1:].data_ptr(), n, eps, grad_scale, _stream()), "refid_charbonnier")
    return buf[:1]


def psnr_loss(pred, gt, grad=None, weight=1.0):
    """PSNRLoss (losses.py:95-120) of (B, ...) tensors; returns the 1-element double device tensor holding the loss."""
    nb = pred.shape[0]
    per = pred.numel() // nb
    buf = torch.empty(nb + 1 + lib().refid_psnr_loss_parts(nb, per), dtype=torch.float64, device=pred.device)
    check(lib().refid_psnr_loss(_c(pred, "pred"), _c(gt, "gt"), grad.data_ptr() if grad is not None else None,
                                buf.data_ptr(), buf[nb:].data_ptr(), buf[nb + 1:].data_ptr(), nb, per, weight, _stream()),
          "refid_psnr_loss")
    return buf[nb:nb + 1]


SQNORM_WORDS = 2049


def grad_sqnorm(flat_g, out=None):
    """out[0] = sum g^2; `out` holds SQNORM_WORDS doubles (result + scratch for the block partials)."""
    if out is None:
        out = torch.empty(SQNORM_WORDS, dtype=torch.float64, device=flat_g.device)
    elif out.numel() < SQNORM_WORDS:
        raise _lib.RefidHipError("grad_sqnorm: out needs SQNORM_WORDS doubles")
    check(lib().refid_grad_sqnorm(_c(flat_g, "g"), out.data_ptr(), flat_g.numel(), _stream()), "refid_grad_sqnorm")
    return out


def clip_adamw(p, g, m, v, sqnorm, *, max_norm, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    check(lib().refid_clip_adamw(_c(p, "p"), _c(g, "g"), _c(m, "m"), _c(v, "v"),
                                 sqnorm.data_ptr() if sqnorm is not None else None, max_norm, grad_scale, lr,
                                 betas[0], betas[1], eps, weight_decay, step, p.numel(), _stream()),
          "refid_clip_adamw")


def clip_adamw_dev(p, g, m, v, sqnorm, hyper, *, max_norm, betas, eps, weight_decay, grad_scale=1.0):
    """clip_adamw with (lr, 1 - beta1^t, sqrt(1 - beta2^t)) read from the 3-float device tensor `hyper`."""
    check(lib().refid_clip_adamw_dev(_c(p, "p"), _c(g, "g"), _c(m, "m"), _c(v, "v"),
                                     sqnorm.data_ptr() if sqnorm is not None else None, max_norm, grad_scale,
                                     _c(hyper, "hyper"), betas[0], betas[1], eps, weight_decay, p.numel(), _stream()),
          "refid_clip_adamw_dev")


# ---------------------------------------------------------------------------------------------
# SingleMultiConnectEVHINet non-GEMM pieces (csrc/evhinet.hip)
# ---------------------------------------------------------------------------------------------
def hin_lrelu_fwd(x, gamma, beta, slope=0.2, eps=1e-5):
    """LeakyReLU([InstanceNorm(x[..., :ch]) * gamma + beta | x[..., ch:]]), ch = len(gamma) (None: plain LeakyReLU).
    Returns (out, stats) with stats (n, 2, ch) = mean / rstd (None when ch == 0)."""
    px, ld = _nhwc(x, "x")
    n, h, w, c = x.shape
    ch = 0 if gamma is None else gamma.numel()
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    stats = parts = None
    if ch:
        stats = torch.empty((n, 2, ch), dtype=torch.float32, device=x.device)
        parts = torch.empty((n, lib().refid_hin_parts(h * w), 2, ch), dtype=torch.float32, device=x.device)
    check(lib().refid_hin_lrelu_fwd(px, ld, _c(gamma, "gamma") if ch else None, _c(beta, "beta") if ch else None,
                                    out.data_ptr(), c, stats.data_ptr() if ch else None,
                                    parts.data_ptr() if ch else None, n, h * w, c, ch, eps, slope, _stream()),
          "refid_hin_lrelu_fwd")
    return out, stats


def hin_lrelu_bwd(g, out, x, gamma, stats, dgamma, dbeta, slope=0.2):
    """Gradient w.r.t. x of hin_lrelu_fwd; dgamma / dbeta accumulate."""
    pg, ldg = _nhwc(g, "g")
    po, ldo = _nhwc(out, "out")
    n, h, w, c = out.shape
    ch = 0 if gamma is None else gamma.numel()
    gx = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
    px, ldx = _nhwc(x, "x") if ch else (None, 0)
    sums = parts = None
    if ch:
        sums = torch.empty((n, 2, ch), dtype=torch.float32, device=g.device)
        parts = torch.empty((n, lib().refid_hin_parts(h * w), 2, ch), dtype=torch.float32, device=g.device)
    check(lib().refid_hin_lrelu_bwd(pg, ldg, po, ldo, px, ldx, _c(gamma, "gamma") if ch else None,
                                    _c(stats, "stats") if ch else None, gx.data_ptr(), c,
                                    _c(dgamma, "dgamma") if ch else None, _c(dbeta, "dbeta") if ch else None,
                                    sums.data_ptr() if ch else None, parts.data_ptr() if ch else None, n, h * w, c, ch,
                                    slope, _stream()), "refid_hin_lrelu_bwd")
    return gx


def fac_fwd(feat, filt):
    """FAC_bias: feat * filt[..., :C] + filt[..., C:]."""
    pf, ldf = _nhwc(feat, "feat")
    pi, ldi = _nhwc(filt, "filt")
    n, h, w, c = feat.shape
    if filt.shape[3] != 2 * c:
        raise _lib.RefidHipError(f"fac_fwd: filter has {filt.shape[3]} channels, expected {2 * c}")
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=feat.device)
    check(lib().refid_fac_fwd(pf, ldf, pi, ldi, out.data_ptr(), c, n * h * w, c, _stream()), "refid_fac_fwd")
    return out


def fac_bwd(g, feat, filt):
    """Returns (g_feat, g_filt)."""
    pg, ldg = _nhwc(g, "g")
    pf, ldf = _nhwc(feat, "feat")
    pi, ldi = _nhwc(filt, "filt")
    n, h, w, c = feat.shape
    gfeat = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
    gfilt = torch.empty((n, h, w, 2 * c), dtype=torch.float32, device=g.device)
    check(lib().refid_fac_bwd(pg, ldg, pf, ldf, pi, ldi, gfeat.data_ptr(), c, gfilt.data_ptr(), 2 * c, n * h * w, c,
                              _stream()), "refid_fac_bwd")
    return gfeat, gfilt
