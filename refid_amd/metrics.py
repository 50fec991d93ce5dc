"""Validation tail on the GPU: tensor2img quantisation + calculate_psnr.

Reference: basicsr/utils/img_util.py:90-117 (clamp to [0,1], x255, round -> uint8; the RGB->BGR
flip is PSNR-invariant) and basicsr/metrics/psnr_ssim.py:48-63 (float64 MSE over H x W x 3,
20*log10(255/sqrt(mse)), inf when identical).  One fused kernel + a fixed-order finish (deterministic, no atomics),
no host round trip per frame."""
import ctypes as C
import math

import torch

from ._lib import RefidHipError, check, lib


def calculate_psnr_frames(pred, gt):
    """pred, gt: (..., 3, H, W) float32 CUDA tensors in [0,1] range; returns a list of per-frame PSNRs."""
    if pred.shape != gt.shape or pred.dim() < 3:
        raise AssertionError(f"Image shapes are differnet: {tuple(pred.shape)}, {tuple(gt.shape)}.")
    if not (pred.is_cuda and gt.is_cuda) or pred.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RefidHipError("calculate_psnr_frames: float32 CUDA tensors required")
    pred, gt = pred.contiguous(), gt.contiguous()
    fe = pred.shape[-1] * pred.shape[-2] * pred.shape[-3]
    nf = pred.numel() // fe
    buf = torch.empty(nf + lib().refid_sqerr_u8_parts(nf, fe), dtype=torch.float64, device=pred.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().refid_sqerr_u8(pred.data_ptr(), gt.data_ptr(), nf, fe, buf.data_ptr(), buf[nf:].data_ptr(), st), "refid_sqerr_u8")
    out = []
    for v in buf[:nf].tolist():                         # one device->host copy for all frames
        mse = v / fe
        out.append(float("inf") if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse)))
    return out


def calculate_ssim_frames(pred, gt):
    """The reference's calculate_ssim (3-D Gaussian variant, metrics/psnr_ssim.py:135-182,225-303) on the
    uint8-quantised frames; pred, gt: (..., 3, H, W) float32 CUDA tensors.  Per-frame list."""
    if pred.shape != gt.shape or pred.dim() < 3 or pred.shape[-3] != 3:
        raise AssertionError(f"Image shapes are differnet: {tuple(pred.shape)}, {tuple(gt.shape)}.")
    if not (pred.is_cuda and gt.is_cuda) or pred.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RefidHipError("calculate_ssim_frames: float32 CUDA tensors required")
    pred, gt = pred.contiguous(), gt.contiguous()
    h, w = pred.shape[-2], pred.shape[-1]
    nf = pred.numel() // (3 * h * w)
    buf = torch.empty(nf + lib().refid_ssim3d_u8_parts(nf, h, w), dtype=torch.float64, device=pred.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().refid_ssim3d_u8(pred.data_ptr(), gt.data_ptr(), nf, h, w, buf.data_ptr(), buf[nf:].data_ptr(), st), "refid_ssim3d_u8")
    return [v / (3 * h * w) for v in buf[:nf].tolist()]


def split_deblur_interp(psnrs, m, n):
    """Mean PSNR over 'interpolation' frames (index in [m, m+n)) and the rest ('deblur'), as the
    reference's validation does (twoImage_event_recurrent_model.py:426-507).  psnrs: per-frame list of
    ONE sample (T = 2m+n frames)."""
    interp = [p for i, p in enumerate(psnrs) if m <= i < m + n]
    deblur = [p for i, p in enumerate(psnrs) if not (m <= i < m + n)]
    mean = lambda xs: sum(xs) / len(xs) if xs else float("nan")     # noqa: E731
    return mean(deblur), mean(interp)
