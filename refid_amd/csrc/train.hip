// Train-step tail of the REFID hot path: Charbonnier loss
// forward+backward in one pass, global gradient norm, and clip + AdamW fused over the
// FLAT parameter / gradient arenas (one launch for all 183 tensors instead of 183 x k
// torch launches).  All HBM-bound streaming kernels, 16 bytes per lane per access.
//
// Reference: losses/losses.py:28-30,143-173 (CharbonnierLoss, eps=1e-12, mean);
// twoImage_event_recurrent_model.py:303-306 (backward, clip_grad_norm_(0.01), AdamW step).
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_double(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;       // valid on thread 0
}

// loss_sum[block] = sum sqrt(d^2 + eps) ; grad = d / sqrt(d^2 + eps) * gscale     (d = pred - gt)
__global__ __launch_bounds__(256) void charbonnier_kernel(const f32x4* __restrict__ pred, const f32x4* __restrict__ gt,
                                                         f32x4* __restrict__ grad, double* __restrict__ loss_sum,
                                                         float eps, float gscale, long long n4) {
    __shared__ double sh[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 d = pred[i] - gt[i];
        f32x4 g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float r = sqrtf(d[k] * d[k] + eps);
            acc += r;
            g[k] = d[k] / r * gscale;
        }
        if (grad) grad[i] = g;
    }
    const double t = block_sum_double((double)acc, sh);
    if (threadIdx.x == 0) loss_sum[blockIdx.x] = t;        // per-block partial; summed in index order (sum_rows_kernel)
}

// PSNRLoss (losses.py:95-120, toY = False): loss = w * 10/ln10 * mean_b log(mse_b + 1e-8), mse_b over (C,H,W).
// Pass 1: partial sums of (pred - gt)^2 of sample b (grid.y = b), finished in index order into sq[b].  Pass 2: grad = (pred - gt) * 2 w scale / (n_b B (mse_b + 1e-8));
// block 0 also writes the loss itself.
__global__ __launch_bounds__(256) void psnr_sq_kernel(const f32x4* __restrict__ pred, const f32x4* __restrict__ gt,
                                                     double* __restrict__ sq, long long per4) {
    __shared__ double sh[4];
    const long long base = (long long)blockIdx.y * per4;
    float acc = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < per4; i += (long long)gridDim.x * 256) {
        const f32x4 d = pred[base + i] - gt[base + i];
        acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    const double t = block_sum_double((double)acc, sh);
    if (threadIdx.x == 0) sq[(long long)blockIdx.y * gridDim.x + blockIdx.x] = t;     // partial [sample][block]
}

__global__ __launch_bounds__(256) void psnr_grad_kernel(const f32x4* __restrict__ pred, const f32x4* __restrict__ gt,
                                                       f32x4* __restrict__ grad, const double* __restrict__ sq,
                                                       double* __restrict__ loss, long long per4, int nb, float weight) {
    const double scale = 10.0 / log(10.0);
    const double per = (double)per4 * 4.0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        double l = 0.0;
        for (int b = 0; b < nb; ++b) l += log(sq[b] / per + 1e-8);
        *loss = weight * scale * l / nb;
    }
    if (!grad) return;
    const float coef = (float)(weight * scale * 2.0 / (per * nb) / (sq[blockIdx.y] / per + 1e-8));
    const long long base = (long long)blockIdx.y * per4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < per4; i += (long long)gridDim.x * 256)
        grad[base + i] = (pred[base + i] - gt[base + i]) * coef;
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const f32x4* __restrict__ g, double* __restrict__ out, long long n4) {
    __shared__ double sh[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = g[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const double t = block_sum_double((double)acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = t;             // per-block partial (deterministic 2nd stage)
}

// fixed-order sum of per-block partials, one row of n partials per workgroup: every rank of a data-parallel job must
// derive the SAME clip coefficient from the same all-reduced gradients, and a logged loss / PSNR / SSIM must not depend
// on the arrival order of workgroups -- so no floating-point atomics anywhere
__global__ __launch_bounds__(256) void sum_rows_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    __shared__ double sh[256];
    part += (long long)blockIdx.x * n;
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// torch.nn.utils.clip_grad_norm_(max_norm) + torch.optim.AdamW single step.
__global__ __launch_bounds__(256) void clip_adamw_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g,
                                                        f32x4* __restrict__ m, f32x4* __restrict__ v,
                                                        const double* __restrict__ sqnorm, float max_norm,
                                                        float gscale, float lr, float b1, float b2, float eps,
                                                        float wd, float bc1, float bc2_sqrt, long long n4,
                                                        const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }     // per-step scalars of a replayed hipGraph
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float norm = (float)sqrt(*sqnorm) * gscale;
        const float c = max_norm / (norm + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    coef *= gscale;
    const float step = lr / bc1;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 pv = p[i], mv = m[i], vv = v[i];
        const f32x4 gv = g[i] * coef;
        pv *= (1.f - lr * wd);
        mv = mv * b1 + gv * (1.f - b1);
        vv = vv * b2 + gv * gv * (1.f - b2);
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] -= step * mv[k] / (sqrtf(vv[k]) / bc2_sqrt + eps);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

int nblocks(long long n4) {
    long long b = (n4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

int refid_launch_sum_rows_f64(const double* part, int rows, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(sum_rows_kernel, dim3(rows), dim3(256), 0, st, part, n, out);
    REFID_LAUNCH_CHECK("sum_rows_f64");
    return 0;
}

extern "C" int refid_charbonnier_parts(long long count) { return count > 0 ? nblocks(count / 4) : 0; }

extern "C" int refid_charbonnier(const float* pred, const float* gt, float* grad, double* loss_sum, double* parts,
                                 long long count, float eps, float grad_scale, void* stream) {
    REFID_CHECK(pred && gt && loss_sum && parts && count > 0 && count % 4 == 0, "charbonnier: bad arguments (count=%lld)", count);
    hipStream_t st = (hipStream_t)stream;
    const int nb = nblocks(count / 4);
    hipLaunchKernelGGL(charbonnier_kernel, dim3(nb), dim3(256), 0, st, (const f32x4*)pred,
                       (const f32x4*)gt, (f32x4*)grad, parts, eps, grad_scale, count / 4);
    REFID_LAUNCH_CHECK("charbonnier");
    return refid_launch_sum_rows_f64(parts, 1, nb, loss_sum, st);
}

static int psnr_blocks(long long per_sample) {
    const int nb = nblocks(per_sample / 4);
    return nb > 256 ? 256 : nb;
}

extern "C" int refid_psnr_loss_parts(int n_samples, long long per_sample) {
    return (n_samples > 0 && per_sample > 0) ? n_samples * psnr_blocks(per_sample) : 0;
}

extern "C" int refid_psnr_loss(const float* pred, const float* gt, float* grad, double* sq, double* loss, double* parts,
                               int n_samples, long long per_sample, float weight, void* stream) {
    REFID_CHECK(pred && gt && sq && loss && parts && n_samples > 0 && per_sample > 0 && per_sample % 4 == 0,
                "psnr_loss: bad arguments (per_sample=%lld)", per_sample);
    hipStream_t st = (hipStream_t)stream;
    const int nb = psnr_blocks(per_sample);
    hipLaunchKernelGGL(psnr_sq_kernel, dim3(nb, n_samples), dim3(256), 0, st, (const f32x4*)pred, (const f32x4*)gt, parts,
                       per_sample / 4);
    REFID_LAUNCH_CHECK("psnr_loss/sq");
    if (int rc = refid_launch_sum_rows_f64(parts, n_samples, nb, sq, st)) return rc;
    hipLaunchKernelGGL(psnr_grad_kernel, dim3(nb, n_samples), dim3(256), 0, st, (const f32x4*)pred, (const f32x4*)gt,
                       (f32x4*)grad, sq, loss, per_sample / 4, n_samples, weight);
    REFID_LAUNCH_CHECK("psnr_loss/grad");
    return 0;
}

extern "C" int refid_grad_sqnorm(const float* g, double* out, long long count, void* stream) {
    // out: REFID_SQNORM_WORDS doubles; out[0] receives the result, out[1..] is scratch for the partials
    REFID_CHECK(g && out && count > 0 && count % 4 == 0, "grad_sqnorm: bad arguments (count=%lld)", count);
    hipStream_t st = (hipStream_t)stream;
    const int nb = nblocks(count / 4);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nb), dim3(256), 0, st, (const f32x4*)g, out + 1, count / 4);
    REFID_LAUNCH_CHECK("grad_sqnorm");
    return refid_launch_sum_rows_f64(out + 1, 1, nb, out, st);
}

extern "C" int refid_clip_adamw(float* p, const float* g, float* m, float* v, const double* sqnorm, float max_norm,
                                float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int step, long long count, void* stream) {
    REFID_CHECK(p && g && m && v && count > 0 && count % 4 == 0 && step >= 1, "clip_adamw: bad arguments");
    REFID_CHECK(max_norm <= 0.f || sqnorm != nullptr, "clip_adamw: clipping needs the squared norm");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(clip_adamw_kernel, dim3(nblocks(count / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)p,
                       (const f32x4*)g, (f32x4*)m, (f32x4*)v, sqnorm, max_norm, grad_scale, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), count / 4, (const float*)nullptr);
    REFID_LAUNCH_CHECK("clip_adamw");
    return 0;
}

extern "C" int refid_clip_adamw_dev(float* p, const float* g, float* m, float* v, const double* sqnorm, float max_norm,
                                    float grad_scale, const float* hyper, float beta1, float beta2, float eps,
                                    float weight_decay, long long count, void* stream) {
    REFID_CHECK(p && g && m && v && hyper && count > 0 && count % 4 == 0, "clip_adamw_dev: bad arguments");
    REFID_CHECK(max_norm <= 0.f || sqnorm != nullptr, "clip_adamw_dev: clipping needs the squared norm");
    hipLaunchKernelGGL(clip_adamw_kernel, dim3(nblocks(count / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)p,
                       (const f32x4*)g, (f32x4*)m, (f32x4*)v, sqnorm, max_norm, grad_scale, 0.f, beta1, beta2, eps,
                       weight_decay, 1.f, 1.f, count / 4, hyper);
    REFID_LAUNCH_CHECK("clip_adamw_dev");
    return 0;
}
