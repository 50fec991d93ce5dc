// Non-GEMM pieces of the second network of the repository, SingleMultiConnectEVHINet (SURVEY.md 8f row 4;
// reference archs/single_multiconnect_evhinet_arch.py = "evh", archs/arch_util.py = "au"):
//   * HIN + LeakyReLU (evh:233-237): InstanceNorm2d(affine, biased variance, eps) on the first `ch` channels,
//     identity on the rest, LeakyReLU on all -- forward and backward;
//   * FAC_bias (au:421-426): out = feat * filt[:, :C] + filt[:, C:]  -- forward and backward.
// NHWC fp32, 16 bytes per lane per access.  All HBM-bound streaming / reduction kernels.  The per-(sample,
// channel) reductions are two-stage (per-workgroup partials, then a fixed-order final sum in double): no
// atomics, bit-reproducible.
#include "common.h"

namespace {

// ---- per-(n, c) sums over the pixels of one part: A = sum a, B = sum b  (a, b given by the functor) -------
// grid (parts, N), 256 threads = R pixel rows x Q channel quads (Q = ch / 4, a power of two <= 256)
template <class F>
__device__ __forceinline__ void part_sums(F f, int HW, int Q, int ch, float* __restrict__ parts) {
    __shared__ f32x4 redA[256], redB[256];
    const int q = threadIdx.x % Q, r = threadIdx.x / Q, R = 256 / Q;
    const int nparts = gridDim.x, part = blockIdx.x, n = blockIdx.y;
    const int chunk = (HW + nparts - 1) / nparts;
    const int p0 = part * chunk, p1 = min(HW, p0 + chunk);
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + r; p < p1; p += R) f((long long)n * HW + p, q, sa, sb);
    redA[threadIdx.x] = sa; redB[threadIdx.x] = sb;
    __syncthreads();
    for (int o = R >> 1; o > 0; o >>= 1) {
        if (r < o) { redA[threadIdx.x] += redA[threadIdx.x + o * Q]; redB[threadIdx.x] += redB[threadIdx.x + o * Q]; }
        __syncthreads();
    }
    if (r == 0) {
        float* dst = parts + ((long long)n * nparts + part) * 2 * ch + q * 4;
        *reinterpret_cast<f32x4*>(dst) = redA[q];
        *reinterpret_cast<f32x4*>(dst + ch) = redB[q];
    }
}

__global__ __launch_bounds__(256) void hin_stats_kernel(const float* __restrict__ x, int ldx, int HW, int Q, int ch,
                                                       float* __restrict__ parts) {
    part_sums([&](long long pix, int q, f32x4& sa, f32x4& sb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ldx + q * 4);
        sa += v; sb += v * v;
    }, HW, Q, ch, parts);
}

// mean / rstd per (n, c) from the partials (fixed order, double)
__global__ __launch_bounds__(256) void hin_finalize_kernel(const float* __restrict__ parts, int nparts, int HW, int ch,
                                                          float eps, float* __restrict__ stats) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < ch; c += 256) {
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < nparts; ++k) {
            const float* p = parts + ((long long)n * nparts + k) * 2 * ch;
            s += p[c]; s2 += p[ch + c];
        }
        const double m = s / HW;
        double var = s2 / HW - m * m;
        if (var < 0.0) var = 0.0;
        stats[(long long)n * 2 * ch + c] = (float)m;
        stats[(long long)n * 2 * ch + ch + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(256) void hin_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ stats,
                                                       float* __restrict__ out, int ldo, long long npix, int HW, int C4,
                                                       int ch, float slope) {
    const long long total = npix * C4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pix = e / C4;
        const int c = (int)(e % C4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ldx + c);
        if (c < ch) {
            const float* st = stats + (pix / HW) * 2 * ch;
            const f32x4 m = *reinterpret_cast<const f32x4*>(st + c), rs = *reinterpret_cast<const f32x4*>(st + ch + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
            v = (v - m) * rs * g + b;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], slope);
        *reinterpret_cast<f32x4*>(out + pix * ldo + c) = v;
    }
}

// backward sums: s1 = sum gy, s2 = sum gy * xhat,  gy = g * lrelu'(out)
__global__ __launch_bounds__(256) void hin_bwd_stats_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ out,
                                                           int ldo, const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ stats, float slope, int HW, int Q,
                                                           int ch, float* __restrict__ parts) {
    const float* st = stats + (long long)blockIdx.y * 2 * ch;
    part_sums([&](long long pix, int q, f32x4& sa, f32x4& sb) {
        const int c = q * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + pix * ldg + c);
        const f32x4 ov = *reinterpret_cast<const f32x4*>(out + pix * ldo + c);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * ldx + c);
        const f32x4 m = *reinterpret_cast<const f32x4*>(st + c), rs = *reinterpret_cast<const f32x4*>(st + ch + c);
        f32x4 gy;
#pragma unroll
        for (int k = 0; k < 4; ++k) gy[k] = gv[k] * (ov[k] > 0.f ? 1.f : slope);
        sa += gy; sb += gy * ((xv - m) * rs);
    }, HW, Q, ch, parts);
}

// sums[n][0/1][c] from the partials; dgamma[c] += sum_n s2, dbeta[c] += sum_n s1 (one workgroup, fixed order)
__global__ __launch_bounds__(256) void hin_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, int N, int ch,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    for (int c = threadIdx.x; c < ch; c += 256) {
        double tg = 0.0, tb = 0.0;
        for (int n = 0; n < N; ++n) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < nparts; ++k) {
                const float* p = parts + ((long long)n * nparts + k) * 2 * ch;
                s1 += p[c]; s2 += p[ch + c];
            }
            sums[(long long)n * 2 * ch + c] = (float)s1;
            sums[(long long)n * 2 * ch + ch + c] = (float)s2;
            tb += s1; tg += s2;
        }
        dgamma[c] += (float)tg;
        dbeta[c] += (float)tb;
    }
}

__global__ __launch_bounds__(256) void hin_bwd_apply_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ out,
                                                           int ldo, const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ gamma, const float* __restrict__ stats,
                                                           const float* __restrict__ sums, float* __restrict__ gx, int ldgx,
                                                           long long npix, int HW, int C4, int ch, float slope) {
    const long long total = npix * C4;
    const float inv = 1.f / (float)HW;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pix = e / C4;
        const int c = (int)(e % C4) * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + pix * ldg + c);
        const f32x4 ov = *reinterpret_cast<const f32x4*>(out + pix * ldo + c);
        f32x4 gy;
#pragma unroll
        for (int k = 0; k < 4; ++k) gy[k] = gv[k] * (ov[k] > 0.f ? 1.f : slope);
        if (c < ch) {
            const long long n = pix / HW;
            const float* st = stats + n * 2 * ch;
            const float* sm = sums + n * 2 * ch;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * ldx + c);
            const f32x4 m = *reinterpret_cast<const f32x4*>(st + c), rs = *reinterpret_cast<const f32x4*>(st + ch + c);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(sm + c), s2 = *reinterpret_cast<const f32x4*>(sm + ch + c);
            const f32x4 xh = (xv - m) * rs;
            gy = rs * ga * (gy - s1 * inv - xh * (s2 * inv));
        }
        *reinterpret_cast<f32x4*>(gx + pix * ldgx + c) = gy;
    }
}

__global__ __launch_bounds__(256) void fac_fwd_kernel(const float* __restrict__ feat, int ldf, const float* __restrict__ filt,
                                                     int ldfi, float* __restrict__ out, int ldo, long long npix, int C4, int C) {
    const long long total = npix * C4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pix = e / C4;
        const int c = (int)(e % C4) * 4;
        const f32x4 f = *reinterpret_cast<const f32x4*>(feat + pix * ldf + c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(filt + pix * ldfi + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(filt + pix * ldfi + C + c);
        *reinterpret_cast<f32x4*>(out + pix * ldo + c) = f * w + b;
    }
}

__global__ __launch_bounds__(256) void fac_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ feat, int ldf,
                                                     const float* __restrict__ filt, int ldfi, float* __restrict__ gfeat, int ldgf,
                                                     float* __restrict__ gfilt, int ldgfi, long long npix, int C4, int C) {
    const long long total = npix * C4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pix = e / C4;
        const int c = (int)(e % C4) * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + pix * ldg + c);
        const f32x4 f = *reinterpret_cast<const f32x4*>(feat + pix * ldf + c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(filt + pix * ldfi + c);
        *reinterpret_cast<f32x4*>(gfeat + pix * ldgf + c) = gv * w;
        *reinterpret_cast<f32x4*>(gfilt + pix * ldgfi + c) = gv * f;
        *reinterpret_cast<f32x4*>(gfilt + pix * ldgfi + C + c) = gv;
    }
}

int ew_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int refid_hin_parts(int hw) {
    const int p = cdiv(hw, 1024);
    return p < 1 ? 1 : (p > 128 ? 128 : p);
}

extern "C" int refid_hin_lrelu_fwd(const float* x, int ld_x, const float* gamma, const float* beta, float* out, int ld_out,
                                   float* stats, float* parts, int n, int hw, int c, int ch, float eps, float slope,
                                   void* stream) {
    REFID_CHECK(x && out && n > 0 && hw > 0 && c > 0 && c % 4 == 0 && ld_x % 4 == 0 && ld_out % 4 == 0,
                "hin_lrelu_fwd: bad arguments (c=%d)", c);
    REFID_CHECK(ch == 0 || (ch % 4 == 0 && ch <= c && pow2(ch / 4) && ch <= 1024 && gamma && beta && stats && parts),
                "hin_lrelu_fwd: normalised channel count %d must be 4 * 2^k with gamma/beta/stats/parts given", ch);
    hipStream_t st = (hipStream_t)stream;
    if (ch > 0) {
        const int np = refid_hin_parts(hw);
        hipLaunchKernelGGL(hin_stats_kernel, dim3(np, n), dim3(256), 0, st, x, ld_x, hw, ch / 4, ch, parts);
        REFID_LAUNCH_CHECK("hin_stats");
        hipLaunchKernelGGL(hin_finalize_kernel, dim3(n), dim3(256), 0, st, parts, np, hw, ch, eps, stats);
        REFID_LAUNCH_CHECK("hin_finalize");
    }
    const long long npix = (long long)n * hw;
    hipLaunchKernelGGL(hin_apply_kernel, dim3(ew_blocks(npix * (c / 4))), dim3(256), 0, st, x, ld_x, gamma, beta, stats, out,
                       ld_out, npix, hw, c / 4, ch, slope);
    REFID_LAUNCH_CHECK("hin_apply");
    return 0;
}

extern "C" int refid_hin_lrelu_bwd(const float* g, int ld_g, const float* out, int ld_out, const float* x, int ld_x,
                                   const float* gamma, const float* stats, float* gx, int ld_gx, float* dgamma,
                                   float* dbeta, float* sums, float* parts, int n, int hw, int c, int ch, float slope,
                                   void* stream) {
    REFID_CHECK(g && out && gx && n > 0 && hw > 0 && c > 0 && c % 4 == 0 && ld_g % 4 == 0 && ld_out % 4 == 0 &&
                    ld_gx % 4 == 0,
                "hin_lrelu_bwd: bad arguments (c=%d)", c);
    REFID_CHECK(ch == 0 || (ch % 4 == 0 && ch <= c && pow2(ch / 4) && ch <= 1024 && x && ld_x % 4 == 0 && gamma && stats &&
                            dgamma && dbeta && sums && parts),
                "hin_lrelu_bwd: normalised channel count %d needs x/gamma/stats/dgamma/dbeta/sums/parts", ch);
    hipStream_t st = (hipStream_t)stream;
    if (ch > 0) {
        const int np = refid_hin_parts(hw);
        hipLaunchKernelGGL(hin_bwd_stats_kernel, dim3(np, n), dim3(256), 0, st, g, ld_g, out, ld_out, x, ld_x, stats, slope,
                           hw, ch / 4, ch, parts);
        REFID_LAUNCH_CHECK("hin_bwd_stats");
        hipLaunchKernelGGL(hin_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, parts, np, n, ch, sums, dgamma, dbeta);
        REFID_LAUNCH_CHECK("hin_bwd_finalize");
    }
    const long long npix = (long long)n * hw;
    hipLaunchKernelGGL(hin_bwd_apply_kernel, dim3(ew_blocks(npix * (c / 4))), dim3(256), 0, st, g, ld_g, out, ld_out, x, ld_x,
                       gamma, stats, sums, gx, ld_gx, npix, hw, c / 4, ch, slope);
    REFID_LAUNCH_CHECK("hin_bwd_apply");
    return 0;
}

extern "C" int refid_fac_fwd(const float* feat, int ld_feat, const float* filt, int ld_filt, float* out, int ld_out,
                             long long npix, int c, void* stream) {
    REFID_CHECK(feat && filt && out && npix > 0 && c > 0 && c % 4 == 0 && ld_feat % 4 == 0 && ld_filt % 4 == 0 &&
                    ld_out % 4 == 0 && ld_filt >= 2 * c,
                "fac_fwd: bad arguments (c=%d)", c);
    hipLaunchKernelGGL(fac_fwd_kernel, dim3(ew_blocks(npix * (c / 4))), dim3(256), 0, (hipStream_t)stream, feat, ld_feat, filt,
                       ld_filt, out, ld_out, npix, c / 4, c);
    REFID_LAUNCH_CHECK("fac_fwd");
    return 0;
}

extern "C" int refid_fac_bwd(const float* g, int ld_g, const float* feat, int ld_feat, const float* filt, int ld_filt,
                             float* gfeat, int ld_gfeat, float* gfilt, int ld_gfilt, long long npix, int c, void* stream) {
    REFID_CHECK(g && feat && filt && gfeat && gfilt && npix > 0 && c > 0 && c % 4 == 0 && ld_g % 4 == 0 &&
                    ld_feat % 4 == 0 && ld_filt % 4 == 0 && ld_gfeat % 4 == 0 && ld_gfilt % 4 == 0 && ld_gfilt >= 2 * c,
                "fac_bwd: bad arguments (c=%d)", c);
    hipLaunchKernelGGL(fac_bwd_kernel, dim3(ew_blocks(npix * (c / 4))), dim3(256), 0, (hipStream_t)stream, g, ld_g, feat,
                       ld_feat, filt, ld_filt, gfeat, ld_gfeat, gfilt, ld_gfilt, npix, c / 4, c);
    REFID_LAUNCH_CHECK("fac_bwd");
    return 0;
}
