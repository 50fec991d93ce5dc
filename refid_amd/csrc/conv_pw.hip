// Pointwise (1x1, stride 1) convolution tile for the short-K layers of the REFID hot path:
// fuse_two_dir (rsm:291-293), ImageEncoderConvBlock.identity (rsm:28,47), the five 1x1 convs of
// EGACA (fm:300-331) and their input gradients.
//
//   out = mask( post( pre(W [a|b] + bias) + res ) )        -- same contract as conv_igemm.hip
//
// These layers have K = 64..512 and are bandwidth/latency bound (5-40 FLOP/B): a workgroup-tiled
// GEMM spends most of its time in prologue/epilogue.  Here there is NO LDS and NO barrier: a 1x1
// conv has no spatial structure, so each wave owns 32 consecutive pixels x all (<=128) output
// channels of its column tile and feeds v_mfma_f32_32x32x2_f32 straight from registers --
// activations: one 16-byte buffer load per lane per 8-channel chunk (lane = pixel, K quad = l>>5);
// weights: the packed [chunk][1][cout][8] rows, one 16-byte buffer load per lane per 32 channels
// (1 KB contiguous per wave, L2 resident).  Activations are prefetched 32-64 channels ahead, weights one
// chunk ahead; <= 128 registers => 4 waves per SIMD overlap the load / MFMA / store phases of different
// waves (measured: occupancy matters more here than activation re-use across column tiles).  Fused 16-byte epilogue as everywhere else.
//
// SIX (refid_conv_desc.mfma_terms = 6; channel counts multiples of 16): the same tile with its products on the bf16 matrix
// cores -- every fp32 operand is the exact sum of three bf16 numbers (weights pre-split by refid_pack_conv_weights_split's
// 1x1 layout, activations split in registers: 44 VALU per 8 values), six v_mfma_f32_32x32x16_bf16 per fp32 product
// (conv_split.hip's list).  Per 16 input channels and 64 output channels: 12 x 32 = 384 matrix-pipe cycles instead of
// 16 x 64 = 1024, at the fp32 tile's distance from the float64 result.  The waves of a workgroup run their load / MFMA /
// store phases in lockstep, so the MFMA phase is a third of the kernel's time when the operands are warm (1.03-1.26x then,
// tools/bench_pw6.py); inside the train step, on cold operands, the two forms time the same (39.8 vs 39.2 ms per step): the
// engine keeps the fp32 form (REFID_PW6=1 switches).
#include "common.h"
#include "conv_args.h"
#include <cstdlib>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KC = 8;
// the six products kept, largest first: (weight plane, activation plane)
__device__ constexpr int PW_TA[6] = {0, 1, 0, 2, 0, 1};
__device__ constexpr int PW_TB[6] = {0, 0, 1, 0, 2, 1};

// v = h + m + l exactly (h = rne(v), m = rne(v - h), l = v - h - m), eight values -> three bf16x8 planes
__device__ __forceinline__ void pw_split8(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[3]) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = k < 4 ? v0[k] : v1[k - 4];
        const __bf16 h = (__bf16)v;
        p0[k] = h;
        const float r = v - (float)h;
        const __bf16 m = (__bf16)r;
        p1[k] = m;
        p2[k] = (__bf16)(r - (float)m);
    }
    pl[0] = __builtin_bit_cast(f32x4, p0);
    pl[1] = __builtin_bit_cast(f32x4, p1);
    pl[2] = __builtin_bit_cast(f32x4, p2);
}
constexpr int OOB = -1;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_d(float x) {      // d/dx GELU_erf = Phi(x) + x phi(x)   (as egaca.hip::gelu_d)
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// NT: 32-channel column tiles per wave (Cout tile = 32*NT); XD: activation prefetch depth; EX: the EGACA fusions of
// PwExtra are compiled in (fusion_modules.py:290-333):
//   * LayerNorm2d PROLOGUE (e.lnG): K <= 8*2*XD channels, so a lane holds its half of the pixel's channels in the
//     2*XD staging registers at once -- mean / variance over the channel axis (two-pass, the partner half arrives by one
//     cross-lane exchange), normalise in registers, optionally store the normalised tensor (training stash), then the
//     MFMAs: conv(LN(x)) in one pass over x;
//   * SQUEEZE-EXCITE in the kernel (e.pool): every workgroup reduces the per-workgroup pool partials of the depthwise
//     kernel for ITS sample, runs the two tiny mat-vecs + sigmoid (global-pool -> 1x1 -> ReLU -> 1x1 -> sigmoid,
//     fm:253-260) and multiplies the operand channels by the result while they are loaded (fm:312-315: the scaled
//     [xi*s | xe*s] concatenation is never a tensor on the inference path; training stores it once for conv3's weight
//     gradient);
//   * a second residual (y = ev + img + beta*conv3(.), fm:319) and a GELU second output (fm:327-329).
template <int NT, int XD, bool EX, bool SIX = false>
__global__ __launch_bounds__(256, (SIX || NT > 2) ? 3 : 4) void conv_pw_kernel(const ConvKArgs a, const PwExtra e) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const long long npix = (long long)a.N * a.H * a.W;
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * 32;
    const long long p = p0 + li;
    const int n0 = blockIdx.y * (32 * NT);

    // (patch form: the sources are the even / odd rows of one tensor of npix / patchW row pairs; live lanes stay inside it)
    const long long spanA = a.patchW ? (npix / a.patchW) * (long long)a.patchRow : npix * a.ldA;
    const long long spanB = a.patchW ? spanA - a.patchRow / 2 : npix * (a.inB ? a.ldB : a.ldA);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.inA), 0, (int)min(spanA * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.inB ? a.inB : a.inA), 0, (int)min(spanB * 4, 0x7fffffffLL), 0x00020000);
    // (SIX: [chunk16][plane][row][16] bf16 = 3 x 32 bytes per row and 16 channels -- 1.5x the fp32 bytes)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)min((long long)a.nchunks * a.CoutPad * KC * (SIX ? 6 : 4), 0x7fffffffLL), 0x00020000);
    const bool pok = p < npix;
    const long long pA = a.patchW ? (p / a.patchW) * a.patchRow + (p % a.patchW) * a.ldA : p * a.ldA;      // floats
    const long long pB = a.patchW ? (p / a.patchW) * a.patchRow + (p % a.patchW) * a.ldB : p * a.ldB;
    const int voA = pok ? (int)(pA * 4) + kh * 16 : OOB;
    const int voB = pok ? (int)(pB * 4) + kh * 16 : OOB;
    int voW[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = a.coBase + n0 + nt * 32 + li;
        voW[nt] = (row < a.CoutPad) ? (SIX ? row * 32 + kh * 16 : (row * KC + kh * 4) * 4) : OOB;
    }
    const int wChunk = a.CoutPad * KC * 4;
    const int wPlane = a.CoutPad * 32;                      // SIX: bytes per (chunk16, plane)

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    auto load_x = [&](int ch) -> f32x4 {
        const int c0 = ch * KC;
        const bool fromA = c0 < a.Ca;
        const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
        const int vo = (c0 + kh * 4 < a.Ctot) ? (fromA ? voA : voB) : OOB;
        const u32x4 v = fromA ? __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, soff, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, soff, 0);
        return __builtin_bit_cast(f32x4, v);
    };
    auto load_w = [&](int ch, f32x4 (&dst)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            dst[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[nt], ch * wChunk, 0));
    };
    // SIX: the three planes of chunk pair c16; lane (li = row, kh) gets channels {4kh..4kh+3, 8+4kh..8+4kh+3} of the 16 --
    // the order in which the activation registers of two consecutive 8-channel chunks line up
    auto load_w6 = [&](int c16, f32x4 (&dst)[3][NT]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dst[pl][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[nt], (c16 * 3 + pl) * wPlane, 0));
    };
    auto mfma6 = [&](const f32x4& x0, const f32x4& x1, const f32x4 (&w6)[3][NT]) {
        f32x4 pl[3];
        pw_split8(x0, x1, pl);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w6[PW_TA[t]][nt]),
                                                                  __builtin_bit_cast(bf16x8, pl[PW_TB[t]]), acc[nt], 0, 0, 0);
    };

    // Activations are prefetched XD chunks (= 8*XD channels, HBM latency) ahead, weights one chunk
    // (L2 latency) ahead.  vmcnt retires in issue order, so inside iteration c the weight load of c+1
    // is issued BEFORE the activation load of c+XD: waiting for the former never waits for the latter.
    f32x4 xr[2 * XD];
    f32x4 wr[SIX ? 1 : 2][NT];
    f32x4 w6[SIX ? 2 : 1][3][NT];
    const int nch = a.nchunks;
    __shared__ float sS[EX ? 128 : 1];                       // squeeze-excite vector of this workgroup's sample
    bool ln_done = false;
    if constexpr (EX) {
        if (e.pool != nullptr) {
            __shared__ float sM[128], sZ[64];
            const int C = e.seC, Ch = C / 2, tid = threadIdx.x;
            const int n = (int)(((long long)blockIdx.x * 128) / e.hw);          // host: hw % 128 == 0
            for (int c = tid; c < C; c += 256) {
                const float* pp = e.pool + (long long)n * e.poolParts * C + c;
                float s0 = 0.f;
                for (int r = 0; r < e.poolParts; ++r) s0 += pp[(long long)r * C];      // fixed order: deterministic
                sM[c] = s0 * e.invHW;
            }
            __syncthreads();
            for (int j = tid; j < Ch; j += 256) {
                float z = e.seB1[j];
                for (int c = 0; c < C; ++c) z += e.seW1[j * C + c] * sM[c];
                sZ[j] = z > 0.f ? z : 0.f;
            }
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                float v = e.seB2[c];
                for (int j = 0; j < Ch; ++j) v += e.seW2[c * Ch + j] * sZ[j];
                sS[c] = 1.f / (1.f + __expf(-v));
            }
            __syncthreads();
            if (e.seS != nullptr && blockIdx.y == 0 && ((long long)blockIdx.x * 128) % e.hw == 0) {   // one writer per sample
                for (int c = tid; c < C; c += 256) { e.seM[n * C + c] = sM[c]; e.seS[n * C + c] = sS[c]; }
                for (int j = tid; j < Ch; j += 256) e.seZ1[n * Ch + j] = sZ[j];
            }
        }
        if (e.lnG != nullptr) {
            // LayerNorm2d prologue: all nch (<= 2*XD) chunks of this lane's half of the pixel in registers
#pragma unroll
            for (int j = 0; j < 2 * XD; ++j) xr[j] = (j < nch) ? load_x(j) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SIX) load_w6(0, w6[0]);
            else load_w(0, wr[0]);
            float s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * XD; ++j) s1 += (xr[j][0] + xr[j][1]) + (xr[j][2] + xr[j][3]);
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = s1 / (float)a.Ctot;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * XD; ++j)
                if (j < nch) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float d = xr[j][k] - mu; s2 += d * d; }
                }
            s2 += __shfl_xor(s2, 32, 64);
            const float rstd = 1.f / sqrtf(s2 / (float)a.Ctot + e.lnEps);
#pragma unroll
            for (int j = 0; j < 2 * XD; ++j)
                if (j < nch) {
                    const int c = j * KC + kh * 4;
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(e.lnG + c);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(e.lnB + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) xr[j][k] = (xr[j][k] - mu) * rstd * gv[k] + bv[k];
                    if (e.lnOut != nullptr && blockIdx.y == 0 && pok)
                        *reinterpret_cast<f32x4*>(e.lnOut + p * e.ldLn + c) = xr[j];
                }
            if constexpr (SIX) {
#pragma unroll
                for (int j = 0; j < 2 * XD; j += 2)
                    if (j < nch) {
                        if (j + 2 < nch) load_w6(j / 2 + 1, w6[(j / 2 + 1) & 1]);
                        mfma6(xr[j], xr[j + 1], w6[(j / 2) & 1]);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2 * XD; ++j)
                    if (j < nch) {
                        if (j + 1 < nch) load_w(j + 1, wr[(j + 1) & 1]);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[j & 1][nt][kk], xr[j][kk], acc[nt], 0, 0, 0);
                    }
            }
            ln_done = true;
        }
    }
    if (!ln_done) {
#pragma unroll
        for (int j = 0; j < XD; ++j)
            if (j < nch) xr[j] = load_x(j);
        if constexpr (SIX) load_w6(0, w6[0]);
        else load_w(0, wr[0]);
        for (int c0 = 0; c0 < nch; c0 += 2 * XD) {
#pragma unroll
            for (int j = 0; j < 2 * XD; ++j) {
                const int c = c0 + j;
                if (c < nch) {
                    if constexpr (SIX) {
                        if ((j & 1) == 0 && c + 2 < nch) load_w6(c / 2 + 1, w6[(j / 2 + 1) & 1]);
                    } else {
                        if (c + 1 < nch) load_w(c + 1, wr[(j + 1) & 1]);
                    }
                    if (c + XD < nch) xr[(j + XD) % (2 * XD)] = load_x(c + XD);
                    if constexpr (EX) {
                        if (e.pool != nullptr) {                   // operand channel k scaled by s[k mod C]
                            const int k0 = c * KC + kh * 4;
                            const f32x4 sv = *reinterpret_cast<const f32x4*>(&sS[k0 % e.seC]);
                            xr[j] *= sv;
                            if (e.xsOut != nullptr && blockIdx.y == 0 && pok)
                                *reinterpret_cast<f32x4*>(e.xsOut + p * e.ldXs + k0) = xr[j];
                        }
                    }
                    if constexpr (SIX) {
                        if (j & 1) mfma6(xr[j - 1], xr[j], w6[(j / 2) & 1]);       // host: an even number of 8-channel chunks
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[j & 1][nt][kk], xr[j][kk], acc[nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- fused epilogue ------------------------------------------------------------------------------
    // D[cout][pixel]: lane (li = pixel, kh) holds, per register quad g, channels 8g + 4kh + {0..3}.  Written
    // straight from that layout a store instruction touches 32 B in each of 32 rows; instead the wave's tile is
    // transposed through a private LDS slab so that consecutive lanes handle consecutive 16-byte pieces of a
    // pixel: bias / residual / mask loads and the output store are then fully coalesced (256 B runs).
    if (a.vecOK) {
        constexpr int HN = (NT > 2) ? 2 : NT;          // column tiles per pass
        constexpr int PITCH = 8 * HN + 1;              // float4 per pixel row (+1: bank spread)
        __shared__ f32x4 sT[4][32 * PITCH];
        f32x4* t = sT[wave];
#pragma unroll
        for (int h0 = 0; h0 < NT; h0 += HN) {
            if (h0) __syncthreads();
#pragma unroll
            for (int nt = 0; nt < HN; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[h0 + nt][4 * g + k];
                    t[li * PITCH + nt * 8 + 2 * g + kh] = v;
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4 * HN; ++it) {
                const int f = it * 64 + lane;
                const int px = f / (8 * HN), c4 = f % (8 * HN);
                const long long pp = p0 + px;
                const int j0 = n0 + h0 * 32 + c4 * 4;
                if (pp >= npix || j0 >= a.Cout) continue;
                f32x4 v = t[px * PITCH + c4];
                if (a.shuffle) {                            // (workgroup-uniform) ConvTranspose2d(2,2): pixel-shuffle store
                    const int Co1 = a.Cout >> 2;
                    const unsigned qd = (unsigned)j0 / (unsigned)Co1, ch = (unsigned)j0 - qd * Co1;
                    const unsigned up = (unsigned)pp, row = up / (unsigned)a.W, x = up - row * a.W;   // row = n H + y
                    const long long op = (long long)(2 * row + (qd >> 1)) * (2 * a.W) + 2 * x + (qd & 1);
                    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + ch);
                    lrelu4(v, a.slopePre, a.slopePre != 1.f);
                    if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + ch);
                    lrelu4(v, a.slopePost, a.slopePost != 1.f);
                    *reinterpret_cast<f32x4*>(a.out + op * a.ldO + ch) = v;
                    continue;
                }
                if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + a.coBase + j0);
                lrelu4(v, a.slopePre, a.slopePre != 1.f);
                if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + pp * a.ldR + j0);
                if constexpr (EX) {
                    if (e.res2) v += *reinterpret_cast<const f32x4*>(e.res2 + pp * e.ldR2 + j0);
                }
                lrelu4(v, a.slopePost, a.slopePost != 1.f);
                if (a.mask) {
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(a.mask + pp * a.ldM + j0);
                    if (a.maskMode == 1) {                  // (workgroup-uniform) GELU backward fused into conv5's input gradient
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] *= gelu_erf_d(mv[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
                    }
                }
                *reinterpret_cast<f32x4*>(a.out + pp * a.ldO + j0) = v;
                if (a.out2)                                  // (workgroup-uniform) the skip sum leaves with the tile
                    *reinterpret_cast<f32x4*>(a.out2 + pp * a.ldO2 + j0) = v + *reinterpret_cast<const f32x4*>(a.add2 + pp * a.ldA2 + j0);
                if constexpr (EX) {
                    if (e.out2) {
                        f32x4 gl;
#pragma unroll
                        for (int k = 0; k < 4; ++k) gl[k] = gelu_erf(v[k]);
                        *reinterpret_cast<f32x4*>(e.out2 + pp * e.ldO2 + j0) = gl;
                    }
                }
            }
        }
        return;
    }
    // scalar tail path (channel counts / pitches that are not multiples of 4)
    if (!pok) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = n0 + nt * 32 + 8 * g + 4 * kh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= a.Cout) break;
                float tv = acc[nt][4 * g + k] + (a.bias ? a.bias[a.coBase + j0 + k] : 0.f);
                tv = lrelu(tv, a.slopePre);
                if (a.res) tv += a.res[p * a.ldR + j0 + k];
                tv = lrelu(tv, a.slopePost);
                if (a.mask) tv *= a.maskMode == 1 ? gelu_erf_d(a.mask[p * a.ldM + j0 + k]) : ((a.mask[p * a.ldM + j0 + k] > 0.f) ? 1.f : a.slopeMask);
                a.out[p * a.ldO + j0 + k] = tv;
            }
        }
    }
}

}  // namespace

int refid_launch_pointwise(const ConvKArgs& ka, hipStream_t st, const PwExtra* ex, int terms) {
    ConvKArgs a = ka;
    a.nchunks = cdiv(a.Ctot, KC);
    const long long npix = (long long)a.N * a.H * a.W;
    const int nb = (int)((npix + 127) / 128);
    const bool six = terms == 6;
    REFID_CHECK(!a.shuffle || (a.vecOK && ex == nullptr && terms == 0 && a.mask == nullptr && a.Cout > 32 && (a.Cout / 4) % 4 == 0 &&
                               npix < 0x7fffffffLL),
                "conv2d: ConvTranspose2d on the pointwise tile needs 16-byte aligned tensors, fp32 products, no mask / fusions and "
                "at least 16 output channels");
    REFID_CHECK(a.out2 == nullptr || (a.vecOK && ex == nullptr), "conv2d: the pointwise tile's second output needs 16-byte aligned tensors and no fusions");
    REFID_CHECK(!a.patchW || (terms == 0 && ex == nullptr && a.inB != nullptr), "conv2d: the patch form of the pointwise tile has fp32 products, no fusions");
    REFID_CHECK(terms == 0 || terms == 6, "conv2d: the pointwise tile has fp32 products (0) or six bf16 products (6), got %d", terms);
    REFID_CHECK(!six || (a.Ctot % 16 == 0 && (a.inB == nullptr || a.Ca % 16 == 0) && a.Cout > 32),
                "conv2d: the six-product pointwise tile needs channel counts that are multiples of 16 and more than 32 outputs");
    if (ex != nullptr) {
        REFID_CHECK(a.vecOK && a.Cout > 32, "conv2d: pointwise fusions need 16-byte aligned tensors and more than 32 outputs");
        REFID_CHECK(!ex->lnG || (a.inB == nullptr && a.Ctot % 8 == 0 && a.Ctot <= 64 && ex->lnB),
                    "conv2d: the LayerNorm prologue needs one source of at most 64 channels (got %d)", a.Ctot);
        REFID_CHECK(!ex->pool || (ex->hw > 0 && ex->hw % 128 == 0 && ex->seC >= 8 && ex->seC <= 128 && ex->seC % 8 == 0 &&
                                  a.Ctot % ex->seC == 0 && ex->poolParts > 0 && ex->seW1 && ex->seB1 && ex->seW2 && ex->seB2 &&
                                  (!ex->seS || (ex->seM && ex->seZ1)) && !ex->lnG),
                    "conv2d: bad squeeze-excite fusion arguments (pixels per sample must be a multiple of 128)");
        if (six) hipLaunchKernelGGL((conv_pw_kernel<2, 4, true, true>), dim3(nb, cdiv(a.Cout, 64)), dim3(256), 0, st, a, *ex);
        else hipLaunchKernelGGL((conv_pw_kernel<2, 4, true>), dim3(nb, cdiv(a.Cout, 64)), dim3(256), 0, st, a, *ex);
        REFID_LAUNCH_CHECK("conv_pw/fused");
        return 0;
    }
    const PwExtra none;
    // wider layers run as 64-channel column tiles (grid.y): 4 waves/SIMD beat re-using the activations
    if (six) hipLaunchKernelGGL((conv_pw_kernel<2, 4, false, true>), dim3(nb, cdiv(a.Cout, 64)), dim3(256), 0, st, a, none);
    else if (a.Cout <= 32) hipLaunchKernelGGL((conv_pw_kernel<1, 8, false>), dim3(nb, 1), dim3(256), 0, st, a, none);
    else {
        // REFID_PW_NT4=1 (experiment, round 4): 128-channel column tiles -- a wave computes four 32-channel tiles from ONE pass
        // over its activations (half the workgroups, no second read of the operand) at 3 instead of 4 waves per SIMD.
        // Measured slower: 42.1 vs 39.4 us at 64 -> 128 @128^2 (warm), train step 463.0 vs 461.0 ms.  Occupancy beats re-use.
        static const bool nt4 = []() { const char* e = getenv("REFID_PW_NT4"); return e && e[0] == '1'; }();
        if (nt4 && a.Cout % 128 == 0)
            hipLaunchKernelGGL((conv_pw_kernel<4, 4, false>), dim3(nb, a.Cout / 128), dim3(256), 0, st, a, none);
        else
            hipLaunchKernelGGL((conv_pw_kernel<2, 4, false>), dim3(nb, cdiv(a.Cout, 64)), dim3(256), 0, st, a, none);
    }
    REFID_LAUNCH_CHECK("conv_pw");
    return 0;
}
