// EGACA (Event-Guided Adaptive Channel Attention) support kernels and the other
// HBM-bound elementwise / reduction pieces of the REFID hot path.
//
// Reference: fusion_modules.py:97-134 (LayerNorm2d, hand-written backward :110-122) and
// :290-333 (CrossmodalAtten_imgeventalladd.forward).  The 1x1 convolutions of EGACA run on
// the fused conv tile (conv_igemm.hip); what is here is everything that is NOT a GEMM:
//   per-pixel channel LayerNorm fwd/bwd, depthwise 3x3 (+bias) -> GELU (+ global-average-pool
//   partial sums) fwd/bwd, the squeeze-excite MLP fwd/bwd, channel scaling + concat, GELU,
//   column sums (bias gradients), and the beta/gamma fold-back.
//
// All of these are bandwidth-bound: every thread moves 16 bytes per access (float4 over 4
// consecutive NHWC channels), a pixel's C channels are read by C/4 adjacent lanes (fully
// coalesced 4*C-byte runs).  Parameter-gradient reductions are DETERMINISTIC: fixed xor-shuffle tree over
// the pixel rows of a wave -> the four waves in order through LDS -> one partial row per workgroup in the
// caller's scratch -> a second tiny kernel adds the rows in a fixed order.  No floating-point atomics.
#include "common.h"
#include <vector>
#include <cstring>

namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// sum over the LPP lanes that share a pixel (LPP power of two <= 64, lanes contiguous)
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the workgroup's pixel rows (threads that share q = tid % LPP) in a FIXED order: xor-shuffle tree over the
// rows of a wave (the value ends up in every lane), then lanes < LPP of each wave put it into sh[wave][col].
template <int LPP>
__device__ __forceinline__ float wave_rows_sum(float v) {
#pragma unroll
    for (int o = LPP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[i] += sum_rows parts[row][i] for i < n_a (into dst_a) and n_a <= i < ncols (into dst_b; dst_b may be NULL when
// n_a == ncols); accumulate = 0 overwrites.  One wave per column: lanes stride over the partial rows, then a fixed
// xor-shuffle tree -- deterministic.
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ parts, int nrows, int ncols,
                                                      float* __restrict__ dst_a, int n_a, float* __restrict__ dst_b,
                                                      int accumulate) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= ncols) return;
    float a = 0.f;
    for (int b = lane; b < nrows; b += 64) a += parts[(long long)b * ncols + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) {
        float* d = i < n_a ? dst_a + i : dst_b + (i - n_a);
        *d = accumulate ? *d + a : a;
    }
}

// Deferred form (refid_rows_sum_defer / refid_rows_sum_flush): BPTT issues ~210 of these 5 us launches per step (LayerNorm,
// depthwise and bias gradients of every time step), each a link of the dependent chain although nothing reads the parameter
// gradients before the end of the BPTT half.  While deferral is on the three callers below QUEUE their sum; the flush groups the
// queued sums by destination and issues them as one launch per ~160 sums: a wave owns a column of a destination and adds the
// queued partial-row sets IN CALL ORDER, each with the same lane-strided sum and xor tree as rows_sum_kernel -- the bits of the
// one-by-one launches.  The caller keeps the `parts` buffers alive and the destinations untouched until the flush.
constexpr int ROWS_JOBS = 160, ROWS_GROUPS = 24;
struct RowsJob { const float* parts; int nrows; int pad; };
struct RowsGroup { float* dst_a; float* dst_b; int ncols, n_a, first, count, blk0, pad; };
struct RowsBatch { RowsJob job[ROWS_JOBS]; RowsGroup grp[ROWS_GROUPS]; int ngroups; };
static_assert(sizeof(RowsBatch) <= 4096, "kernel-argument block");

__global__ __launch_bounds__(256) void rows_sum_batch_kernel(const RowsBatch b) {
    int gi = 0;
    for (int k = 1; k < b.ngroups; ++k) gi = (int)blockIdx.x >= b.grp[k].blk0 ? k : gi;        // (workgroup-uniform)
    const RowsGroup& g = b.grp[gi];
    const int lane = threadIdx.x & 63;
    const int i = ((int)blockIdx.x - g.blk0) * 4 + (threadIdx.x >> 6);
    if (i >= g.ncols) return;
    float* d = i < g.n_a ? g.dst_a + i : g.dst_b + (i - g.n_a);
    float tot = *d;
    for (int j = g.first; j < g.first + g.count; ++j) {
        const float* parts = b.job[j].parts;
        const int nrows = b.job[j].nrows;
        float a = 0.f;
        for (int r = lane; r < nrows; r += 64) a += parts[(long long)r * g.ncols + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        tot += a;
    }
    if (lane == 0) *d = tot;
}

struct RowsPending { const float* parts; int nrows, ncols; float* dst_a; int n_a; float* dst_b; };
thread_local std::vector<RowsPending> rows_queue;
thread_local bool rows_defer = false;

// accumulate == 1 sums only (every caller below)
int rows_sum_issue(const float* parts, int nrows, int ncols, float* dst_a, int n_a, float* dst_b, hipStream_t st, const char* what) {
    if (rows_defer) {
        rows_queue.push_back({parts, nrows, ncols, dst_a, n_a, dst_b});
        return 0;
    }
    hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(ncols, 4)), dim3(256), 0, st, parts, nrows, ncols, dst_a, n_a, dst_b, 1);
    REFID_LAUNCH_CHECK(what);
    return 0;
}

int rows_sum_flush(hipStream_t st) {
    // groups in order of first appearance, jobs of a group in call order; a group split over the job limit continues in a LATER
    // launch (stream order keeps the order of additions)
    std::vector<char> done(rows_queue.size(), 0);
    size_t left = rows_queue.size();
    while (left) {
        RowsBatch b;
        memset(&b, 0, sizeof(b));
        int nj = 0, ng = 0, blk = 0;
        for (size_t k = 0; k < rows_queue.size() && ng < ROWS_GROUPS && nj < ROWS_JOBS; ++k) {
            if (done[k]) continue;
            const RowsPending& q = rows_queue[k];
            bool open = false;                              // an earlier, unfinished part of this destination in this batch?
            for (int g = 0; g < ng; ++g) open = open || (b.grp[g].dst_a == q.dst_a && b.grp[g].dst_b == q.dst_b);
            if (open) continue;
            RowsGroup& g = b.grp[ng];
            g.dst_a = q.dst_a; g.dst_b = q.dst_b; g.ncols = q.ncols; g.n_a = q.n_a; g.first = nj; g.count = 0; g.blk0 = blk;
            for (size_t m = k; m < rows_queue.size() && nj < ROWS_JOBS; ++m) {
                const RowsPending& r = rows_queue[m];
                if (done[m] || r.dst_a != q.dst_a || r.dst_b != q.dst_b) continue;
                if (r.ncols != q.ncols || r.n_a != q.n_a) break;     // (same destination, another shape: next launch, in order)
                b.job[nj].parts = r.parts; b.job[nj].nrows = r.nrows;
                ++nj; ++g.count; done[m] = 1; --left;
            }
            blk += cdiv(g.ncols, 4);
            ++ng;
        }
        b.ngroups = ng;
        hipLaunchKernelGGL(rows_sum_batch_kernel, dim3(blk), dim3(256), 0, st, b);
        if (hipGetLastError() != hipSuccess) { rows_queue.clear(); refid_set_error("rows_sum_batch: launch failed"); return 1; }
    }
    rows_queue.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------
// LayerNorm2d forward: fm:100-108.  mean / biased variance over C per pixel, eps inside sqrt.
template <int LPP>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    float* __restrict__ out, int ldo, long long npix, float eps) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    const int q = threadIdx.x % LPP;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + q * 4);
    for (long long p = blockIdx.x * (long long)PPB + threadIdx.x / LPP; p < npix; p += (long long)gridDim.x * PPB) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ldx + q * 4);
        const float mu = group_sum<LPP>(v[0] + v[1] + v[2] + v[3]) * (1.f / C);
        const f32x4 d = v - mu;
        const float var = group_sum<LPP>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
        const float rstd = 1.f / sqrtf(var + eps);
        *reinterpret_cast<f32x4*>(out + p * ldo + q * 4) = wv * (d * rstd) + bv;
    }
}

// LayerNorm2d backward: fm:110-122.  gx = rstd * (g*w - y*mean(g*w*y) - mean(g*w)),
// parts[block][0:C] = sum g*y, parts[block][C:2C] = sum g over the block's pixels (rows_sum_kernel adds them into dw / db).
template <int LPP>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ g, int ldg,
                                                    const float* __restrict__ x, int ldx,
                                                    const float* __restrict__ w, float* gx, int ldgx,
                                                    const float* res, int ldr, float* __restrict__ parts,
                                                    long long npix, float eps) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    __shared__ float sred[4][2 * C];
    const int q = threadIdx.x % LPP;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + q * 4);
    f32x4 pdw = {0.f, 0.f, 0.f, 0.f}, pdb = {0.f, 0.f, 0.f, 0.f};
    // U pixels per thread per iteration: their 2-3 loads each are issued together (the four lane-group reductions per
    // pixel are dependent shuffle chains: with one pixel in flight the kernel sat at 0.42 of the HBM rate; now 0.54)
    constexpr int U = 4;
    const long long stride = (long long)gridDim.x * PPB;
    for (long long p0 = blockIdx.x * (long long)PPB + threadIdx.x / LPP; p0 < npix; p0 += stride * U) {
        f32x4 v[U], gv[U], rv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + u * stride;
            ok[u] = p < npix;
            const long long pc = ok[u] ? p : p0;                 // clamped: loads stay unconditional
            v[u] = *reinterpret_cast<const f32x4*>(x + pc * ldx + q * 4);
            gv[u] = *reinterpret_cast<const f32x4*>(g + pc * ldg + q * 4);
            rv[u] = res ? *reinterpret_cast<const f32x4*>(res + pc * ldr + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mu = group_sum<LPP>(v[u][0] + v[u][1] + v[u][2] + v[u][3]) * (1.f / C);
            const f32x4 d = v[u] - mu;
            const float var = group_sum<LPP>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
            const float rstd = 1.f / sqrtf(var + eps);
            const f32x4 y = d * rstd;
            const f32x4 gh = gv[u] * wv;
            const float m1 = group_sum<LPP>(gh[0] + gh[1] + gh[2] + gh[3]) * (1.f / C);
            const float m2 = group_sum<LPP>(gh[0] * y[0] + gh[1] * y[1] + gh[2] * y[2] + gh[3] * y[3]) * (1.f / C);
            const f32x4 r = (gh - y * m2 - m1) * rstd + rv[u];
            if (ok[u]) {                                           // res may alias gx: each pixel is read before it is written
                *reinterpret_cast<f32x4*>(gx + (p0 + u * stride) * ldgx + q * 4) = r;
                pdw += gv[u] * y;
                pdb += gv[u];
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = wave_rows_sum<LPP>(pdw[k]), b = wave_rows_sum<LPP>(pdb[k]);
        if (lane < LPP) { sred[wave][q * 4 + k] = a; sred[wave][C + q * 4 + k] = b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        parts[(long long)blockIdx.x * (2 * C) + i] = ((sred[0][i] + sred[1][i]) + sred[2][i]) + sred[3][i];
}

// ------------------------------------------------------------------------------------------
// depthwise 3x3 (pad 1) + bias -> pre ; GELU(pre) -> act ; pool[n][c] += sum_pixels act
// (fm:304-309 + the AdaptiveAvgPool2d of se_1, fm:253-260).  grid = (pixel chunks, N).
template <int LPP>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ in, int ldi,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    float* __restrict__ pre, float* __restrict__ act,
                                                    float* __restrict__ pool, int H, int W) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    const int q = threadIdx.x % LPP, n = blockIdx.y;
    float wr[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[k][t] = w[(q * 4 + k) * 9 + t];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + q * 4);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    const int HW = H * W;
    for (int p = blockIdx.x * PPB + threadIdx.x / LPP; p < HW; p += gridDim.x * PPB) {
        const int y = p / W, x = p % W;
        f32x4 a = bv;
        // 9 loads from clamped coordinates issued together, masked afterwards (a conditional load per tap is a
        // branch + load + full wait: 9 serialised latencies per pixel); same summation order as before
        f32x4 nb[9];
        float mk[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                mk[dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.f : 0.f;
                const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                nb[dy * 3 + dx] = *reinterpret_cast<const f32x4*>(in + ((long long)(n * H + yc) * W + xc) * ldi + q * 4);
            }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += (mk[t] != 0.f ? nb[t][k] : 0.f) * wr[k][t];
        f32x4 gl;
#pragma unroll
        for (int k = 0; k < 4; ++k) gl[k] = gelu_f(a[k]);
        const long long o = ((long long)n * HW + p) * C + q * 4;
        *reinterpret_cast<f32x4*>(pre + o) = a;
        *reinterpret_cast<f32x4*>(act + o) = gl;
        psum += gl;
    }
    if (pool != nullptr) {
        // deterministic: fixed-order sum over the block's pixel lanes, one partial row per block;
        // se_fwd adds the rows in block order (no atomics on the forward path)
        __shared__ float part[PPB][C];
#pragma unroll
        for (int k = 0; k < 4; ++k) part[threadIdx.x / LPP][q * 4 + k] = psum[k];
        __syncthreads();
        if (threadIdx.x < C) {
            float a = 0.f;
            for (int r = 0; r < PPB; ++r) a += part[r][threadIdx.x];
            pool[((long long)n * gridDim.x + blockIdx.x) * C + threadIdx.x] = a;
        }
    }
}

// depthwise 3x3 backward: gin = dgrad(gd), dw[c][tap] += sum gd*in(shifted), db[c] += sum gd.
template <int LPP>
__global__ __launch_bounds__(256) void dw_bwd_kernel(const float* __restrict__ gd, const float* __restrict__ in,
                                                    int ldi, const float* __restrict__ w, float* __restrict__ gin,
                                                    float* __restrict__ parts, int H, int W) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    __shared__ float sred[4][C * 10];
    const int q = threadIdx.x % LPP, n = blockIdx.y;
    float wr[4][9], pw[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) { wr[k][t] = w[(q * 4 + k) * 9 + t]; pw[k][t] = 0.f; }
    f32x4 pb = {0.f, 0.f, 0.f, 0.f};
    const int HW = H * W;
    for (int p = blockIdx.x * PPB + threadIdx.x / LPP; p < HW; p += gridDim.x * PPB) {
        const int y = p / W, x = p % W;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gd + ((long long)n * HW + p) * C + q * 4);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        // all 18 neighbour loads are issued up front from CLAMPED coordinates and masked afterwards: a conditional
        // load per tap compiles to a branch + load + full wait, i.e. 18 serialised memory latencies per pixel
        f32x4 gn[9], in9[9];
        float mg[9], mi[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int t = dy * 3 + dx;
                const int gy = y - dy + 1, gxx = x - dx + 1;           // dgrad: gin[y][x] += gd[y-dy+1][x-dx+1] * w[dy][dx]
                const int iy = y + dy - 1, ix = x + dx - 1;            // wgrad: dw[dy][dx] += gd[y][x] * in[y+dy-1][x+dx-1]
                mg[t] = (gy >= 0 && gy < H && gxx >= 0 && gxx < W) ? 1.f : 0.f;
                mi[t] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? 1.f : 0.f;
                const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gxx, 0), W - 1);
                const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
                gn[t] = *reinterpret_cast<const f32x4*>(gd + ((long long)(n * H + gyc) * W + gxc) * C + q * 4);
                in9[t] = *reinterpret_cast<const f32x4*>(in + ((long long)(n * H + iyc) * W + ixc) * ldi + q * 4);
            }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] += (mg[t] != 0.f ? gn[t][k] : 0.f) * wr[k][t];
                pw[k][t] += g0[k] * (mi[t] != 0.f ? in9[t][k] : 0.f);
            }
        *reinterpret_cast<f32x4*>(gin + ((long long)n * HW + p) * C + q * 4) = a;
        pb += g0;
    }
    // per-workgroup partials [block][C*10] (9 taps per channel, then the bias sums), fixed order; rows_sum_kernel adds them
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float a = wave_rows_sum<LPP>(pw[k][t]);
            if (lane < LPP) sred[wave][(q * 4 + k) * 9 + t] = a;
        }
        const float b = wave_rows_sum<LPP>(pb[k]);
        if (lane < LPP) sred[wave][C * 9 + q * 4 + k] = b;
    }
    __syncthreads();
    float* dst = parts + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (C * 10);
    for (int i = threadIdx.x; i < C * 10; i += 256) dst[i] = ((sred[0][i] + sred[1][i]) + sred[2][i]) + sred[3][i];
}

// ------------------------------------------------------------------------------------------
// squeeze-excite MLP (se_1, fm:253-260): s = sigmoid(W2 relu(W1 m + b1) + b2), m = pool/HW.
// One block per sample.
__global__ __launch_bounds__(256) void se_fwd_kernel(const float* __restrict__ pool, int nparts, float invHW,
                                                    const float* __restrict__ W1, const float* __restrict__ b1,
                                                    const float* __restrict__ W2, const float* __restrict__ b2,
                                                    float* __restrict__ m, float* __restrict__ z1, float* __restrict__ s,
                                                    int C) {
    __shared__ float sm[256], sz[128], part[256];
    const int n = blockIdx.x, Ch = C / 2;
    // fixed-order sum of the per-workgroup pool partials: 256/C segments per channel, 4 independent
    // accumulators each, then a serial pass over the segments (deterministic, no atomics)
    {
        const int nseg = 256 / C, c = threadIdx.x % C, seg = threadIdx.x / C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (seg < nseg) {
            const float* pp = pool + (long long)n * nparts * C + c;
            int r = seg;
            for (; r + 3 * nseg < nparts; r += 4 * nseg) {
                a0 += pp[(long long)r * C];
                a1 += pp[(long long)(r + nseg) * C];
                a2 += pp[(long long)(r + 2 * nseg) * C];
                a3 += pp[(long long)(r + 3 * nseg) * C];
            }
            for (; r < nparts; r += nseg) a0 += pp[(long long)r * C];
        }
        part[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (threadIdx.x < C) {
            float a = 0.f;
            for (int g = 0; g < nseg; ++g) a += part[g * C + threadIdx.x];
            sm[threadIdx.x] = a * invHW; m[n * C + threadIdx.x] = sm[threadIdx.x];
        }
    }
    __syncthreads();
    // the two mat-vecs: one wave per output row, lanes over the reduction index (coalesced rows of W)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < Ch; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += W1[j * C + c] * sm[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) {
            a += b1[j];
            a = a > 0.f ? a : 0.f;
            sz[j] = a; z1[n * Ch + j] = a;
        }
    }
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        float a = 0.f;
        for (int j = lane; j < Ch; j += 64) a += W2[c * Ch + j] * sz[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) s[n * C + c] = 1.f / (1.f + __expf(-(a + b2[c])));
    }
}

// SE backward, one block per sample: given gs = dL/ds: gm = dL/dm, and the per-sample vectors d2 (C) / d1 (C/2) go to
// `scratch` (n, C + C/2); se_bwd_params_kernel then adds the samples' outer products in sample order (deterministic).
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ s,
                                                    const float* __restrict__ z1, const float* __restrict__ m,
                                                    const float* __restrict__ W1, const float* __restrict__ W2,
                                                    float* __restrict__ gm, float* __restrict__ scratch, int N, int C) {
    __shared__ float d2[256], d1[128], sz[128], sm[256];
    const int Ch = C / 2;
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sv = s[n * C + c];
        d2[c] = gs[n * C + c] * sv * (1.f - sv);
        sm[c] = m[n * C + c];
    }
    for (int j = threadIdx.x; j < Ch; j += 256) sz[j] = z1[n * Ch + j];
    __syncthreads();
    for (int j = threadIdx.x; j < Ch; j += 256) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += W2[c * Ch + j] * d2[c];
        d1[j] = sz[j] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < Ch; ++j) a += W1[j * C + c] * d1[j];
        gm[n * C + c] = a;
        scratch[n * (C + Ch) + c] = d2[c];
    }
    for (int j = threadIdx.x; j < Ch; j += 256) scratch[n * (C + Ch) + C + j] = d1[j];
}

__global__ __launch_bounds__(256) void se_bwd_params_kernel(const float* __restrict__ scratch, const float* __restrict__ z1,
                                                           const float* __restrict__ m, float* __restrict__ dW1,
                                                           float* __restrict__ db1, float* __restrict__ dW2,
                                                           float* __restrict__ db2, int N, int C) {
    const int Ch = C / 2, S = C + Ch;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < C * Ch) {
        float a2 = 0.f, a1 = 0.f;
        for (int n = 0; n < N; ++n) {
            a2 += scratch[n * S + e / Ch] * z1[n * Ch + e % Ch];          // W2: (C, Ch)
            a1 += scratch[n * S + C + e / C] * m[n * C + e % C];          // W1: (Ch, C)
        }
        dW2[e] += a2;
        dW1[e] += a1;
    }
    if (e < C) { float a = 0.f; for (int n = 0; n < N; ++n) a += scratch[n * S + e]; db2[e] += a; }
    if (e < Ch) { float a = 0.f; for (int n = 0; n < N; ++n) a += scratch[n * S + C + e]; db1[e] += a; }
}

// out[n,p,0:C] = xi*s[n] ; out[n,p,C:2C] = xe*s[n]      (fm:312-315 without the cat temp)
template <int LPP>
__global__ __launch_bounds__(256) void scale_cat_kernel(const float* __restrict__ xi, const float* __restrict__ xe,
                                                       const float* __restrict__ s, float* __restrict__ out,
                                                       int HW, long long npix) {
    constexpr int C = LPP * 4, L2 = LPP * 2, PPB = 256 / L2;
    const int q = threadIdx.x % L2;
    for (long long p = blockIdx.x * (long long)PPB + threadIdx.x / L2; p < npix; p += (long long)gridDim.x * PPB) {
        const int n = (int)(p / HW);
        const int c = (q % LPP) * 4;
        const float* src = (q < LPP ? xi : xe) + p * C + c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + n * C + c);
        *reinterpret_cast<f32x4*>(out + p * 2 * C + q * 4) = v * sv;
    }
}

// parts[n][block][c] = sum_p(block) gxs[n,p,c]*xi[n,p,c] + gxs[n,p,C+c]*xe[n,p,c]   ; grid = (chunks, N)
template <int LPP>
__global__ __launch_bounds__(256) void gs_reduce_kernel(const float* __restrict__ gxs, const float* __restrict__ xi,
                                                       const float* __restrict__ xe, float* __restrict__ parts, int HW) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    __shared__ float sred[4][C];
    const int q = threadIdx.x % LPP, n = blockIdx.y;
    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
    for (int p = blockIdx.x * PPB + threadIdx.x / LPP; p < HW; p += gridDim.x * PPB) {
        const long long pix = (long long)n * HW + p;
        const f32x4 gi = *reinterpret_cast<const f32x4*>(gxs + pix * 2 * C + q * 4);
        const f32x4 ge = *reinterpret_cast<const f32x4*>(gxs + pix * 2 * C + C + q * 4);
        const f32x4 a = *reinterpret_cast<const f32x4*>(xi + pix * C + q * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xe + pix * C + q * 4);
        ps += gi * a + ge * b;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = wave_rows_sum<LPP>(ps[k]);
        if (lane < LPP) sred[wave][q * 4 + k] = a;
    }
    __syncthreads();
    if (threadIdx.x < C)
        parts[((long long)n * gridDim.x + blockIdx.x) * C + threadIdx.x] =
            ((sred[0][threadIdx.x] + sred[1][threadIdx.x]) + sred[2][threadIdx.x]) + sred[3][threadIdx.x];
}

// gs[n][c] = sum_blocks parts[n][block][c]: one wave per 64 / C samples' worth of columns -- lane (r, c) adds blocks r, r + R,
// ... in order, a fixed xor-shuffle tree over r combines them (R = 64 / C rows of lanes; C <= 64) -- deterministic
__global__ __launch_bounds__(256) void gs_sum_kernel(const float* __restrict__ parts, int nblk, int C, int total,
                                                    float* __restrict__ gs) {
    // 256 threads = (256 / C) block-rows x C columns of ONE sample (C in {16, 32, 64, 128}: C <= 256)
    const int n = blockIdx.x, c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    __shared__ float sh[256];
    float a = 0.f;
    for (int b = r; b < nblk; b += R) a += parts[((long long)n * nblk + b) * C + c];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = sh[threadIdx.x];
        for (int k = 1; k < R; ++k) t += sh[k * C + threadIdx.x];
        gs[n * C + threadIdx.x] = t;
    }
}

// gdwe = (gxs_e*s + gm/HW) * gelu'(dwe) ;  gxi_acc (+)= gxs_i*s
template <int LPP>
__global__ __launch_bounds__(256) void egaca_bwd_elem_kernel(const float* __restrict__ gxs, const float* __restrict__ s,
                                                            const float* __restrict__ gm, float invHW,
                                                            const float* __restrict__ dwe, float* __restrict__ gdwe,
                                                            float* __restrict__ gxi, int accXi, int HW, long long npix) {
    constexpr int C = LPP * 4, PPB = 256 / LPP;
    const int q = threadIdx.x % LPP;
    for (long long p = blockIdx.x * (long long)PPB + threadIdx.x / LPP; p < npix; p += (long long)gridDim.x * PPB) {
        const int n = (int)(p / HW);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + n * C + q * 4);
        const f32x4 gmv = *reinterpret_cast<const f32x4*>(gm + n * C + q * 4);
        const f32x4 gi = *reinterpret_cast<const f32x4*>(gxs + p * 2 * C + q * 4);
        const f32x4 ge = *reinterpret_cast<const f32x4*>(gxs + p * 2 * C + C + q * 4);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dwe + p * C + q * 4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (ge[k] * sv[k] + gmv[k] * invHW) * gelu_d(dv[k]);
        *reinterpret_cast<f32x4*>(gdwe + p * C + q * 4) = o;
        f32x4 xi = gi * sv;
        float* xo = gxi + p * C + q * 4;
        if (accXi) xi += *reinterpret_cast<const f32x4*>(xo);
        *reinterpret_cast<f32x4*>(xo) = xi;
    }
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = in[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = gelu_f(v[k]);
        out[i] = o;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ in,
                                                      f32x4* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = in[i], gv = g[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = gv[k] * gelu_d(v[k]);
        out[i] = o;
    }
}

// parts[block][c] = sum_p(block) g[p][c]  (bias gradient of ConvTranspose2d); C/4 a power of two <= 256.
// Fixed order: every thread's pixels in sequence, the block's pixel rows in sequence, rows_sum_kernel over the blocks.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, int ldg, float* __restrict__ parts,
                                                    int C, long long npix) {
    __shared__ f32x4 sb[256];
    const int Q = C / 4, PPB = 256 / Q;
    const int q = threadIdx.x % Q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long p = blockIdx.x * (long long)PPB + threadIdx.x / Q; p < npix; p += (long long)gridDim.x * PPB)
        acc += *reinterpret_cast<const f32x4*>(g + p * ldg + q * 4);
    sb[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < Q) {
        f32x4 a = sb[threadIdx.x];
        for (int r = 1; r < PPB; ++r) a += sb[r * Q + threadIdx.x];
        *reinterpret_cast<f32x4*>(parts + (long long)blockIdx.x * C + threadIdx.x * 4) = a;
    }
}

// beta/gamma fold-back (see refid_hip.h): per row r, Gf / gbf = gradient of the FOLDED weights of this backward:
//   dscale[r] += sum_k W[r][k]*Gf[r][k] + b[r]*gbf[r] ;  G[r][:] += scale[r]*Gf[r][:] ; gb[r] += scale[r]*gbf[r]
__global__ __launch_bounds__(64) void fold_back_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                      const float* __restrict__ scale, const float* __restrict__ Gf,
                                                      const float* __restrict__ gbf, float* __restrict__ G,
                                                      float* __restrict__ gb, float* __restrict__ dscale, int K) {
    const int r = blockIdx.x;
    const float sc = scale[r];
    float a = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) {
        const float gf = Gf[(long long)r * K + k];
        a += W[(long long)r * K + k] * gf;
        G[(long long)r * K + k] += sc * gf;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) {
        dscale[r] += a + b[r] * gbf[r];
        gb[r] += sc * gbf[r];
    }
}

__global__ __launch_bounds__(256) void mul_vec_kernel(const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

int blocks_for(long long items, int per_block, int cap = 2048) {
    long long b = (items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define LPP_DISPATCH(C, CALL)                                                     \
    switch (C) {                                                                  \
        case 16: { constexpr int LPP = 4; CALL; break; }                          \
        case 32: { constexpr int LPP = 8; CALL; break; }                          \
        case 64: { constexpr int LPP = 16; CALL; break; }                         \
        case 128: { constexpr int LPP = 32; CALL; break; }                        \
        default: refid_set_error("egaca: unsupported channel count %d (16/32/64/128)", C); return 1; \
    }

extern "C" int refid_layernorm2d_fwd(const float* x, int ld_x, const float* w, const float* b, float* out,
                                     int ld_out, long long npix, int c, float eps, void* stream) {
    REFID_CHECK(x && w && b && out && npix > 0, "layernorm2d_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    LPP_DISPATCH(c, hipLaunchKernelGGL(ln_fwd_kernel<LPP>, dim3(blocks_for(npix, 256 / LPP)), dim3(256), 0, st, x,
                                       ld_x, w, b, out, ld_out, npix, eps));
    REFID_LAUNCH_CHECK("layernorm2d_fwd");
    return 0;
}

extern "C" int refid_layernorm2d_bwd_parts(long long npix, int c) {
    if (c != 16 && c != 32 && c != 64 && c != 128) return -1;
    return blocks_for(npix, 256 / (c / 4), 512);
}

extern "C" int refid_layernorm2d_bwd(const float* g, int ld_g, const float* x, int ld_x, const float* w, float* gx,
                                     int ld_gx, const float* res, int ld_res, float* dw, float* db, float* parts,
                                     long long npix, int c, float eps, void* stream) {
    REFID_CHECK(g && x && w && gx && dw && db && parts && npix > 0, "layernorm2d_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = refid_layernorm2d_bwd_parts(npix, c);
    LPP_DISPATCH(c, hipLaunchKernelGGL(ln_bwd_kernel<LPP>, dim3(nb), dim3(256), 0, st,
                                       g, ld_g, x, ld_x, w, gx, ld_gx, res, ld_res, parts, npix, eps));
    REFID_LAUNCH_CHECK("layernorm2d_bwd");
    return rows_sum_issue(parts, nb, 2 * c, dw, c, db, st, "layernorm2d_bwd/sum");
}

extern "C" int refid_dwconv_pool_parts(int h, int wd, int c) {
    if (c != 16 && c != 32 && c != 64 && c != 128) return -1;
    return blocks_for((long long)h * wd, 256 / (c / 4), 64);
}

extern "C" int refid_dwconv3x3_gelu_fwd(const float* in, int ld_in, const float* w, const float* b, float* pre,
                                        float* act, float* pool, int n, int h, int wd, int c, void* stream) {
    REFID_CHECK(in && w && b && pre && act && n > 0 && h > 0 && wd > 0, "dwconv3x3_gelu_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    LPP_DISPATCH(c, hipLaunchKernelGGL(dw_fwd_kernel<LPP>, dim3(refid_dwconv_pool_parts(h, wd, c), n),
                                       dim3(256), 0, st, in, ld_in, w, b, pre, act, pool, h, wd));
    REFID_LAUNCH_CHECK("dwconv3x3_gelu_fwd");
    return 0;
}

extern "C" int refid_dwconv3x3_bwd_parts(int h, int wd, int c) {
    if (c != 16 && c != 32 && c != 64 && c != 128) return -1;
    return blocks_for((long long)h * wd, 256 / (c / 4), 64);
}

extern "C" int refid_dwconv3x3_bwd(const float* gd, const float* in, int ld_in, const float* w, float* gin,
                                   float* dw, float* db, float* parts, int n, int h, int wd, int c, void* stream) {
    REFID_CHECK(gd && in && w && gin && dw && db && parts && n > 0, "dwconv3x3_bwd: bad arguments (parts is required)");
    hipStream_t st = (hipStream_t)stream;
    // the parameter gradients are reduced deterministically in two stages through the caller's scratch buffer
    // (n * refid_dwconv3x3_bwd_parts() * 10c floats)
    const int nb = refid_dwconv3x3_bwd_parts(h, wd, c);
    LPP_DISPATCH(c, hipLaunchKernelGGL(dw_bwd_kernel<LPP>, dim3(nb, n), dim3(256), 0, st, gd, in, ld_in, w, gin, parts, h, wd));
    REFID_LAUNCH_CHECK("dwconv3x3_bwd");
    return rows_sum_issue(parts, nb * n, c * 10, dw, c * 9, db, st, "dwconv3x3_bwd/sum");
}

extern "C" int refid_se_fwd(const float* pool, int n_parts, float inv_hw, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* m, float* z1, float* s, int n, int c,
                            void* stream) {
    REFID_CHECK(pool && w1 && b1 && w2 && b2 && m && z1 && s && n > 0 && n_parts > 0 && c > 0 && c <= 256 && c % 2 == 0 && 256 % c == 0,
                "se_fwd: bad arguments (c=%d; must divide 256)", c);
    hipLaunchKernelGGL(se_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, pool, n_parts, inv_hw, w1, b1, w2,
                       b2, m, z1, s, c);
    REFID_LAUNCH_CHECK("se_fwd");
    return 0;
}

extern "C" int refid_se_bwd(const float* gs, const float* s, const float* z1, const float* m, const float* w1,
                            const float* w2, float* gm, float* dw1, float* db1, float* dw2, float* db2, float* scratch,
                            int n, int c, void* stream) {
    REFID_CHECK(gs && s && z1 && m && w1 && w2 && gm && dw1 && db1 && dw2 && db2 && scratch && n > 0 && c <= 256,
                "se_bwd: bad arguments");
    hipLaunchKernelGGL(se_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, gs, s, z1, m, w1, w2, gm, scratch, n, c);
    REFID_LAUNCH_CHECK("se_bwd");
    hipLaunchKernelGGL(se_bwd_params_kernel, dim3(cdiv(c * (c / 2), 256)), dim3(256), 0, (hipStream_t)stream, scratch, z1, m,
                       dw1, db1, dw2, db2, n, c);
    REFID_LAUNCH_CHECK("se_bwd/params");
    return 0;
}

extern "C" int refid_scale_cat(const float* xi, const float* xe, const float* s, float* out, int n, int hw, int c,
                               void* stream) {
    REFID_CHECK(xi && xe && s && out && n > 0 && hw > 0, "scale_cat: bad arguments");
    const long long npix = (long long)n * hw;
    hipStream_t st = (hipStream_t)stream;
    LPP_DISPATCH(c, hipLaunchKernelGGL(scale_cat_kernel<LPP>, dim3(blocks_for(npix, 256 / (2 * LPP))), dim3(256), 0,
                                       st, xi, xe, s, out, hw, npix));
    REFID_LAUNCH_CHECK("scale_cat");
    return 0;
}

extern "C" int refid_egaca_gs_reduce_parts(int hw, int c) {
    if (c != 16 && c != 32 && c != 64 && c != 128) return -1;
    return blocks_for(hw, 256 / (c / 4), 128);
}

extern "C" int refid_egaca_gs_reduce(const float* gxs, const float* xi, const float* xe, float* gs, float* parts, int n,
                                     int hw, int c, void* stream) {
    REFID_CHECK(gxs && xi && xe && gs && parts && n > 0 && hw > 0, "egaca_gs_reduce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = refid_egaca_gs_reduce_parts(hw, c);
    LPP_DISPATCH(c, hipLaunchKernelGGL(gs_reduce_kernel<LPP>, dim3(nb, n), dim3(256), 0, st, gxs, xi, xe, parts, hw));
    REFID_LAUNCH_CHECK("egaca_gs_reduce");
    hipLaunchKernelGGL(gs_sum_kernel, dim3(n), dim3(256), 0, st, parts, nb, c, n * c, gs);
    REFID_LAUNCH_CHECK("egaca_gs_reduce/sum");
    return 0;
}

extern "C" int refid_egaca_bwd_elem(const float* gxs, const float* s, const float* gm, float inv_hw, const float* dwe,
                                    float* gdwe, float* gxi, int accumulate_xi, int n, int hw, int c, void* stream) {
    REFID_CHECK(gxs && s && gm && dwe && gdwe && gxi && n > 0 && hw > 0, "egaca_bwd_elem: bad arguments");
    const long long npix = (long long)n * hw;
    hipStream_t st = (hipStream_t)stream;
    LPP_DISPATCH(c, hipLaunchKernelGGL(egaca_bwd_elem_kernel<LPP>, dim3(blocks_for(npix, 256 / LPP)), dim3(256), 0, st,
                                       gxs, s, gm, inv_hw, dwe, gdwe, gxi, accumulate_xi, hw, npix));
    REFID_LAUNCH_CHECK("egaca_bwd_elem");
    return 0;
}

extern "C" int refid_gelu_fwd(const float* in, float* out, long long count, void* stream) {
    REFID_CHECK(in && out && count > 0 && count % 4 == 0, "gelu_fwd: bad arguments");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks_for(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)in, (f32x4*)out, count / 4);
    REFID_LAUNCH_CHECK("gelu_fwd");
    return 0;
}

extern "C" int refid_gelu_bwd(const float* g, const float* in, float* out, long long count, void* stream) {
    REFID_CHECK(g && in && out && count > 0 && count % 4 == 0, "gelu_bwd: bad arguments");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks_for(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)g, (const f32x4*)in, (f32x4*)out, count / 4);
    REFID_LAUNCH_CHECK("gelu_bwd");
    return 0;
}

extern "C" int refid_colsum_parts(long long npix, int c) {
    if (c < 4 || c > 1024 || c % 4 != 0 || 256 % (c / 4) != 0) return -1;
    return blocks_for(npix, 256 / (c / 4), 256);
}

extern "C" int refid_colsum(const float* g, int ld_g, float* db, float* parts, long long npix, int c, void* stream) {
    REFID_CHECK(g && db && parts && npix > 0 && c >= 4 && c <= 1024 && c % 4 == 0 && 256 % (c / 4) == 0,
                "colsum: bad arguments (c=%d; c/4 must divide 256)", c);
    const int nb = refid_colsum_parts(npix, c);
    hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, ld_g, parts, c, npix);
    REFID_LAUNCH_CHECK("colsum");
    return rows_sum_issue(parts, nb, c, db, c, nullptr, (hipStream_t)stream, "colsum/sum");
}

extern "C" int refid_rows_sum_defer(int on) {
    rows_queue.clear();                                     // (a queue left behind by a failed backward pass holds dead pointers)
    rows_defer = on != 0;
    return 0;
}

extern "C" int refid_rows_sum_flush(void* stream) {
    rows_defer = false;
    return rows_sum_flush((hipStream_t)stream);
}

extern "C" int refid_fold_back(const float* w, const float* b, const float* scale, const float* gw_folded,
                               const float* gb_folded, float* gw, float* gb, float* dscale, int rows, int k,
                               void* stream) {
    REFID_CHECK(w && b && scale && gw_folded && gb_folded && gw && gb && dscale && rows > 0 && k > 0,
                "fold_back: bad arguments");
    REFID_CHECK(gw_folded != gw && gb_folded != gb, "fold_back: the folded gradient must not alias the accumulated one");
    hipLaunchKernelGGL(fold_back_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, w, b, scale, gw_folded, gb_folded,
                       gw, gb, dscale, k);
    REFID_LAUNCH_CHECK("fold_back");
    return 0;
}

extern "C" int refid_mul_vec(const float* a, const float* b, float* out, int n, void* stream) {
    REFID_CHECK(a && b && out && n > 0, "mul_vec: bad arguments");
    hipLaunchKernelGGL(mul_vec_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    REFID_LAUNCH_CHECK("mul_vec");
    return 0;
}
