// Error plumbing, device queries, layout conversion, weight packing and small
// elementwise kernels of the REFID C ABI.
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void refid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* refid_last_error(void) { return g_err; }
extern "C" int refid_abi_version(void) { return REFID_ABI_VERSION; }
extern "C" int refid_experimental_tiles(void) {
    return 0;
}

extern "C" int refid_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return p.multiProcessorCount;
}

namespace {

// ---- NCHW -> NHWC (padded) : one block transposes a 32-pixel x C strip through LDS ----------
// (tCount > 1: the source is (n, tCount, C, HW) and the tCount slices are summed on the way -- the gradient of a tensor
// every time step reads, e.g. `head` in `pred(z_t + head)`, XXNet_final_attenfusion_arch.py:215)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                          long long srcBatchStride,
                                                          float* __restrict__ dst, int C, int HW,
                                                          int Cpad, int tCount, long long tStride, int nb) {
    // grid.x = pixel blocks of 64, grid.y = n.  nb > 0 (time-major form, tCount = 1): output sample n = t nb + b reads the
    // source's (b, t) block, at b srcBatchStride + t tStride
    __shared__ float tile[64][65];
    const int n = blockIdx.y;
    const long long srcOff = nb > 0 ? (long long)(n % nb) * srcBatchStride + (long long)(n / nb) * tStride
                                    : (long long)n * srcBatchStride;
    const int p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // ty in 0..3
    for (int c0 = 0; c0 < Cpad; c0 += 64) {
        // read: coalesced along pixels
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc, p = p0 + tx;
            float v = 0.f;
            if (c < C && p < HW) {
                const float* sp = src + srcOff + (long long)c * HW + p;
                for (int t = 0; t < tCount; ++t) v += sp[t * tStride];
            }
            tile[cc][tx] = v;
        }
        __syncthreads();
        // write: coalesced along channels
        for (int pp = ty; pp < 64; pp += 4) {
            const int c = c0 + tx, p = p0 + pp;
            if (c < Cpad && p < HW) dst[((long long)n * HW + p) * Cpad + c] = tile[tx][pp];
        }
        __syncthreads();
    }
}

// Thin tensors (C <= 4: event voxels 2 -> 4, the 3-channel output and its gradient): the 64-channel LDS transpose above moves
// 16 bytes per store instruction for them (0.06-0.13 of HBM).  Here a thread owns a PIXEL: C coalesced plane reads, one 16-byte
// NHWC access (a wave writes 1 KB contiguous) -- and the reverse.  Same sample addressing (time-major form, sum over T).
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, long long srcBatchStride,
                                                           f32x4* __restrict__ dst, int C, int HW, int tCount, long long tStride,
                                                           int nb) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const long long srcOff = nb > 0 ? (long long)(n % nb) * srcBatchStride + (long long)(n / nb) * tStride
                                    : (long long)n * srcBatchStride;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= C) break;
        const float* sp = src + srcOff + (long long)c * HW + p;
        float acc = 0.f;
        for (int t = 0; t < tCount; ++t) acc += sp[t * tStride];
        v[c] = acc;
    }
    dst[(long long)n * HW + p] = v;
}

__global__ __launch_bounds__(256) void nhwc4_to_nchw_kernel(const f32x4* __restrict__ src, float* __restrict__ dst,
                                                           long long dstBatchStride, int C, int HW, int nb, long long tStride) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const long long dstOff = nb > 0 ? (long long)(n % nb) * dstBatchStride + (long long)(n / nb) * tStride
                                    : (long long)n * dstBatchStride;
    const f32x4 v = src[(long long)n * HW + p];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= C) break;
        dst[dstOff + (long long)c * HW + p] = v[c];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int ld,
                                                          float* __restrict__ dst,
                                                          long long dstBatchStride, int C, int HW, int nb, long long tStride) {
    // nb > 0: source sample n = t nb + b (time-major) goes to the destination's (b, t) block
    __shared__ float tile[64][65];
    const int n = blockIdx.y;
    const long long dstOff = nb > 0 ? (long long)(n % nb) * dstBatchStride + (long long)(n / nb) * tStride
                                    : (long long)n * dstBatchStride;
    const int p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int pp = ty; pp < 64; pp += 4) {
            const int c = c0 + tx, p = p0 + pp;
            tile[pp][tx] = (c < C && p < HW) ? src[((long long)n * HW + p) * ld + c] : 0.f;
        }
        __syncthreads();
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc, p = p0 + tx;
            if (c < C && p < HW) dst[dstOff + (long long)c * HW + p] = tile[tx][cc];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void add_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                 f32x4* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        out[i] = a[i] + b[i];
    }
}

// out = in[0] + in[1] + ... + in[n-1], added in index order (deterministic).  Gradients that several time steps contribute to
// (the image branch's x_blocks, the final backward states) are summed ONCE after BPTT from the per-step tensors -- (n + 1)
// tensor passes instead of the 3 (n - 1) of n - 1 in-place accumulations, one launch instead of n - 1.
struct SumNArgs { const f32x4* in[REFID_SUM_MAX]; int n; };
__global__ __launch_bounds__(256) void sum_n_kernel(const SumNArgs a, f32x4* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 s = a.in[0][i];
        for (int k = 1; k < a.n; ++k) s += a.in[k][i];
        out[i] = s;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ y,
                                                     f32x4* __restrict__ out, float slope, int acc,
                                                     long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 gv = g[i], yv = y[i];
        f32x4 o = acc ? out[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += gv[k] * (yv[k] > 0.f ? 1.f : slope);
        out[i] = o;
    }
}

// ---- weight packing --------------------------------------------------------------------------
// packed[cls][chunk][tap][row][k] = W[src index] or 0.
struct PackArgs {
    const float* w; float* dst; const float* oscale;   // oscale: optional per-output-channel factor
    int role, O, I, KH, KW, KC, rows, rowsPad, K, nchunks, ntaps, ncls;
    int bf16;                  // write __bf16 (RNE) instead of float
};

__device__ __forceinline__ float pack_fetch(const PackArgs& p, int cls, int tap, int row, int k) {
    const int KK = p.KH * p.KW;
    switch (p.role) {
        case REFID_ROLE_FWD:            // rows = O, k = I ; W[o][i][tap]
            return p.w[((long long)row * p.I + k) * KK + tap] * (p.oscale ? p.oscale[row] : 1.f);
        case REFID_ROLE_DGRAD:          // rows = I, k = O ; flipped taps
            return p.w[((long long)k * p.I + row) * KK + (KK - 1 - tap)] * (p.oscale ? p.oscale[k] : 1.f);
        case REFID_ROLE_CONVT: {        // W is (I=Ci, O=Co, 2, 2); rows = (q, co), k = ci
            const int Co = p.O;
            const int qd = row / Co, co = row - qd * Co;
            return p.w[((long long)k * Co + co) * 4 + qd];
        }
        case REFID_ROLE_CONVT_DGRAD:    // rows = ci, k = co, tap = dy*2+dx
            return p.w[((long long)row * p.O + k) * 4 + tap];
        case REFID_ROLE_CONVT_DGRAD_PW: // rows = ci, k = (dy, dx, co): the patch GEMM of the pointwise tile
            return p.w[((long long)row * p.O + k % p.O) * 4 + k / p.O];
        case REFID_ROLE_WINO_FWD:       // U[xi=(i,j)] = sum_ab G[i][a] G[j][b] g[a][b]
        case REFID_ROLE_WINO_DGRAD: {
            const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
            const int i = tap >> 2, j = tap & 3;
            const bool fwd = p.role == REFID_ROLE_WINO_FWD;
            const float* g = fwd ? p.w + ((long long)row * p.I + k) * 9 : p.w + ((long long)k * p.I + row) * 9;
            float u = 0.f;
#pragma unroll
            for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    const float gv = fwd ? g[aa * 3 + bb] : g[(2 - aa) * 3 + (2 - bb)];   // dgrad: flipped taps
                    u += G[i][aa] * G[j][bb] * gv;
                }
            return u * (p.oscale ? p.oscale[fwd ? row : k] : 1.f);
        }
        case REFID_ROLE_DOWN_DGRAD: {   // W (O,I,4,4); rows = i, k = o; class (py,px), tap (ta,tb)
            const int py = cls >> 1, px = cls & 1, ta = tap >> 1, tb = tap & 1;
            const int ky = (ta == 0) ? (py ? 2 : 1) : (py ? 0 : 3);
            const int kx = (tb == 0) ? (px ? 2 : 1) : (px ? 0 : 3);
            return p.w[((long long)k * p.I + row) * 16 + ky * 4 + kx];
        }
    }
    return 0.f;
}

// (IDX: the element index type -- unsigned where the packing has < 2^31 elements: a 64-bit division by a run-time value costs
//  ~4x a 32-bit one, and the index decode is most of this kernel's instructions)
template <class IDX>
__device__ __forceinline__ void pack_elem(const PackArgs& p, IDX e) {
    IDX r = e;
    const int kk = r % p.KC; r /= p.KC;
    const int row = r % p.rowsPad; r /= p.rowsPad;
    const int tap = r % p.ntaps; r /= p.ntaps;
    const int chunk = r % p.nchunks;
    const int cls = r / p.nchunks;
    const int k = chunk * p.KC + kk;
    float v = 0.f;
    if (row < p.rows && k < p.K) v = pack_fetch(p, cls, tap, row, k);
    if (p.bf16) reinterpret_cast<__bf16*>(p.dst)[e] = (__bf16)v;
    else p.dst[e] = v;
}

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs p) {
    const long long total = (long long)p.ncls * p.nchunks * p.ntaps * p.rowsPad * p.KC;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) pack_elem<long long>(p, e);
}

// split-bf16 planes for conv_split.hip (8-channel sub-chunks, `planes` bf16 numbers per weight):
//   mode 0 (3x3 s1):        [chunk8][plane][tap 0..9][rowsPad][8]            the tenth tap is zero (tap pairs fill K = 16)
//   mode 1 (4x4 s2 forward): [chunk8][sy][sx][plane][tap (ty,tx)][rowsPad][8]  = W[row][k][2ty+sy][2tx+sx]  (2x2 input blocks)
//   mode 2 (its dgrad):      [class][chunk8][plane][tap (ta,tb)][rowsPad][8]   (REFID_ROLE_DOWN_DGRAD's classes / taps)
template <class IDX>
__device__ __forceinline__ void pack_split_elem(const PackArgs& p, int planes, int mode, IDX e) {
    const int ntp = mode == 0 ? p.ntaps + 1 : 4;
    const int nsub = mode == 1 ? 4 : 1;
    __bf16* dst = reinterpret_cast<__bf16*>(p.dst);
    {
        IDX r = e;
        const int k8 = r % 8; r /= 8;
        const int row = r % p.rowsPad; r /= p.rowsPad;
        const int tap = r % ntp; r /= ntp;
        const int plane = r % planes; r /= planes;
        const int sub = r % nsub; r /= nsub;
        const int chunk = r % p.nchunks;
        const int cls = r / p.nchunks;
        const int k = chunk * 8 + k8;
        float v = 0.f;
        if (row < p.rows && k < p.K) {
            if (mode == 0) { if (tap < p.ntaps) v = pack_fetch(p, 0, tap, row, k); }
            else if (mode == 1) v = pack_fetch(p, 0, (2 * (tap >> 1) + (sub >> 1)) * 4 + 2 * (tap & 1) + (sub & 1), row, k);
            else v = pack_fetch(p, cls, tap, row, k);
        }
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        dst[e] = plane == 0 ? h : (plane == 1 ? m : l);
    }
}

// The same layouts with TWO fp16 planes h = rne16(v 2^eW), l = rne16(v 2^eW - h) behind a 64-byte header (int eW at byte 0, as
// the Winograd x three packing below): conv_split.hip's three-fp16-product form (refid_conv2d algo 4, mfma_terms 19).
template <class IDX>
__device__ __forceinline__ void pack_split_f16_elem(const PackArgs& p, int mode, IDX e) {
    const int ntp = mode == 0 ? p.ntaps + 1 : 4;
    const int nsub = mode == 1 ? 4 : 1;
    const int eW = *reinterpret_cast<const int*>(p.dst);
    _Float16* dst = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(p.dst) + 64);
    IDX r = e;
    const int k8 = r % 8; r /= 8;
    const int row = r % p.rowsPad; r /= p.rowsPad;
    const int tap = r % ntp; r /= ntp;
    const int plane = r % 2; r /= 2;
    const int sub = r % nsub; r /= nsub;
    const int chunk = r % p.nchunks;
    const int cls = r / p.nchunks;
    const int k = chunk * 8 + k8;
    float v = 0.f;
    if (row < p.rows && k < p.K) {
        if (mode == 0) { if (tap < p.ntaps) v = pack_fetch(p, 0, tap, row, k); }
        else if (mode == 1) v = pack_fetch(p, 0, (2 * (tap >> 1) + (sub >> 1)) * 4 + 2 * (tap & 1) + (sub & 1), row, k);
        else v = pack_fetch(p, cls, tap, row, k);
    }
    v = ldexpf(v, eW);
    const _Float16 h = (_Float16)v;
    dst[e] = plane == 0 ? h : (_Float16)(v - (float)h);
}

__global__ __launch_bounds__(256) void pack_split_f16_kernel(const PackArgs p, int mode) {
    const long long total = (long long)p.ncls * p.nchunks * (mode == 1 ? 4 : 1) * 2 * (mode == 0 ? p.ntaps + 1 : 4) * p.rowsPad * 8;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) pack_split_f16_elem<long long>(p, mode, e);
}

__global__ __launch_bounds__(256) void pack_split_kernel(const PackArgs p, int planes, int mode) {
    const long long total = (long long)p.ncls * p.nchunks * (mode == 1 ? 4 : 1) * planes * (mode == 0 ? p.ntaps + 1 : 4) * p.rowsPad * 8;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) pack_split_elem<long long>(p, planes, mode, e);
}

// 1x1 (conv_pw.hip, six products): [chunk16][plane][rowsPad][16]; the 16 channels of a row are stored as the two MFMA K
// halves of the pointwise tile's lanes: slot 8h + t = channel 4h + t (t < 4) or 8 + 4h + (t - 4)
template <class IDX>
__device__ __forceinline__ void pack_pw6_elem(const PackArgs& p, int planes, IDX e) {
    __bf16* dst = reinterpret_cast<__bf16*>(p.dst);
    {
        IDX r = e;
        const int slot = r % 16; r /= 16;
        const int row = r % p.rowsPad; r /= p.rowsPad;
        const int plane = r % planes;
        const int chunk = r / planes;
        const int h = slot >> 3, t = slot & 7;
        const int k = chunk * 16 + (t < 4 ? 4 * h + t : 8 + 4 * h + (t - 4));
        float v = 0.f;
        if (row < p.rows && k < p.K) v = pack_fetch(p, 0, 0, row, k);
        const __bf16 hh = (__bf16)v;
        const float r1 = v - (float)hh;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        dst[e] = plane == 0 ? hh : (plane == 1 ? m : l);
    }
}

__global__ __launch_bounds__(256) void pack_pw6_kernel(const PackArgs p, int planes) {
    const long long total = (long long)((p.K + 15) / 16) * planes * p.rowsPad * 16;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) pack_pw6_elem<long long>(p, planes, e);
}

// Winograd-domain weights U = G g G^T as three bf16 planes for conv_wino6.hip: [chunk16][xi][plane][rowsPad][16].
// One thread per WEIGHT (chunk, row, k16): nine loads, the 16 transform points once, 48 two-byte stores -- a wave's 64 threads
// (4 rows x 16 k) write 128 contiguous bytes per (xi, plane).  (The first form had one thread per OUTPUT element: 48 x the
// loads and transforms and five 64-bit divisions each -- 0.05 of HBM for the model's 238 MB of these planes.)
__device__ __forceinline__ void pack_wino6_weight(const PackArgs& p, unsigned f) {
    const unsigned k16 = f & 15, r = f >> 4;
    const unsigned row = r % (unsigned)p.rowsPad, chunk = r / (unsigned)p.rowsPad;
    const int k = chunk * 16 + k16;
    float u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = 0.f;
    if ((int)row < p.rows && k < p.K) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const bool fwd = p.role == REFID_ROLE_WINO_FWD;
        const float* g = fwd ? p.w + ((long long)row * p.I + k) * 9 : p.w + ((long long)k * p.I + row) * 9;
        float gv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) gv[t] = fwd ? g[t] : g[8 - t];              // dgrad: flipped taps
        const float sc = p.oscale ? p.oscale[fwd ? row : k] : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;                                               // (pack_fetch's order of additions)
#pragma unroll
                for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) acc += G[i][aa] * G[j][bb] * gv[aa * 3 + bb];
                u[i * 4 + j] = acc * sc;
            }
    }
    const long long plane = (long long)p.rowsPad * 16;
    __bf16* dst = reinterpret_cast<__bf16*>(p.dst) + ((long long)chunk * 48 * p.rowsPad + row) * 16 + k16;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        const float v = u[xi];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        dst[(xi * 3 + 0) * plane] = h;
        dst[(xi * 3 + 1) * plane] = m;
        dst[(xi * 3 + 2) * plane] = l;
    }
}

__global__ __launch_bounds__(256) void pack_wino6_kernel(const PackArgs p) {
    const unsigned total = (unsigned)p.nchunks * p.rowsPad * 16;                  // weights (padding included)
    for (unsigned f = blockIdx.x * 256u + threadIdx.x; f < total; f += gridDim.x * 256u) pack_wino6_weight(p, f);
}

// ---- Winograd-domain weights for the THREE-fp16-product form of conv_wino6.hip (refid_conv2d algo 5, mfma_terms 3) ----------
// fp16 keeps 11 significand bits, so two planes h = rne16(v), l = rne16(v - h) carry 22 bits and  u v = uh vh + uh vl + ul vh
// + O(2^-22 |u v|)  needs three MFMAs and two U planes where the bf16 form needs six and three.  The price is fp16's RANGE
// (2^-14 .. 2^16; the low plane of a value below 2^-3 is subnormal): both operands travel multiplied by an exact power of two.
// U's is per packing: 2^eU with max |U| 2^eU in [2^12, 2^15) -- |U| <= 2.25 max |w| --, found by a reduction over the tensor
// (pack_absmax_*: one workgroup per tensor, max is order independent => deterministic) and kept in the packing's 64-byte HEADER
// (int eU at byte 0), where the pack kernel and the conv tile read it; the conv tile undoes 2^eU together with its own per-tile
// scale of V in its output transform.  Layout after the header: [chunk16][xi][plane h/l][rowsPad][16] fp16.
constexpr int W3H_HEADER = 64;

__device__ __forceinline__ int wino3h_scale_exp(float maxabs) {
    if (!(maxabs > 0.f)) return 0;
    const int ew = (int)(__float_as_uint(maxabs) >> 23) - 127;                 // max |w| < 2^(ew+1), max |U| < 2^(ew+2.17)
    const int e = 12 - ew;
    return e < -110 ? -110 : (e > 110 ? 110 : e);
}

__device__ __forceinline__ void pack_absmax_block(const PackArgs& p) {       // one workgroup (any size that is a multiple of 64)
    __shared__ float part[16];
    const long long total = (long long)p.O * p.I * p.KH * p.KW;
    const int per_o = p.I * p.KH * p.KW;
    float m = 0.f;
    if (p.oscale == nullptr && total % 4 == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15) == 0) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(p.w);
        for (long long e = threadIdx.x; e < total / 4; e += blockDim.x) {
            const f32x4 v = w4[e];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        for (long long e = threadIdx.x; e < total; e += blockDim.x)
            m = fmaxf(m, fabsf(p.w[e] * (p.oscale ? p.oscale[e / per_o] : 1.f)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmaxf(m, part[k]);
        int* hdr = reinterpret_cast<int*>(p.dst);            // the whole header is defined: eU, then zeros
        hdr[0] = wino3h_scale_exp(m);
        for (int k = 1; k < W3H_HEADER / 4; ++k) hdr[k] = 0;
    }
}

__global__ __launch_bounds__(1024) void pack_absmax_kernel(const PackArgs p) { pack_absmax_block(p); }

// one thread per WEIGHT, as pack_wino6_weight: nine loads, the 16 transform points once, 32 two-byte stores
__device__ __forceinline__ void pack_wino3h_weight(const PackArgs& p, unsigned f) {
    const unsigned k16 = f & 15, r = f >> 4;
    const unsigned row = r % (unsigned)p.rowsPad, chunk = r / (unsigned)p.rowsPad;
    const int k = chunk * 16 + k16;
    const int eU = *reinterpret_cast<const int*>(p.dst);
    float u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = 0.f;
    if ((int)row < p.rows && k < p.K) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const bool fwd = p.role == REFID_ROLE_WINO_FWD;
        const float* g = fwd ? p.w + ((long long)row * p.I + k) * 9 : p.w + ((long long)k * p.I + row) * 9;
        float gv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) gv[t] = fwd ? g[t] : g[8 - t];              // dgrad: flipped taps
        const float sc = p.oscale ? p.oscale[fwd ? row : k] : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;                                               // (pack_fetch's order of additions: the fp32 U)
#pragma unroll
                for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) acc += G[i][aa] * G[j][bb] * gv[aa * 3 + bb];
                u[i * 4 + j] = ldexpf(acc * sc, eU);                           // exact: a power of two, inside the fp32 range
            }
    }
    const long long plane = (long long)p.rowsPad * 16;
    _Float16* dst = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(p.dst) + W3H_HEADER) +
                    ((long long)chunk * 32 * p.rowsPad + row) * 16 + k16;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        const float v = u[xi];
        const _Float16 h = (_Float16)v;
        dst[(xi * 2 + 0) * plane] = h;
        dst[(xi * 2 + 1) * plane] = (_Float16)(v - (float)h);
    }
}

__global__ __launch_bounds__(256) void pack_wino3h_kernel(const PackArgs p) {
    const unsigned total = (unsigned)p.nchunks * p.rowsPad * 16;
    for (unsigned f = blockIdx.x * 256u + threadIdx.x; f < total; f += gridDim.x * 256u) pack_wino3h_weight(p, f);
}

// ---- all packings of a model in ONE launch (refid_pack_batch): the table lives in device memory, a workgroup finds its
// entry by binary search over the entries' first block.  ~220 dependent 6-20 us launches per optimiser step become one.
struct PackEntry {
    PackArgs p;
    int kind;                  // 0 pack_kernel (fp32 / bf16), 1 split (modes 0-2), 2 1x1 split, 3 Winograd x six, 4 out = a * b (vectors),
                               // 5 Winograd x three fp16 products, 6 split tile with two fp16 planes (modes 0-2): both need
                               // refid_pack_batch_prepass before the batch (their scale exponents)
    int planes, mode;
    int blk0, nblk;            // this entry's workgroups: [blk0, blk0 + nblk)
    long long total;           // elements
};

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackEntry* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                       // last entry with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackEntry& en = table[lo];
    const PackArgs p = en.p;
    const int kind = en.kind, planes = en.planes, mode = en.mode;
    const long long total = en.total, stride = (long long)en.nblk * 256;
    const long long e0 = (long long)(blockIdx.x - en.blk0) * 256 + threadIdx.x;
    if (kind == 3) {                                         // (total = weights, < 2^31: refid_pack_entry_fill)
        for (unsigned f = (unsigned)e0; f < (unsigned)total; f += (unsigned)stride) pack_wino6_weight(p, f);
    } else if (kind == 5) {
        for (unsigned f = (unsigned)e0; f < (unsigned)total; f += (unsigned)stride) pack_wino3h_weight(p, f);
    } else if (kind == 6) {
        for (long long e = e0; e < total; e += stride) pack_split_f16_elem<long long>(p, mode, e);
    } else if (total < 0x7fffffffLL) {
        for (unsigned e = (unsigned)e0; e < (unsigned)total; e += (unsigned)stride) {
            switch (kind) {
                case 0: pack_elem<unsigned>(p, e); break;
                case 1: pack_split_elem<unsigned>(p, planes, mode, e); break;
                case 2: pack_pw6_elem<unsigned>(p, planes, e); break;
                default: p.dst[e] = p.w[e] * p.oscale[e]; break;
            }
        }
    } else {
        for (long long e = e0; e < total; e += stride) {
            switch (kind) {
                case 0: pack_elem<long long>(p, e); break;
                case 1: pack_split_elem<long long>(p, planes, mode, e); break;
                case 2: pack_pw6_elem<long long>(p, planes, e); break;
                default: p.dst[e] = p.w[e] * p.oscale[e]; break;
            }
        }
    }
}

// the scale exponents of every kind-5 record of a table (one workgroup per record; the others leave at once)
__global__ __launch_bounds__(1024) void pack_prepass_kernel(const PackEntry* __restrict__ table, int n) {
    const PackEntry& en = table[blockIdx.x];
    if (en.kind != 5 && en.kind != 6) return;
    const PackArgs p = en.p;
    pack_absmax_block(p);
}

int pack_geometry(int role, int o, int i, int kh, int kw, int kc, int bn, PackArgs* p) {
    p->role = role; p->O = o; p->I = i; p->KH = kh; p->KW = kw; p->KC = kc;
    p->ncls = 1;
    switch (role) {
        case REFID_ROLE_FWD: p->rows = o; p->K = i; p->ntaps = kh * kw; break;
        case REFID_ROLE_DGRAD: p->rows = i; p->K = o; p->ntaps = kh * kw; break;
        case REFID_ROLE_CONVT: p->rows = 4 * o; p->K = i; p->ntaps = 1; break;
        case REFID_ROLE_CONVT_DGRAD: p->rows = i; p->K = o; p->ntaps = 4; break;
        case REFID_ROLE_CONVT_DGRAD_PW: p->rows = i; p->K = 4 * o; p->ntaps = 1; break;
        case REFID_ROLE_DOWN_DGRAD: p->rows = i; p->K = o; p->ntaps = 4; p->ncls = 4; break;
        case REFID_ROLE_WINO_FWD: p->rows = o; p->K = i; p->ntaps = 16; break;
        case REFID_ROLE_WINO_DGRAD: p->rows = i; p->K = o; p->ntaps = 16; break;
        default: return 1;
    }
    p->rowsPad = round_up(p->rows, bn);
    p->nchunks = cdiv(p->K, kc);
    return 0;
}

int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t refid_packed_weight_floats(int role, int o, int i, int kh, int kw, int kc, int bn) {
    PackArgs p;
    if (pack_geometry(role, o, i, kh, kw, kc, bn, &p)) return 0;
    return (size_t)p.ncls * p.nchunks * p.ntaps * p.rowsPad * p.KC;
}

static int pack_impl(const float* w, const float* oscale, float* packed, int role, int o, int i, int kh,
                     int kw, int kc, int bn, void* stream);

extern "C" int refid_pack_conv_weights(const float* w, float* packed, int role, int o, int i, int kh,
                                       int kw, int kc, int bn, void* stream) {
    return pack_impl(w, nullptr, packed, role, o, i, kh, kw, kc, bn, stream);
}

extern "C" int refid_pack_conv_weights_bf16(const float* w, const float* oscale, void* packed_bf16, int role, int o,
                                            int i, int kh, int kw, int kc, int bn, void* stream) {
    // same layout with bf16 elements (kc = twice the fp32 tile's chunk); used by refid_conv2d algo 2
    PackArgs p;
    REFID_CHECK(w && packed_bf16, "pack_bf16: null pointer");
    REFID_CHECK(role != REFID_ROLE_WINO_FWD && role != REFID_ROLE_WINO_DGRAD, "pack_bf16: no Winograd roles");
    REFID_CHECK(pack_geometry(role, o, i, kh, kw, kc, bn, &p) == 0, "pack_bf16: unknown role %d", role);
    p.w = w; p.dst = reinterpret_cast<float*>(packed_bf16); p.oscale = oscale; p.bf16 = 1;
    const long long total = (long long)p.ncls * p.nchunks * p.ntaps * p.rowsPad * p.KC;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights_bf16");
    return 0;
}

static int split_pack_mode(int role, int kh, int kw) {
    if ((role == REFID_ROLE_FWD || role == REFID_ROLE_DGRAD) && kh == 3 && kw == 3) return 0;
    if (role == REFID_ROLE_FWD && kh == 4 && kw == 4) return 1;
    if (role == REFID_ROLE_DOWN_DGRAD && kh == 4 && kw == 4) return 2;
    if ((role == REFID_ROLE_FWD || role == REFID_ROLE_DGRAD) && kh == 1 && kw == 1) return 3;
    return -1;
}

static long long split_pack_elems(const PackArgs& p, int planes, int mode) {
    if (mode == 3) return (long long)((p.K + 15) / 16) * planes * p.rowsPad * 16;
    return (long long)p.ncls * p.nchunks * (mode == 1 ? 4 : 1) * planes * (mode == 0 ? p.ntaps + 1 : 4) * p.rowsPad * 8;
}

extern "C" size_t refid_packed_weight_split_bytes(int role, int o, int i, int kh, int kw, int bn, int planes) {
    PackArgs p;
    const int mode = split_pack_mode(role, kh, kw);
    if (mode < 0 || planes < 1 || planes > 3) return 0;
    if (pack_geometry(role, o, i, kh, kw, 8, bn, &p)) return 0;
    return (size_t)split_pack_elems(p, planes, mode) * 2;
}

extern "C" int refid_pack_conv_weights_split(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                             int kh, int kw, int bn, int planes, void* stream) {
    PackArgs p;
    REFID_CHECK(w && packed, "pack_split: null pointer");
    const int mode = split_pack_mode(role, kh, kw);
    REFID_CHECK(mode >= 0, "pack_split: FWD / DGRAD of a 3x3 or 1x1 kernel, FWD / DOWN_DGRAD of a 4x4 (stride 2) kernel");
    REFID_CHECK(planes >= 1 && planes <= 3, "pack_split: 1, 2 or 3 planes (got %d)", planes);
    REFID_CHECK(pack_geometry(role, o, i, kh, kw, 8, bn, &p) == 0, "pack_split: unknown role %d", role);
    p.w = w; p.dst = reinterpret_cast<float*>(packed); p.oscale = oscale; p.bf16 = 1;
    const long long total = split_pack_elems(p, planes, mode);
    if (mode == 3) hipLaunchKernelGGL(pack_pw6_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p, planes);
    else hipLaunchKernelGGL(pack_split_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p, planes, mode);
    REFID_LAUNCH_CHECK("pack_conv_weights_split");
    return 0;
}

extern "C" size_t refid_packed_weight_split_f16_bytes(int role, int o, int i, int kh, int kw, int bn) {
    PackArgs p;
    const int mode = split_pack_mode(role, kh, kw);
    if (mode < 0 || mode == 3) return 0;
    if (pack_geometry(role, o, i, kh, kw, 8, bn, &p)) return 0;
    return 64 + (size_t)split_pack_elems(p, 2, mode) * 2;
}

extern "C" int refid_pack_conv_weights_split_f16(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                                 int kh, int kw, int bn, void* stream) {
    PackArgs p;
    REFID_CHECK(w && packed, "pack_split_f16: null pointer");
    REFID_CHECK((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "pack_split_f16: the packing must be 16-byte aligned");
    const int mode = split_pack_mode(role, kh, kw);
    REFID_CHECK(mode >= 0 && mode != 3, "pack_split_f16: FWD / DGRAD of a 3x3 kernel, FWD / DOWN_DGRAD of a 4x4 (stride 2) kernel");
    REFID_CHECK(pack_geometry(role, o, i, kh, kw, 8, bn, &p) == 0, "pack_split_f16: unknown role %d", role);
    p.w = w; p.dst = reinterpret_cast<float*>(packed); p.oscale = oscale; p.bf16 = 1;
    const long long total = split_pack_elems(p, 2, mode);
    hipLaunchKernelGGL(pack_absmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights_split_f16/absmax");
    hipLaunchKernelGGL(pack_split_f16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p, mode);
    REFID_LAUNCH_CHECK("pack_conv_weights_split_f16");
    return 0;
}

extern "C" size_t refid_packed_weight_wino6_bytes(int role, int o, int i, int bn) {
    PackArgs p;
    if (role != REFID_ROLE_WINO_FWD && role != REFID_ROLE_WINO_DGRAD) return 0;
    if (pack_geometry(role, o, i, 3, 3, 16, bn, &p)) return 0;
    return (size_t)p.nchunks * 16 * 3 * p.rowsPad * 16 * 2;
}

extern "C" int refid_pack_conv_weights_wino6(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                             int bn, void* stream) {
    PackArgs p;
    REFID_CHECK(w && packed, "pack_wino6: null pointer");
    REFID_CHECK(role == REFID_ROLE_WINO_FWD || role == REFID_ROLE_WINO_DGRAD, "pack_wino6: Winograd roles only");
    REFID_CHECK(pack_geometry(role, o, i, 3, 3, 16, bn, &p) == 0, "pack_wino6: bad geometry");
    p.w = w; p.dst = reinterpret_cast<float*>(packed); p.oscale = oscale; p.bf16 = 1;
    const long long total = (long long)p.nchunks * p.rowsPad * 16;                 // one thread per weight
    REFID_CHECK(total * 48 < 0x7fffffffLL, "pack_wino6: packing of %lld elements exceeds the 32-bit index range", total * 48);
    hipLaunchKernelGGL(pack_wino6_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights_wino6");
    return 0;
}

extern "C" size_t refid_packed_weight_wino3h_bytes(int role, int o, int i, int bn) {
    PackArgs p;
    if (role != REFID_ROLE_WINO_FWD && role != REFID_ROLE_WINO_DGRAD) return 0;
    if (pack_geometry(role, o, i, 3, 3, 16, bn, &p)) return 0;
    return (size_t)W3H_HEADER + (size_t)p.nchunks * 16 * 2 * p.rowsPad * 16 * 2;
}

extern "C" int refid_pack_conv_weights_wino3h(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                              int bn, void* stream) {
    PackArgs p;
    REFID_CHECK(w && packed, "pack_wino3h: null pointer");
    REFID_CHECK((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "pack_wino3h: the packing must be 16-byte aligned");
    REFID_CHECK(role == REFID_ROLE_WINO_FWD || role == REFID_ROLE_WINO_DGRAD, "pack_wino3h: Winograd roles only");
    REFID_CHECK(pack_geometry(role, o, i, 3, 3, 16, bn, &p) == 0, "pack_wino3h: bad geometry");
    p.w = w; p.dst = reinterpret_cast<float*>(packed); p.oscale = oscale; p.bf16 = 1;
    const long long total = (long long)p.nchunks * p.rowsPad * 16;                 // one thread per weight
    REFID_CHECK(total * 32 < 0x7fffffffLL, "pack_wino3h: packing of %lld elements exceeds the 32-bit index range", total * 32);
    hipLaunchKernelGGL(pack_absmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights_wino3h/absmax");
    hipLaunchKernelGGL(pack_wino3h_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights_wino3h");
    return 0;
}

// ---- batched packing ------------------------------------------------------------------------------------------------
extern "C" size_t refid_pack_entry_bytes(void) { return sizeof(PackEntry); }

// error paths of the table builder return -1 (a positive value is a workgroup count), never REFID_CHECK's 1
#define REFID_FILL_CHECK(cond, ...)               \
    do {                                          \
        if (!(cond)) {                            \
            refid_set_error(__VA_ARGS__);         \
            return -1;                            \
        }                                         \
    } while (0)

extern "C" int refid_pack_entry_fill(void* entry_host, int kind, const float* w, const float* oscale, void* dst, int role, int o,
                                     int i, int kh, int kw, int kc, int bn, int planes, int blk0) {
    REFID_FILL_CHECK(entry_host && w && dst, "pack_entry_fill: null pointer");
    REFID_FILL_CHECK(blk0 >= 0, "pack_entry_fill: negative first block %d", blk0);
    PackEntry en;
    memset(&en, 0, sizeof(en));
    en.kind = kind; en.planes = planes; en.mode = 0; en.blk0 = blk0;
    PackArgs& p = en.p;
    if (kind == 4) {                                        // out[e] = w[e] * oscale[e], e < o
        REFID_FILL_CHECK(oscale != nullptr && o > 0, "pack_entry_fill: the vector product needs both factors");
        p.w = w; p.oscale = oscale; p.dst = reinterpret_cast<float*>(dst);
        en.total = o;
    } else if (kind == 0) {
        // the same conditions as refid_pack_conv_weights / _scaled / _bf16 (pack_impl)
        REFID_FILL_CHECK(planes == 0 || planes == 1, "pack_entry_fill: kind 0 takes planes = 0 (fp32) or 1 (bf16), got %d", planes);
        REFID_FILL_CHECK(pack_geometry(role, o, i, kh, kw, kc, bn, &p) == 0, "pack_entry_fill: unknown role %d", role);
        REFID_FILL_CHECK((role != REFID_ROLE_CONVT && role != REFID_ROLE_CONVT_DGRAD && role != REFID_ROLE_CONVT_DGRAD_PW) || (kh == 2 && kw == 2),
                         "pack_entry_fill: convT roles need a 2x2 kernel");
        REFID_FILL_CHECK(role != REFID_ROLE_DOWN_DGRAD || (kh == 4 && kw == 4), "pack_entry_fill: down-dgrad needs a 4x4 kernel");
        const bool wino = role == REFID_ROLE_WINO_FWD || role == REFID_ROLE_WINO_DGRAD;
        REFID_FILL_CHECK(!wino || (kh == 3 && kw == 3 && kc == 8 && planes == 0),
                         "pack_entry_fill: Winograd roles need a 3x3 kernel, kc = 8 and fp32 output");
        REFID_FILL_CHECK(oscale == nullptr || role == REFID_ROLE_FWD || role == REFID_ROLE_DGRAD || wino,
                         "pack_entry_fill: only FWD/DGRAD roles take a scale");
        p.bf16 = planes;                                    // (kind 0: `planes` = 1 selects bf16 output)
        en.total = (long long)p.ncls * p.nchunks * p.ntaps * p.rowsPad * p.KC;
    } else if (kind == 1 || kind == 2) {
        const int mode = split_pack_mode(role, kh, kw);
        REFID_FILL_CHECK(mode >= 0 && (kind == 2) == (mode == 3) && planes >= 1 && planes <= 3, "pack_entry_fill: bad split geometry");
        REFID_FILL_CHECK(pack_geometry(role, o, i, kh, kw, 8, bn, &p) == 0, "pack_entry_fill: unknown role %d", role);
        p.bf16 = 1;
        en.mode = mode;
        en.total = split_pack_elems(p, planes, mode);
    } else if (kind == 3) {
        REFID_FILL_CHECK((role == REFID_ROLE_WINO_FWD || role == REFID_ROLE_WINO_DGRAD) && kh == 3 && kw == 3, "pack_entry_fill: Winograd roles, 3x3");
        REFID_FILL_CHECK(pack_geometry(role, o, i, 3, 3, 16, bn, &p) == 0, "pack_entry_fill: unknown role %d", role);
        p.bf16 = 1;
        en.total = (long long)p.nchunks * p.rowsPad * 16;    // WEIGHTS (one thread each writes its 48 plane entries)
        REFID_FILL_CHECK(en.total * 48 < 0x7fffffffLL, "pack_entry_fill: Winograd x six packing exceeds the 32-bit index range");
    } else if (kind == 6) {
        const int mode = split_pack_mode(role, kh, kw);
        REFID_FILL_CHECK(mode >= 0 && mode != 3, "pack_entry_fill: bad split geometry for the fp16 planes");
        REFID_FILL_CHECK((reinterpret_cast<uintptr_t>(dst) & 15) == 0, "pack_entry_fill: the fp16 split packing must be 16-byte aligned");
        REFID_FILL_CHECK(pack_geometry(role, o, i, kh, kw, 8, bn, &p) == 0, "pack_entry_fill: unknown role %d", role);
        p.bf16 = 1;
        en.mode = mode;
        en.total = split_pack_elems(p, 2, mode);
    } else if (kind == 5) {
        REFID_FILL_CHECK((role == REFID_ROLE_WINO_FWD || role == REFID_ROLE_WINO_DGRAD) && kh == 3 && kw == 3, "pack_entry_fill: Winograd roles, 3x3");
        REFID_FILL_CHECK((reinterpret_cast<uintptr_t>(dst) & 15) == 0, "pack_entry_fill: the fp16 Winograd packing must be 16-byte aligned");
        REFID_FILL_CHECK(pack_geometry(role, o, i, 3, 3, 16, bn, &p) == 0, "pack_entry_fill: unknown role %d", role);
        p.bf16 = 1;
        en.total = (long long)p.nchunks * p.rowsPad * 16;    // WEIGHTS (one thread each writes its 32 plane entries)
        REFID_FILL_CHECK(en.total * 32 < 0x7fffffffLL, "pack_entry_fill: Winograd x three packing exceeds the 32-bit index range");
    } else {
        REFID_FILL_CHECK(false, "pack_entry_fill: unknown kind %d", kind);
    }
    REFID_FILL_CHECK(en.total > 0, "pack_entry_fill: empty packing (o=%d, i=%d)", o, i);
    if (kind != 4) { p.w = w; p.dst = reinterpret_cast<float*>(dst); p.oscale = oscale; }
    long long nb = (en.total + 255) / 256;
    en.nblk = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    memcpy(entry_host, &en, sizeof(en));
    return en.nblk;
}

// Host-side check of a finished table (before it is copied to the device): kinds valid, first blocks strictly increasing from
// 0 with blk0[k+1] = blk0[k] + nblk[k] (the kernel's binary search relies on it).  Returns the total workgroup count, -1 on error.
extern "C" int refid_pack_table_check(const void* table_host, int n) {
    REFID_FILL_CHECK(table_host != nullptr && n > 0, "pack_table_check: empty table");
    const PackEntry* t = reinterpret_cast<const PackEntry*>(table_host);
    long long blk = 0;
    for (int k = 0; k < n; ++k) {
        REFID_FILL_CHECK(t[k].kind >= 0 && t[k].kind <= 6 && t[k].total > 0 && t[k].nblk >= 1,
                         "pack_table_check: record %d was never filled (kind %d, total %lld)", k, t[k].kind, t[k].total);
        REFID_FILL_CHECK(t[k].blk0 == blk, "pack_table_check: record %d starts at block %d, expected %lld", k, t[k].blk0, blk);
        blk += t[k].nblk;
    }
    REFID_FILL_CHECK(blk < 0x7fffffffLL, "pack_table_check: too many workgroups");
    return (int)blk;
}

// The scale exponents of the table's kind-5 records (Winograd x three fp16 products: max |w| per tensor): one launch, BEFORE
// refid_pack_batch on the same stream, whenever the table holds such records (a no-op for the others).
extern "C" int refid_pack_batch_prepass(const void* table_dev, int n, void* stream) {
    REFID_CHECK(table_dev != nullptr && n > 0, "pack_batch_prepass: empty table");
    hipLaunchKernelGGL(pack_prepass_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackEntry*>(table_dev), n);
    REFID_LAUNCH_CHECK("pack_batch_prepass");
    return 0;
}

extern "C" int refid_pack_batch(const void* table_dev, int n, int nblocks, void* stream) {
    REFID_CHECK(table_dev != nullptr && n > 0 && nblocks > 0, "pack_batch: empty table");
    hipLaunchKernelGGL(pack_batch_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackEntry*>(table_dev), n);
    REFID_LAUNCH_CHECK("pack_batch");
    return 0;
}

extern "C" int refid_pack_conv_weights_scaled(const float* w, const float* oscale, float* packed, int role, int o,
                                              int i, int kh, int kw, int kc, int bn, void* stream) {
    REFID_CHECK(role == REFID_ROLE_FWD || role == REFID_ROLE_DGRAD || role == REFID_ROLE_WINO_FWD ||
                    role == REFID_ROLE_WINO_DGRAD, "pack_scaled: only FWD/DGRAD roles take a scale");
    return pack_impl(w, oscale, packed, role, o, i, kh, kw, kc, bn, stream);
}

static int pack_impl(const float* w, const float* oscale, float* packed, int role, int o, int i, int kh,
                     int kw, int kc, int bn, void* stream) {
    PackArgs p;
    p.oscale = oscale;
    p.bf16 = 0;
    REFID_CHECK(w && packed, "pack: null pointer");
    REFID_CHECK(pack_geometry(role, o, i, kh, kw, kc, bn, &p) == 0, "pack: unknown role %d", role);
    REFID_CHECK((role != REFID_ROLE_CONVT && role != REFID_ROLE_CONVT_DGRAD && role != REFID_ROLE_CONVT_DGRAD_PW) || (kh == 2 && kw == 2),
                "pack: convT roles need a 2x2 kernel");
    REFID_CHECK(role != REFID_ROLE_DOWN_DGRAD || (kh == 4 && kw == 4), "pack: down-dgrad needs a 4x4 kernel");
    REFID_CHECK((role != REFID_ROLE_WINO_FWD && role != REFID_ROLE_WINO_DGRAD) || (kh == 3 && kw == 3 && kc == 8),
                "pack: Winograd roles need a 3x3 kernel and kc = 8");
    p.w = w; p.dst = packed;
    const long long total = (long long)p.ncls * p.nchunks * p.ntaps * p.rowsPad * p.KC;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    REFID_LAUNCH_CHECK("pack_conv_weights");
    return 0;
}

// one launcher for the four NCHW -> NHWC entry points: thin tensors (c_pad == 4, 16-byte aligned dst) take the pixel-per-thread form
static int launch_to_nhwc(const float* src, long long b_stride, float* dst, int n, int c, int hw, int c_pad, int t_count,
                          long long t_stride, int nb, hipStream_t st, const char* what) {
    // a tCount sum in the time-major form would need two time strides: the entry points never combine them
    if (c_pad == 4 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(cdiv(hw, 256), n), dim3(256), 0, st, src, b_stride,
                           reinterpret_cast<f32x4*>(dst), c, hw, t_count, t_stride, nb);
    } else {
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(hw, 64), n), dim3(256), 0, st, src, b_stride, dst, c, hw, c_pad, t_count,
                           t_stride, nb);
    }
    REFID_LAUNCH_CHECK(what);
    return 0;
}

extern "C" int refid_nchw_to_nhwc(const float* src, long long src_batch_stride, float* dst, int n, int c, int h,
                                  int w, int c_pad, void* stream) {
    REFID_CHECK(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c, "nchw_to_nhwc: bad arguments");
    return launch_to_nhwc(src, src_batch_stride, dst, n, c, h * w, c_pad, 1, 0ll, 0, (hipStream_t)stream, "nchw_to_nhwc");
}

extern "C" int refid_nchw_to_nhwc_tb(const float* src, long long b_stride, long long t_stride, float* dst, int nb, int nt,
                                     int c, int h, int w, int c_pad, void* stream) {
    REFID_CHECK(src && dst && nb > 0 && nt > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c, "nchw_to_nhwc_tb: bad arguments");
    REFID_CHECK((long long)nb * nt <= 65535, "nchw_to_nhwc_tb: more than 65535 (sample, step) blocks");
    return launch_to_nhwc(src, b_stride, dst, nb * nt, c, h * w, c_pad, 1, t_stride, nb, (hipStream_t)stream, "nchw_to_nhwc_tb");
}

extern "C" int refid_nchw_tsum_to_nhwc(const float* src, long long src_batch_stride, long long t_stride, int t_count,
                                       float* dst, int n, int c, int h, int w, int c_pad, void* stream) {
    REFID_CHECK(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c && t_count > 0,
                "nchw_tsum_to_nhwc: bad arguments");
    return launch_to_nhwc(src, src_batch_stride, dst, n, c, h * w, c_pad, t_count, t_stride, 0, (hipStream_t)stream,
                          "nchw_tsum_to_nhwc");
}

static int launch_to_nchw(const float* src, int ld, float* dst, long long b_stride, int n, int c, int hw, int nb,
                          long long t_stride, hipStream_t st, const char* what) {
    if (ld == 4 && c <= 4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(cdiv(hw, 256), n), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src), dst,
                           b_stride, c, hw, nb, t_stride);
    } else {
        hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(hw, 64), n), dim3(256), 0, st, src, ld, dst, b_stride, c, hw, nb, t_stride);
    }
    REFID_LAUNCH_CHECK(what);
    return 0;
}

extern "C" int refid_nhwc_to_nchw(const float* src, int ld, float* dst, long long dst_batch_stride, int n,
                                  int c, int h, int w, void* stream) {
    REFID_CHECK(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && ld >= c, "nhwc_to_nchw: bad arguments");
    return launch_to_nchw(src, ld, dst, dst_batch_stride, n, c, h * w, 0, 0ll, (hipStream_t)stream, "nhwc_to_nchw");
}

extern "C" int refid_nhwc_to_nchw_tb(const float* src, int ld, float* dst, long long b_stride, long long t_stride, int nb,
                                     int nt, int c, int h, int w, void* stream) {
    REFID_CHECK(src && dst && nb > 0 && nt > 0 && c > 0 && h > 0 && w > 0 && ld >= c, "nhwc_to_nchw_tb: bad arguments");
    REFID_CHECK((long long)nb * nt <= 65535, "nhwc_to_nchw_tb: more than 65535 (sample, step) blocks");
    return launch_to_nchw(src, ld, dst, b_stride, nb * nt, c, h * w, nb, t_stride, (hipStream_t)stream, "nhwc_to_nchw_tb");
}

extern "C" int refid_add(const float* a, const float* b, float* out, long long count, void* stream) {
    REFID_CHECK(a && b && out && count > 0 && count % 4 == 0, "add: bad arguments (count=%lld)", count);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)a, (const f32x4*)b, (f32x4*)out, count / 4);
    REFID_LAUNCH_CHECK("add");
    return 0;
}

extern "C" int refid_sum_n(const float* const* in, int n, float* out, long long count, void* stream) {
    REFID_CHECK(in && out && n >= 1 && n <= REFID_SUM_MAX && count > 0 && count % 4 == 0,
                "sum_n: 1..%d inputs, count a positive multiple of 4 (n=%d, count=%lld)", REFID_SUM_MAX, n, count);
    SumNArgs a;
    for (int k = 0; k < REFID_SUM_MAX; ++k) {
        a.in[k] = reinterpret_cast<const f32x4*>(k < n ? in[k] : in[0]);
        REFID_CHECK(a.in[k] != nullptr && (reinterpret_cast<uintptr_t>(a.in[k]) & 15) == 0, "sum_n: input %d is null or not 16-byte aligned", k);
    }
    a.n = n;
    hipLaunchKernelGGL(sum_n_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, a, (f32x4*)out, count / 4);
    REFID_LAUNCH_CHECK("sum_n");
    return 0;
}

extern "C" int refid_act_bwd(const float* g, const float* y, float* out, float slope, int accumulate,
                             long long count, void* stream) {
    REFID_CHECK(g && y && out && count > 0 && count % 4 == 0, "act_bwd: bad arguments (count=%lld)", count);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)g, (const f32x4*)y, (f32x4*)out, slope, accumulate, count / 4);
    REFID_LAUNCH_CHECK("act_bwd");
    return 0;
}
