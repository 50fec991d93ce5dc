// Weight-gradient (and bias-gradient) of the REFID convolutions on the fp32 matrix cores.
//
//   dW[o][i][ky][kx] += sum_{n,y,x} g[n,y,x,o] * src[n, y*S+ky-pad, x*S+kx-pad, i]
//   db[o]            += sum_{n,y,x} g[n,y,x,o]
//
// GEMM view: M = output channels (o), N = input channels (i), K = output pixels; one
// accumulator tile per tap.  v_mfma_f32_32x32x2_f32 consumes two adjacent pixels per
// instruction: lane (r = l&31, kh = l>>5) supplies g[pixel 2q+kh][o = r] and
// src[pixel' (2q+kh)*S + tap][i = r] -- both are plain conflict-free ds_read_b32 from the
// NHWC LDS images (32 consecutive channels per half-wave).
//
// A workgroup owns a (COT x CIT) channel tile for ALL taps and walks a strided subset of
// the pixel tiles (split-K over pixels); per pixel tile the gradient tile and the input
// HALO tile are staged once in LDS and re-used by every tap.  Partial sums go to private
// slabs [split][tap][o][i] with plain coalesced stores (no atomics, deterministic); a
// second kernel reduces the slabs and ACCUMULATES into the parameter-layout gradient,
// because weights are shared over the T recurrent steps.
#include "common.h"
#include <vector>
#include <cstring>
#include "wgrad_args.h"
#include <cstdlib>

namespace {

const bool USE_THIN_WGRAD = !(getenv("REFID_THIN_WGRAD") && getenv("REFID_THIN_WGRAD")[0] == '0');


template <int KH_, int KW_, int S_, int TPW_, int SM_, int SN_, int WR_, int WC_, int WT_, int TH_, int TW_>
struct WCfg {
    static constexpr int KH = KH_, KW = KW_, S = S_, TPW = TPW_, SM = SM_, SN = SN_;
    static constexpr int WR = WR_, WC = WC_, WT = WT_, TH = TH_, TW = TW_;
    static constexpr int NTAPS = KH * KW;
    static constexpr int COT = WR * SM * 32, CIT = WC * SN * 32;
    static constexpr int PX = TH * TW;
    static constexpr int HH = (TH - 1) * S + KH, HWD = (TW - 1) * S + KW, HP = HH * HWD;
    static constexpr int G4 = COT / 4, X4 = CIT / 4;
    static constexpr int G_TOTAL = PX * G4, X_TOTAL = HP * X4;
    static constexpr int G_ITEMS = (G_TOTAL + 255) / 256, X_ITEMS = (X_TOTAL + 255) / 256;
    static constexpr int LDS_BYTES = (G_TOTAL + X_TOTAL) * 16 + COT * 4;
    static_assert(WR * WC * WT == 4, "4 waves");
    static_assert(WT * TPW >= NTAPS, "tap groups must cover the kernel");
    static_assert(TW % 2 == 0 && 256 % G4 == 0 && 256 % X4 == 0, "static thread->channel mapping");
};

template <class C>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sG4 = reinterpret_cast<f32x4*>(smem);
    f32x4* sX4 = sG4 + C::G_TOTAL;
    float* sBias = reinterpret_cast<float*>(sX4 + C::X_TOTAL);
    const float* sG = reinterpret_cast<const float*>(sG4);
    const float* sX = reinterpret_cast<const float*>(sX4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int tg = wave % C::WT;                       // tap group
    const int wc = (wave / C::WT) % C::WC;
    const int wr = wave / (C::WT * C::WC);
    const int co0 = blockIdx.z * C::COT, ci0 = blockIdx.y * C::CIT;
    const int split = blockIdx.x;

    // static thread -> channel mapping of the loaders
    const int gq = tid % C::G4, xq = tid % C::X4;
    const int gco = co0 + gq * 4;
    const bool gcok = gco < a.Co;
    const int xc = ci0 + xq * 4;
    const bool xFromA = xc < a.Ca;
    const bool xcok = xc < a.Ctot;
    const int xld = xFromA ? a.ldA : a.ldB;
    const int ntAll = a.ntiles * a.groups;             // tiles of all grouped time steps
    const int xcc = xFromA ? xc : xc - a.Ca;

    int tdy[C::TPW], tdx[C::TPW];
    bool tok[C::TPW];
#pragma unroll
    for (int tt = 0; tt < C::TPW; ++tt) {
        const int tap = tg * C::TPW + tt;
        tok[tt] = tap < C::NTAPS;
        tdy[tt] = tap / C::KW; tdx[tt] = tap % C::KW;
    }

    f32x16 acc[C::TPW][C::SM][C::SN];
#pragma unroll
    for (int tt = 0; tt < C::TPW; ++tt)
#pragma unroll
        for (int sm = 0; sm < C::SM; ++sm)
#pragma unroll
            for (int sn = 0; sn < C::SN; ++sn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][sm][sn][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    f32x4 rg[C::G_ITEMS], rx[C::X_ITEMS];

    auto load_tile = [&](int pt) {
        const int grp = pt / a.ntiles;                  // workgroup-uniform: the time step this tile belongs to
        const float* gsrc = a.g[grp];
        const float* xsrc = xFromA ? a.inA[grp] : a.inB[grp];
        int t = pt - grp * a.ntiles;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        const int n = t / a.tilesY;
        const int oy0 = ty * C::TH, ox0 = tx * C::TW;
        const int iy0 = oy0 * C::S - a.pad, ix0 = ox0 * C::S - a.pad;
#pragma unroll
        for (int it = 0; it < C::G_ITEMS; ++it) {
            const int p = tid / C::G4 + it * (256 / C::G4);
            const int oy = oy0 + p / C::TW, ox = ox0 + p % C::TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gcok && p < C::PX && oy < a.Ho && ox < a.Wo)
                v = *reinterpret_cast<const f32x4*>(gsrc + ((long long)(n * a.Ho + oy) * a.Wo + ox) * a.ldG + gco);
            rg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const int hp = tid / C::X4 + it * (256 / C::X4);
            const int iy = iy0 + hp / C::HWD, ix = ix0 + hp % C::HWD;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xcok && hp < C::HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const f32x4*>(xsrc + ((long long)(n * a.H + iy) * a.W + ix) * xld + xcc);
            rx[it] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < C::G_ITEMS; ++it) {
            const int p = tid / C::G4 + it * (256 / C::G4);
            if (p < C::PX) sG4[p * C::G4 + gq] = rg[it];
            bsum += rg[it];          // bias partial: summed HERE, not at load time -- using a prefetched register right
                                     // after its load was issued forced a vmcnt(0) before the MFMA section
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const int hp = tid / C::X4 + it * (256 / C::X4);
            if (hp < C::HP) sX4[hp * C::X4 + xq] = rx[it];
        }
    };

    int pt = split;
    if (pt < ntAll) {
        load_tile(pt);
        store_tile();
    }
    __syncthreads();

    const int aoff = wr * C::SM * 32 + li;
    const int boff = wc * C::SN * 32 + li;

    for (; pt < ntAll; pt += a.nsplit) {
        const bool more = pt + a.nsplit < ntAll;
        if (more) load_tile(pt + a.nsplit);

        // All LDS operands of a K step (pixel pair) are read up front, then the MFMAs consume them as they arrive
        // (the naive "read, wait, 2 MFMAs" order exposed the LDS latency every two MFMAs).
        for (int row = 0; row < C::TH; ++row) {
#pragma unroll 2
            for (int qk = 0; qk < C::TW / 2; ++qk) {
                const int pcol = 2 * qk + kh;
                float av[C::SM], bv[C::TPW][C::SN];
#pragma unroll
                for (int sm = 0; sm < C::SM; ++sm) av[sm] = sG[(row * C::TW + pcol) * C::COT + aoff + sm * 32];
#pragma unroll
                for (int tt = 0; tt < C::TPW; ++tt) {
                    const int hp = (row * C::S + tdy[tt]) * C::HWD + pcol * C::S + tdx[tt];
#pragma unroll
                    for (int sn = 0; sn < C::SN; ++sn)
                        bv[tt][sn] = (C::WT > 1 && !tok[tt]) ? 0.f : sX[hp * C::CIT + boff + sn * 32];
                }
#pragma unroll
                for (int tt = 0; tt < C::TPW; ++tt) {
                    if (C::WT > 1 && !tok[tt]) continue;
#pragma unroll
                    for (int sm = 0; sm < C::SM; ++sm)
#pragma unroll
                        for (int sn = 0; sn < C::SN; ++sn)
                            acc[tt][sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tt][sn], av[sm],
                                                                                  acc[tt][sm][sn], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- write the partial slab: [split][tap][co][ci], 128-byte runs along ci -------------
#pragma unroll
    for (int tt = 0; tt < C::TPW; ++tt) {
        const int tap = tg * C::TPW + tt;
        if (tap >= C::NTAPS) continue;
        float* sl = a.slabs + ((long long)(split * C::NTAPS + tap) * a.CoP) * a.CiP;
#pragma unroll
        for (int sm = 0; sm < C::SM; ++sm)
#pragma unroll
            for (int sn = 0; sn < C::SN; ++sn) {
                // D[ci][co]: lane li = output channel, register quad q = 4 consecutive input channels
                const int co = co0 + wr * C::SM * 32 + sm * 32 + li;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = ci0 + wc * C::SN * 32 + sn * 32 + 8 * q + 4 * kh;
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[tt][sm][sn][4 * q + k];
                    f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
                    if (a.accum) v += *dst;
                    *dst = v;
                }
            }
    }
    // ---- bias partial: only the first input-channel tile column owns it -------------------
    if (a.bslabs != nullptr && blockIdx.y == 0) {
        // fixed order (no LDS atomics): xor-shuffle tree over the wave's threads that share gq, then the four waves
        // in sequence; the operand tiles at the start of the LDS are dead here (every wave is past its last read)
        float* sred = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = refid_wave_rows_sum<C::G4>(bsum[k]);
            if ((tid & 63) < C::G4) sred[(tid >> 6) * C::COT + gq * 4 + k] = v;
        }
        __syncthreads();
        if (tid < C::COT) {
            const float tot = ((sred[tid] + sred[C::COT + tid]) + sred[2 * C::COT + tid]) + sred[3 * C::COT + tid];
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + tid;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 1x1 (pointwise) weight gradient, register-operand tile.
//   dW[co][ci] = sum_p g[p][co] * x[p][ci]   -- a plain GEMM whose K index is the PIXEL, at 20-40 FLOP/B.
// The LDS-tiled kernel above spends its time in barriers (a 64-pixel tile is 32 K steps); here a wave
// streams its pixel range straight from global memory into MFMA operands: one 4-byte buffer load per lane
// per operand row (a half wave reads 128 contiguous bytes of one pixel), PD pixel pairs in flight, no LDS,
// no barrier.  Workgroup = WR x WC waves, wave tile = 32 output channels x 32*SN input channels.
// Same slab layout / reduce kernel / phases as wgrad_kernel.
struct WpArgs {
    const float* g; int ldG, Co;
    const float* inA; const float* inB; int ldA, ldB, Ca, Ctot;
    float* slabs; float* bslabs;
    long long npix; int chunk;          // pixels per split (even)
    int CoP, CiP, WR, accum;
};
typedef unsigned int u32;

template <int SN>
__global__ __launch_bounds__(256) void wgrad_pw_kernel(const WpArgs a) {
    constexpr int PD = 8;                               // pixel pairs per prefetch block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave % a.WR, wc = wave / a.WR;
    const int WC = 4 / a.WR;
    const int co0 = (blockIdx.z * a.WR + wr) * 32;
    const int ci0 = (blockIdx.y * WC + wc) * (32 * SN);
    const int split = blockIdx.x;
    const long long k0 = (long long)split * a.chunk;

    const int lim = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.g), 0, (int)min(a.npix * a.ldG * 4, (long long)lim), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.inA), 0, (int)min(a.npix * a.ldA * 4, (long long)lim), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.inB ? a.inB : a.inA), 0, (int)min(a.npix * (a.inB ? a.ldB : a.ldA) * 4, (long long)lim),
        0x00020000);
    // per-lane byte offsets (pixel kh of a pair, channel li of the row); -1 = out of range -> reads 0
    const int voG = (co0 + li < a.Co) ? (kh * a.ldG + co0 + li) * 4 : -1;
    int voX[SN]; bool fromA[SN];
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
        const int c = ci0 + sn * 32 + li;
        fromA[sn] = (ci0 + sn * 32) < a.Ca;             // wave-uniform: Ca % 32 == 0 for two sources
        const int cc = fromA[sn] ? c : c - a.Ca;
        const int ld = fromA[sn] ? a.ldA : a.ldB;
        voX[sn] = (c < a.Ctot) ? (kh * ld + cc) * 4 : -1;
    }
    const int npairs = a.chunk / 2;

    f32x16 acc[SN];
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[sn][r] = 0.f;
    float bsum = 0.f;

    float gv[2][PD], xv[2][PD][SN];
    auto load_block = [&](int blk, float (&gd)[PD], float (&xd)[PD][SN]) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const long long p = k0 + 2ll * (blk * PD + j);
            const bool ok = blk * PD + j < npairs;
            gd[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, ok ? voG : -1, (int)(p * a.ldG * 4), 0));
#pragma unroll
            for (int sn = 0; sn < SN; ++sn) {
                const u32 v = fromA[sn] ? __builtin_amdgcn_raw_buffer_load_b32(rsA, ok ? voX[sn] : -1, (int)(p * a.ldA * 4), 0)
                                        : __builtin_amdgcn_raw_buffer_load_b32(rsB, ok ? voX[sn] : -1, (int)(p * a.ldB * 4), 0);
                xd[j][sn] = __builtin_bit_cast(float, v);
            }
        }
    };
    auto mma_block = [&](const float (&gd)[PD], const float (&xd)[PD][SN]) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            bsum += gd[j];
#pragma unroll
            for (int sn = 0; sn < SN; ++sn)
                acc[sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(xd[j][sn], gd[j], acc[sn], 0, 0, 0);
        }
    };
    const int nblk = (npairs + PD - 1) / PD;
    load_block(0, gv[0], xv[0]);
    for (int b = 0; b < nblk; b += 2) {
        if (b + 1 < nblk) load_block(b + 1, gv[1], xv[1]);
        mma_block(gv[0], xv[0]);
        if (b + 1 >= nblk) break;
        if (b + 2 < nblk) load_block(b + 2, gv[0], xv[0]);
        mma_block(gv[1], xv[1]);
    }

    // ---- partial slab [split][co][ci] (one tap): D[ci][co], lane li = co, register quad q = 4 consecutive ci
    float* sl = a.slabs + (long long)split * a.CoP * a.CiP;
    const int co = co0 + li;
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = ci0 + sn * 32 + 8 * q + 4 * kh;
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[sn][4 * q + k];
            f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
            if (a.accum) v += *dst;
            *dst = v;
        }
    if (a.bslabs != nullptr && blockIdx.y == 0 && wc == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (kh == 0) {
            float* dst = a.bslabs + (long long)split * a.CoP + co;
            *dst = a.accum ? *dst + bsum : bsum;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Thin-input KxK weight gradient (the event head: conv5x5 of 2 -> 32 channels over all B*T frames at once,
// XXNet_final_attenfusion_arch.py:98-99,149).  With 4 (padded) input channels a 32-column MFMA tile of input
// channels is 8x waste; here the 32 columns are (8 taps x 4 channels), so a pixel pair costs ceil(K*K/8) MFMAs
// instead of K*K.  The output gradient streams straight from global memory into MFMA operands (one dword per lane
// per pixel pair, coalesced 128 B per pixel); only the small input halo goes through LDS (double buffered).
// Wave w owns row w of a 4x32-pixel tile; a workgroup walks over many tiles and writes one slab per wave.
struct WtArgs {
    const float* g; int ldG, Co;
    const float* x; int ldX;                 // 4 channels per pixel (ldX >= 4)
    float* slabs; float* bslabs;
    int N, H, W, pad, K;                      // stride 1, Ho = H, Wo = W
    int tilesX, tilesY, ntiles, nsplit, accum;
};
constexpr int WT_TH = 4, WT_TW = 32, WT_MAXG = 4;           // up to 32 taps (5x5 = 25)

__global__ __launch_bounds__(256) void wgrad_thin_kernel(const WtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.K, HWD = WT_TW + K - 1, HP = (WT_TH + K - 1) * HWD;
    f32x4* sX4 = reinterpret_cast<f32x4*>(smem);            // 2 x HP pixels x 4 channels
    const float* sX = reinterpret_cast<const float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ntaps = K * K, ngrp = (ntaps + 7) / 8;
    // B operand column li = (tap t = li >> 2 of the group, channel li & 3): per-lane LDS offset per group
    int boff[WT_MAXG];
    float bmask[WT_MAXG];                               // columns beyond the last tap contribute zeros (branch-free)
#pragma unroll
    for (int gidx = 0; gidx < WT_MAXG; ++gidx) {
        const int tap = gidx * 8 + (li >> 2);
        const bool ok = tap < ntaps;
        boff[gidx] = ok ? ((tap / K) * HWD + (tap % K)) * 4 + (li & 3) : 0;
        bmask[gidx] = ok ? 1.f : 0.f;
    }
    const long long gmax = (long long)a.N * a.H * a.W * a.ldG * 4;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.g), 0, (int)min(gmax, 0x7fffffffLL), 0x00020000);
    const bool cok = li < a.Co;

    f32x16 acc[WT_MAXG];
#pragma unroll
    for (int gidx = 0; gidx < WT_MAXG; ++gidx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[gidx][r] = 0.f;
    float bsum = 0.f;

    auto tile_origin = [&](int pt, int& n, int& oy0, int& ox0) {
        int t = pt;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        n = t / a.tilesY; oy0 = ty * WT_TH; ox0 = tx * WT_TW;
    };
    f32x4 rx[2];
    auto load_x = [&](int pt) {
        int n, oy0, ox0; tile_origin(pt, n, oy0, ox0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hp = tid + it * 256;
            const int iy = oy0 - a.pad + hp / HWD, ix = ox0 - a.pad + hp % HWD;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const f32x4*>(a.x + ((long long)(n * a.H + iy) * a.W + ix) * a.ldX);
            rx[it] = v;
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hp = tid + it * 256;
            if (hp < HP) sX4[buf * HP + hp] = rx[it];
        }
    };
    // this wave's row of the tile: 16 pixel pairs of the output gradient, prefetched one tile ahead into the OTHER
    // register set (two explicit sets and a loop unrolled by two: indexing a register array with a runtime buffer
    // index makes the compiler select element by element and wait after every single load)
    float gvA[16], gvB[16];
    // per-lane byte offset inside a tile row (pixel kh of a pair, channel li); the pair index and the tile origin go
    // into the SCALAR offset of the buffer load: no per-load vector address math, no divergent branches (those made
    // the compiler recycle in-flight destination registers and wait after every load)
    const int voG = cok ? (kh * a.ldG + li) * 4 : -1;
    auto load_g = [&](int pt, float (&dst)[16]) {
        int n, oy0, ox0; tile_origin(pt, n, oy0, ox0);
        const int oy = oy0 + wave;                                  // wave-uniform
        const int rowbase = (oy < a.H) ? (int)((((long long)(n * a.H + oy) * a.W + ox0) * a.ldG) * 4) : 0;
        const int rowok = (oy < a.H) ? 0 : -1;                      // all-ones -> out-of-range offset -> zeros
        const int wleft = a.W - ox0;                                 // valid columns in this tile row
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int colbad = (2 * j + kh < wleft) ? 0 : -1;
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, voG | rowok | colbad,
                                                                                     rowbase + 2 * j * a.ldG * 4, 0));
        }
    };
    auto tile = [&](int pt, int buf, const float (&cur)[16], float (&nxt)[16]) {
        const bool more = pt + a.nsplit < a.ntiles;
        if (more) { load_x(pt + a.nsplit); load_g(pt + a.nsplit, nxt); }
        const float* xb = sX + (buf * HP + wave * HWD + kh) * 4;          // + (2j) * 4 + boff
        float bva[WT_MAXG], bvb[WT_MAXG];                 // LDS operands of pair j+1 are fetched before the MFMAs of pair j
#pragma unroll
        for (int gidx = 0; gidx < WT_MAXG; ++gidx) bva[gidx] = xb[boff[gidx]];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float gq = cur[j];
            bsum += gq;
            float (&bc)[WT_MAXG] = (j & 1) ? bvb : bva;
            float (&bn)[WT_MAXG] = (j & 1) ? bva : bvb;
            if (j + 1 < 16) {
#pragma unroll
                for (int gidx = 0; gidx < WT_MAXG; ++gidx) bn[gidx] = xb[2 * (j + 1) * 4 + boff[gidx]];
            }
#pragma unroll
            for (int gidx = 0; gidx < WT_MAXG; ++gidx) {
                if (gidx >= ngrp) break;
                acc[gidx] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[gidx] * bmask[gidx], gq, acc[gidx], 0, 0, 0);
            }
        }
        if (more) store_x(buf ^ 1);
        __syncthreads();
    };

    int pt = blockIdx.x;
    if (pt < a.ntiles) { load_x(pt); store_x(0); load_g(pt, gvA); }
    __syncthreads();
    for (; pt < a.ntiles; pt += 2 * a.nsplit) {
        tile(pt, 0, gvA, gvB);
        if (pt + a.nsplit < a.ntiles) tile(pt + a.nsplit, 1, gvB, gvA);
    }
    // slab [split*4 + wave][group][co 32][col 32]; D[col][co]: lane li = co, register quad = 4 consecutive columns
    float* sl = a.slabs + ((long long)(blockIdx.x * 4 + wave) * WT_MAXG) * 1024;
#pragma unroll
    for (int gidx = 0; gidx < WT_MAXG; ++gidx) {
        if (gidx >= ngrp) break;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[gidx][4 * qd + k];
            f32x4* dst = reinterpret_cast<f32x4*>(sl + gidx * 1024 + li * 32 + 8 * qd + 4 * kh);
            if (a.accum) v += *dst;
            *dst = v;
        }
    }
    if (a.bslabs) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (kh == 0) {
            float* dst = a.bslabs + (long long)(blockIdx.x * 4 + wave) * 32 + li;
            *dst = a.accum ? *dst + bsum : bsum;
        }
    }
}

// dw[co][ci][tap] += sum_slabs D[grp][co][(t, ci)],  db[co] += sum_slabs; one wave per output element, fixed order
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bslabs,
                                                               int nslab, int Co, int Ci, int iTotal, int ntaps,
                                                               float* __restrict__ dw, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = Co * Ci * ntaps;
    if (e >= total + Co) return;
    float s = 0.f;
    if (e < total) {
        const int tap = e % ntaps, ci = (e / ntaps) % Ci, co = e / (ntaps * Ci);
        const long long off = (long long)(tap >> 3) * 1024 + co * 32 + (tap & 7) * 4 + ci;
        for (int k = lane; k < nslab; k += 64) s += slabs[(long long)k * WT_MAXG * 1024 + off];
    } else if (db != nullptr) {
        for (int k = lane; k < nslab; k += 64) s += bslabs[(long long)k * 32 + (e - total)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        if (e < total) {
            const int tap = e % ntaps, ci = (e / ntaps) % Ci, co = e / (ntaps * Ci);
            dw[((long long)co * iTotal + ci) * ntaps + tap] += s;
        } else if (db != nullptr) db[e - total] += s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Thin-OUTPUT 3x3 weight gradient (c_o <= 4: `pred`, 32 -> 3, XXNet_final_attenfusion_arch.py:215; its gradient has one real
// channel triple per pixel).  As a GEMM tile it fills 3 of 32 output-channel rows (the W3_32x32 plan ran at 0.07 of HBM,
// 913 us per grouped launch); it is a STREAMING REDUCTION: per input pixel q and lane = input channel i,
//     dW[o][i][ky][kx] += x[q][i] * g[q + (1 - ky, 1 - kx)][o]
// -- one coalesced 128-byte read of x per pixel, the 3 x 3 window of gradient pixels (one float4 each, the same address
// for the 32 lanes: a broadcast read) slides along the row, 9 NO multiply-adds per lane, nothing staged.  A workgroup =
// 8 half-waves, each walking whole image rows of a contiguous row range (split); the eight partial sums are added in order
// through LDS into the split's slab, laid out exactly like the MFMA tile's ([split][tap][CoP][CiP]), so phases, grouping
// and the deterministic slab reduction are the plan's own.
template <int NO>
__global__ __launch_bounds__(256) void wgrad_thinout_kernel(const WgKArgs a) {
    __shared__ float sred[8 * (NO * 9 * 32 + 4)];
    const int hw = threadIdx.x >> 5, li = threadIdx.x & 31;
    const int split = blockIdx.x;
    const long long rows = (long long)a.groups * a.N * a.H;
    const long long chunk = (rows + a.nsplit - 1) / a.nsplit;
    const long long r0 = split * chunk, r1 = min(r0 + chunk, rows);
    const bool iok = li < a.Ctot;
    float acc[NO][9];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[o][t] = 0.f;
    float bs[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) bs[o] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long row = r0 + hw; row < r1; row += 8) {
        const int grp = (int)(row / ((long long)a.N * a.H));
        const int rem = (int)(row - (long long)grp * a.N * a.H);
        const int y = rem % a.H;
        const float* xp = a.inA[grp] + (long long)rem * a.W * a.ldA + li;
        const float* gp = a.g[grp] + (long long)rem * a.W * a.ldG;              // gradient row y (same pixel grid: stride 1, pad 1)
        const bool up = y > 0, dn = y + 1 < a.H;
        // window rows: r = 0 -> gradient row y - 1 (taps ky = 2), 1 -> y, 2 -> y + 1 (ky = 0); columns slide
        auto gload = [&](int r, int x) -> f32x4 {
            const bool ok = (r == 1 || (r == 0 ? up : dn)) && x >= 0 && x < a.W;
            const float* p = gp + ((long long)(r - 1) * a.W + (ok ? x : 0)) * a.ldG;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? p : gp);
            return ok ? v : zero4;
        };
        // Four pixels per iteration: their x values and the four new gradient columns (12 broadcast float4) are requested
        // together, then consumed -- one memory latency per four pixels instead of per pixel (a half-wave walks its row
        // alone: 520 -> ~150 us per grouped launch).  Window columns: w[0] = x - 1, w[1] = x, w[2 .. 5] = x + 1 .. x + 4.
        f32x4 w[6][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { w[0][r] = zero4; w[1][r] = gload(r, 0); }
        for (int x0 = 0; x0 < a.W; x0 += 4) {
            float xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int r = 0; r < 3; ++r) w[2 + k][r] = gload(r, x0 + 1 + k);
                xv[k] = (iok && x0 + k < a.W) ? xp[(long long)(x0 + k) * a.ldA] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool live = x0 + k < a.W;             // (row tail: xv is 0 there, the bias sum must skip it too)
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    // tap (ky, kx) pairs with the gradient at (y + 1 - ky, x + 1 - kx) = window[2 - ky][2 - kx]
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        acc[o][ky * 3 + 0] = fmaf(xv[k], w[k + 2][2 - ky][o], acc[o][ky * 3 + 0]);
                        acc[o][ky * 3 + 1] = fmaf(xv[k], w[k + 1][2 - ky][o], acc[o][ky * 3 + 1]);
                        acc[o][ky * 3 + 2] = fmaf(xv[k], w[k][2 - ky][o], acc[o][ky * 3 + 2]);
                    }
                    if (live) bs[o] += w[k + 1][1][o];      // (every lane of the half-wave holds the same sum)
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) { w[0][r] = w[4][r]; w[1][r] = w[5][r]; }
        }
    }
    // the eight half-waves in order
    float* mine = sred + hw * (NO * 9 * 32 + 4);
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) mine[(o * 9 + t) * 32 + li] = acc[o][t];
    if (li == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) mine[NO * 9 * 32 + o] = bs[o];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NO * 9 * 32 + NO; e += 256) {
        const int src = e < NO * 9 * 32 ? e : NO * 9 * 32 + (e - NO * 9 * 32);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sred[k * (NO * 9 * 32 + 4) + src];
        if (e < NO * 9 * 32) {
            const int i = e & 31, t = (e >> 5) % 9, o = (e >> 5) / 9;
            if (i < a.CiP) {
                float* dst = a.slabs + (((long long)split * 9 + t) * a.CoP + o) * a.CiP + i;
                *dst = a.accum ? *dst + s : s;
            }
        } else if (a.bslabs != nullptr) {
            float* dst = a.bslabs + (long long)split * a.CoP + (e - NO * 9 * 32);
            *dst = a.accum ? *dst + s : s;
        }
    }
}

const bool USE_THINOUT_WGRAD = !(getenv("REFID_THINOUT_WGRAD") && getenv("REFID_THINOUT_WGRAD")[0] == '0');
bool thinout_ok(const refid_wgrad_desc* d) {
    return USE_THINOUT_WGRAD && d->algo == 0 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->c_o <= 4 &&
           d->c_b == 0 && d->c_a <= 32 && d->i_total - d->i_base <= 32 && d->ld_g % 4 == 0 && d->ho == d->h && d->wo == d->w;
}

bool thin_ok(const refid_wgrad_desc* d) {
    return USE_THIN_WGRAD && d->algo == 0 && d->kh == d->kw && d->kh * d->kw <= 32 && d->kh >= 3 && d->stride == 1 &&
           d->c_a == 4 && d->c_b == 0 && d->c_o <= 32 && d->i_base == 0 && d->i_total <= 4 && 2 * d->pad == d->kh - 1 &&
           (long long)d->n * d->h * d->w * d->ld_g * 4 < 0x7fffffffLL;
}
int thin_nsplit(const refid_wgrad_desc* d) {
    const int ntiles = cdiv(d->w, WT_TW) * cdiv(d->h, WT_TH) * d->n;
    return ntiles < 768 ? ntiles : 768;        // 3 resident workgroups per CU: one round
}

struct RedArgs {
    const float* slabs; const float* bslabs; float* dw; float* db;
    int nsplit, ntaps, Co, Ci, CoP, CiP, iBase, iTotal, perGroup;
    int nsplitW;               // slabs behind `slabs` (nsplit, or the folded count); bslabs always has nsplit rows
    int permCo;                // algo 8 (ntaps = 1): slab column k = (tap, co) goes to dw column (k % permCo) * 4 + k / permCo; 0 = none
};

// Slab reduction, deterministic.  `perGroup` = LPE (a power of two <= 8) adjacent lanes share ONE float4 element (4
// consecutive input channels): lane `sub` adds slabs sub, sub + LPE, ... in order, a fixed xor-shuffle tree combines the
// LPE partial sums and lane 0 adds the total into the parameter-layout gradient (which it owns: no atomics).  LPE > 1
// only for small weight tensors, where one thread per element would leave the chip idle.
__device__ __forceinline__ void wgrad_reduce_body(const RedArgs& a, const int blk) {
    const int Ci4 = a.CiP / 4;
    const long long total4 = (long long)a.ntaps * a.CoP * Ci4;
    const long long slabStride4 = total4;
    const int lpe = a.perGroup;
    const long long gid = blk * 256ll + threadIdx.x;
    const long long e = gid / lpe;
    const int sub = (int)(gid % lpe);
    {
        const bool in = e < total4;
        long long r = in ? e : 0;
        const int ci = (int)(r % Ci4) * 4; r /= Ci4;
        const int co = (int)(r % a.CoP);
        const int tap = (int)(r / a.CoP);
        const bool live = in && co < a.Co && ci < a.Ci;
        const f32x4* p = reinterpret_cast<const f32x4*>(a.slabs) + (in ? e : 0);
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) {
            int k = sub;
            for (; k + 3 * lpe < a.nsplitW; k += 4 * lpe) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += p[(long long)(k + u * lpe) * slabStride4];
            }
            for (; k < a.nsplitW; k += lpe) acc[0] += p[(long long)k * slabStride4];
        }
        f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        for (int o = 1; o < lpe; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[j] += __shfl_xor(sum[j], o, 64);
        }
        if (live && sub == 0) {
            if (a.permCo) {                                  // [ci][(tap, co)] -> IOHW [ci][co][tap] (ConvTranspose2d, algo 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = ci + j;
                    if (k >= a.Ci) break;
                    a.dw[(long long)co * a.iTotal + (k % a.permCo) * 4 + k / a.permCo] += sum[j];
                }
            } else {
                float* d = a.dw + ((long long)co * a.iTotal + a.iBase + ci) * a.ntaps + tap;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ci + j >= a.Ci) break;
                    d[(long long)j * a.ntaps] += sum[j];
                }
            }
        }
    }
    const int s0 = 0, s1 = a.nsplit;
    if (a.db != nullptr && blk == 0) {
        for (int co = threadIdx.x; co < a.Co; co += 256) {
            float s = 0.f;
            for (int k = s0; k < s1; ++k) s += a.bslabs[(long long)k * a.CoP + co];
            a.db[co] += s;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedArgs a) { wgrad_reduce_body(a, blockIdx.x); }

// every queued reduction in ONE launch (phase 4 + refid_wgrad_finish_flush): a workgroup finds its job by its block range
struct RedBatch { RedArgs job[REFID_FINISH_BATCH]; int blk0[REFID_FINISH_BATCH + 1]; int n; };
static_assert(sizeof(RedBatch) <= 4096, "kernel-argument block");
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const RedBatch b) {
    int j = 0;
    for (int k = 1; k < b.n; ++k) j = (int)blockIdx.x >= b.blk0[k] ? k : j;       // (workgroup-uniform)
    wgrad_reduce_body(b.job[j], (int)blockIdx.x - b.blk0[j]);
}

struct RedQueued { RedArgs r; int nblocks; };
thread_local std::vector<RedQueued> red_queue;
thread_local bool defer_now = false;

int red_flush(hipStream_t st) {
    size_t at = 0;
    while (at < red_queue.size()) {
        RedBatch b;
        memset(&b, 0, sizeof(b));
        int n = 0, blk = 0;
        for (; n < REFID_FINISH_BATCH && at < red_queue.size(); ++n, ++at) {
            b.job[n] = red_queue[at].r;
            b.blk0[n] = blk;
            blk += red_queue[at].nblocks;
        }
        b.blk0[n] = blk; b.n = n;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blk), dim3(256), 0, st, b);
        if (hipGetLastError() != hipSuccess) { red_queue.clear(); refid_set_error("wgrad_reduce_batch: launch failed"); return 1; }
    }
    red_queue.clear();
    return 0;
}

// <KH,KW,S, TPW, SM,SN, WR,WC,WT, TH,TW>
using W3 = WCfg<3, 3, 1, 9, 1, 1, 2, 2, 1, 2, 32>;          // 64 x 64 channels, 9 taps per wave
using W3_64x32 = WCfg<3, 3, 1, 5, 1, 1, 2, 1, 2, 2, 32>;    // narrow input  (Ci <= 32): taps split over 2 waves
using W3_32x64 = WCfg<3, 3, 1, 5, 1, 1, 1, 2, 2, 2, 32>;    // narrow output (Co <= 32)
using W3_32x32 = WCfg<3, 3, 1, 3, 1, 1, 1, 1, 4, 2, 32>;    // both narrow: taps split over 4 waves
using W1 = WCfg<1, 1, 1, 1, 2, 2, 2, 2, 1, 2, 32>;
using W4S2 = WCfg<4, 4, 2, 8, 1, 1, 2, 1, 2, 2, 16>;
using W5 = WCfg<5, 5, 1, 7, 1, 1, 1, 1, 4, 2, 32>;
using W2S2 = WCfg<2, 2, 2, 4, 1, 1, 2, 2, 1, 2, 16>;

enum PlanId { P_W3, P_W3_64x32, P_W3_32x64, P_W3_32x32, P_W1, P_W4S2, P_W5, P_W2S2, P_PW, P_NONE };
const bool USE_PW_WGRAD = !(getenv("REFID_PW_WGRAD") && getenv("REFID_PW_WGRAD")[0] == '0');
struct Plan { PlanId id; int cot, cit, th, tw, ntaps; bool ok; };

template <class C>
Plan mk(PlanId id) { return {id, C::COT, C::CIT, C::TH, C::TW, C::NTAPS, true}; }

Plan plan_of(int kh, int kw, int s, int co, int ci) {
    if (kh == 3 && kw == 3 && s == 1) {
        if (co <= 32 && ci <= 32) return mk<W3_32x32>(P_W3_32x32);
        if (ci <= 32) return mk<W3_64x32>(P_W3_64x32);
        if (co <= 32) return mk<W3_32x64>(P_W3_32x64);
        return mk<W3>(P_W3);
    }
    if (kh == 1 && kw == 1 && s == 1) {
        if (USE_PW_WGRAD && co % 32 == 0 && ci % 32 == 0) {
            // register-operand tile: WR x WC waves of 32 x 32*SN channels (see wgrad_pw_kernel)
            const int wr = co >= 128 ? 4 : (co >= 64 ? 2 : 1), wc = 4 / wr;
            int sn = ci / (32 * wc);
            sn = sn >= 2 ? 2 : 1;                   // SN = 4 costs occupancy (PD x 5 operand registers)
            // only where one workgroup tile covers the whole weight matrix: with several channel tiles every
            // operand is re-streamed per tile and the LDS-tiled kernel's 128 x 128 tile wins (measured)
            if (co <= 32 * wr && ci <= 32 * sn * wc) return {P_PW, 32 * wr, 32 * sn * wc, 2, 32, 1, true};
        }
        return mk<W1>(P_W1);
    }
    if (kh == 4 && kw == 4 && s == 2) return mk<W4S2>(P_W4S2);
    if (kh == 5 && kw == 5 && s == 1) return mk<W5>(P_W5);
    if (kh == 2 && kw == 2 && s == 2) return mk<W2S2>(P_W2S2);
    return {P_NONE, 0, 0, 0, 0, 0, false};
}

struct Geo { int ncoT, nciT, tilesX, tilesY, ntiles, nsplit, CoP, CiP; };

Geo geo_of(const refid_wgrad_desc* d, const Plan& p) {
    Geo g;
    g.ncoT = cdiv(d->c_o, p.cot);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;   // stable across steps
    g.nciT = cdiv(ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b, p.cit);
    g.tilesX = cdiv(d->wo, p.tw);
    g.tilesY = cdiv(d->ho, p.th);
    g.ntiles = g.tilesX * g.tilesY * d->n;
    int want = cdiv(512, g.ncoT * g.nciT);           // 2 resident workgroups per CU on 256 CUs, one round
    if (want < 1) want = 1;
    if (want > g.ntiles) want = g.ntiles;
    g.nsplit = want;
    g.CoP = g.ncoT * p.cot;
    g.CiP = g.nciT * p.cit;
    return g;
}

template <class C>
int launch_w(const WgKArgs& a, const Geo& g, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_kernel<C>, C::LDS_BYTES, "wgrad")) return rc;
    dim3 grid(g.nsplit, g.nciT, g.ncoT);
    hipLaunchKernelGGL(wgrad_kernel<C>, grid, dim3(256), C::LDS_BYTES, st, a);
    REFID_LAUNCH_CHECK("wgrad");
    return 0;
}

}  // namespace

// wgrad_wino.hip
size_t refid_wgrad_wino_workspace_bytes(const refid_wgrad_desc* d);
int refid_wgrad_wino_launch(const refid_wgrad_desc* d, hipStream_t st);
// wgrad_wino24.hip: Winograd over 2x4 tiles, F(3,2) x F(3,4) (algo 5)
size_t refid_wgrad_wino24_workspace_bytes(const refid_wgrad_desc* d);
int refid_wgrad_wino24_launch(const refid_wgrad_desc* d, hipStream_t st);

// algo 8: the weight gradient of a 2x2 stride-2 conv over NON-overlapping patches (ConvTranspose2d(2,2) with the roles swapped,
// refid_hip.h) as ONE streaming 1x1 weight gradient: an output pixel's patch is two contiguous runs of 2 c_a floats (rows 2y and
// 2y + 1 of dense pixels), i.e. a two-source 1x1 problem with 4 c_a "input channels" k = (dy, dx, c) over row-pitched sources
// (WgKArgs.patchW / patchRow); the reduction permutes the columns into the [o][c][dy][dx] gradient.  Returns false when the
// geometry does not fit (caller: error).
static bool patch_translate(const refid_wgrad_desc* d, refid_wgrad_desc* t, int* patchW, int* patchRow) {
    if (d->kh != 2 || d->kw != 2 || d->stride != 2 || d->pad != 0 || d->c_b != 0 || d->ld_a != d->c_a || d->h != 2 * d->ho ||
        d->w != 2 * d->wo || d->i_base != 0 || d->i_total != d->c_a || d->db != nullptr)
        return false;
    *t = *d;
    t->algo = 0; t->kh = t->kw = 1; t->stride = 1; t->pad = 0;
    t->h = d->ho; t->w = d->wo;
    t->in_b = d->in_a + (long long)d->w * d->ld_a;
    t->ld_a = t->ld_b = 2 * d->ld_a;
    t->c_a = t->c_b = 2 * d->c_a;
    t->i_total = 4 * d->c_a;
    for (int k = 0; k + 1 < REFID_WGRAD_MAX_GROUPS; ++k)
        t->in_b_more[k] = d->in_a_more[k] ? d->in_a_more[k] + (long long)d->w * d->ld_a : nullptr;
    *patchW = d->wo; *patchRow = 2 * d->w * d->ld_a;
    if (!refid_wgrad_pws_ok(t) || (long long)d->n * d->h * d->w * d->ld_a * 4 >= 0x7fffffffLL) return false;
    // a patch row must fill whole ring buffers (the launch checks the same: the workspace query and the launch agree)
    const int pb = refid_wgrad_pws_pixels_per_buffer(t);
    return pb > 0 && *patchW % pb == 0;
}

extern "C" size_t refid_wgrad_workspace_bytes(const refid_wgrad_desc* d) {
    if (!d) return 0;
    refid_wgrad_desc tt;
    if (d->algo == 8) {
        int pw, pr;
        if (!patch_translate(d, &tt, &pw, &pr)) return 0;
        d = &tt;
    }
    if (d->algo == 1 || d->algo == 3 || d->algo == 4) return (d->kh == 3 && d->kw == 3 && d->stride == 1) ? refid_wgrad_wino_workspace_bytes(d) : 0;
    if (d->algo == 5) return (d->kh == 3 && d->kw == 3 && d->stride == 1) ? refid_wgrad_wino24_workspace_bytes(d) : 0;
    if (d->algo == 7) return (d->kh == 4 && d->kw == 4 && d->stride == 2) ? refid_wgrad_wino24_workspace_bytes(d) : 0;
    if (thin_ok(d)) return (size_t)thin_nsplit(d) * 4 * (WT_MAXG * 1024 + 32) * sizeof(float);
    const Plan p = plan_of(d->kh, d->kw, d->stride, d->c_o, d->phase != 0 ? d->i_total - d->i_base : d->c_a + d->c_b);
    if (!p.ok) return 0;
    Geo g = geo_of(d, p);
    if (refid_wgrad_pws_ok(d)) refid_wgrad_pws_geo(d, &g.ncoT, &g.nciT, &g.nsplit, &g.CoP, &g.CiP);
    const size_t slab = (size_t)p.ntaps * g.CoP * g.CiP;
    return ((size_t)g.nsplit * slab + (size_t)g.nsplit * g.CoP + (size_t)refid_slab_fold_count((long long)slab, g.nsplit) * slab) * sizeof(float);
}

bool refid_finish_defer_now() { return defer_now; }

static int conv2d_wgrad_impl(const refid_wgrad_desc* d, hipStream_t st);

extern "C" int refid_conv2d_wgrad(const refid_wgrad_desc* d, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REFID_CHECK(d != nullptr, "wgrad: null descriptor");
    if (d->phase != 4) return conv2d_wgrad_impl(d, st);
    // phase 4 = phase 3 with the element-wise stage queued for refid_wgrad_finish_flush (the families that have no batched
    // form -- algo 1's tile, the thin-input tile -- run theirs at once)
    refid_wgrad_desc dd = *d;
    dd.phase = 3;
    defer_now = true;
    const int rc = conv2d_wgrad_impl(&dd, st);
    defer_now = false;
    return rc;
}

extern "C" int refid_wgrad_finish_flush(void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc0 = refid_slab_fold_flush(st);             // first stages before the element-wise stages that read them
    const int rc = red_flush(st);
    const int rc2 = refid_wino24_finish_flush(st);
    return rc0 ? rc0 : (rc ? rc : rc2);
}

static int conv2d_wgrad_impl(const refid_wgrad_desc* d, hipStream_t st) {
    refid_wgrad_desc tt;
    int patchW = 0, patchRow = 0, permCo = 0;
    if (d->algo == 8) {
        REFID_CHECK(d->g && d->in_a && d->dw && d->slabs, "wgrad: null tensor pointer");
        REFID_CHECK(patch_translate(d, &tt, &patchW, &patchRow),
                    "wgrad (algo 8): a 2x2 stride-2 pad-0 conv over dense pixels (ld_a == c_a, one source, no bias), even input sizes, "
                    "c_o >= 64 and a multiple of 32, c_a a multiple of 16");
        permCo = d->c_a;
        d = &tt;
    }
    const Plan p = plan_of(d->kh, d->kw, d->stride, d->c_o, d->phase != 0 ? d->i_total - d->i_base : d->c_a + d->c_b);
    REFID_CHECK(p.ok, "wgrad: unsupported geometry k=%dx%d stride=%d", d->kh, d->kw, d->stride);
    REFID_CHECK(d->g && d->in_a && d->dw && d->slabs, "wgrad: null tensor pointer");
    REFID_CHECK(d->c_o > 0 && d->c_o % 4 == 0 && d->c_a > 0 && d->c_a % 4 == 0 && d->c_b >= 0 && d->c_b % 4 == 0,
                "wgrad: channel counts must be multiples of 4 (c_o=%d c_a=%d c_b=%d)", d->c_o, d->c_a, d->c_b);
    REFID_CHECK(d->ld_g % 4 == 0 && d->ld_a % 4 == 0 && (d->c_b == 0 || (d->in_b && d->ld_b % 4 == 0)),
                "wgrad: bad pitches / missing in_b");
    const int eh = (d->h + 2 * d->pad - d->kh) / d->stride + 1;
    const int ew = (d->w + 2 * d->pad - d->kw) / d->stride + 1;
    REFID_CHECK(eh == d->ho && ew == d->wo, "wgrad: output size %dx%d does not match geometry (%dx%d)", d->ho,
                d->wo, eh, ew);
    REFID_CHECK(d->i_total > d->i_base && d->i_base >= 0 && d->o_real > 0 && d->o_real <= d->c_o,
                "wgrad: i_base/i_total/o_real inconsistent");
    REFID_CHECK(d->algo == 0 || ((d->algo >= 1 && d->algo <= 6) && d->kh == 3 && d->kw == 3 && d->stride == 1) ||
                    (d->algo == 7 && d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1),
                "wgrad: algo %d needs a 3x3 stride-1 conv (algo 7: a 4x4 stride-2 pad-1 conv)", d->algo);
    REFID_CHECK(d->groups <= REFID_WGRAD_MAX_GROUPS, "wgrad: at most %d grouped time steps", REFID_WGRAD_MAX_GROUPS);
    const bool pws = refid_wgrad_pws_ok(d);               // streaming 1x1 form (wgrad_pws.hip): its own slab geometry
    REFID_CHECK(d->groups <= 1 || (d->phase != 3 && !thin_ok(d) && (d->algo != 0 || p.id != P_PW || pws)),
                "wgrad: grouped time steps are not implemented by the thin-input and 1x1 register tiles (and mean nothing in phase 3)");
    REFID_CHECK(d->algo != 2 || (p.id == P_W3 && d->pad == 1),
                "wgrad: algo 2 (bf16 operands) needs more than 32 output and input channels and pad 1");
    if (d->algo == 1 || d->algo == 3 || d->algo == 4) return refid_wgrad_wino_launch(d, st);
    if (d->algo == 5 || d->algo == 7) return refid_wgrad_wino24_launch(d, st);
    REFID_CHECK(d->algo != 6, "wgrad: algo 6 (Winograd F(3x3,4x4)) was an experiment that did not beat algo 5 (round 5, "
                              "DESIGN.md section 7); it is no longer built");
    if (thin_ok(d)) {
        REFID_CHECK(d->phase >= 0 && d->phase <= 3, "wgrad: bad phase %d", d->phase);
        const int ns = thin_nsplit(d), nslab = ns * 4;
        WtArgs t;
        t.g = d->g; t.ldG = d->ld_g; t.Co = d->c_o; t.x = d->in_a; t.ldX = d->ld_a;
        t.slabs = d->slabs; t.bslabs = d->db ? d->slabs + (size_t)nslab * WT_MAXG * 1024 : nullptr;
        t.N = d->n; t.H = d->h; t.W = d->w; t.pad = d->pad; t.K = d->kh;
        t.tilesX = cdiv(d->w, WT_TW); t.tilesY = cdiv(d->h, WT_TH); t.ntiles = t.tilesX * t.tilesY * d->n;
        t.nsplit = ns; t.accum = (d->phase == 2);
        if (d->phase != 3) {
            const int lds = 2 * (WT_TH + d->kh - 1) * (WT_TW + d->kh - 1) * 16;
            hipLaunchKernelGGL(wgrad_thin_kernel, dim3(ns), dim3(256), lds, st, t);
            REFID_LAUNCH_CHECK("wgrad_thin");
        }
        if (d->phase == 1 || d->phase == 2) return 0;
        const int ci = d->i_total, total = d->o_real * ci * d->kh * d->kw + d->o_real;
        hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(cdiv(total, 4)), dim3(256), 0, st, t.slabs, t.bslabs, nslab,
                           d->o_real, ci, d->i_total, d->kh * d->kw, d->dw, d->db);
        REFID_LAUNCH_CHECK("wgrad_thin_reduce");
        return 0;
    }
    Geo g = geo_of(d, p);
    if (pws) refid_wgrad_pws_geo(d, &g.ncoT, &g.nciT, &g.nsplit, &g.CoP, &g.CiP);
    WgKArgs a;
    const int ngrp = d->groups > 1 ? d->groups : 1;
    for (int k = 0; k < REFID_WGRAD_MAX_GROUPS; ++k) {
        const bool on = k > 0 && k < ngrp;
        a.g[k] = on ? d->g_more[k - 1] : d->g;
        a.inA[k] = on ? d->in_a_more[k - 1] : d->in_a;
        a.inB[k] = on ? d->in_b_more[k - 1] : d->in_b;
        REFID_CHECK(a.g[k] && a.inA[k] && (d->c_b == 0 || a.inB[k]), "wgrad: null tensor pointer in group %d", k);
    }
    a.groups = ngrp;
    a.ldG = d->ld_g; a.Co = d->c_o;
    a.ldA = d->ld_a; a.ldB = d->ld_b;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.slabs = d->slabs;
    a.bslabs = d->db ? d->slabs + (size_t)g.nsplit * p.ntaps * g.CoP * g.CiP : nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pad = d->pad;
    a.tilesX = g.tilesX; a.tilesY = g.tilesY; a.ntiles = g.ntiles; a.nsplit = g.nsplit;
    a.CoP = g.CoP; a.CiP = g.CiP;
    a.accum = (d->phase == 2);
    a.patchW = patchW; a.patchRow = patchRow;
    REFID_CHECK(!patchW || pws, "wgrad (algo 8): the streaming 1x1 form is switched off (REFID_PWS_WGRAD)");
    REFID_CHECK(d->phase >= 0 && d->phase <= 3, "wgrad: bad phase %d", d->phase);
    int rc = (d->phase == 3) ? 0 : 1;
    if (d->phase != 3 && pws) {
        rc = refid_wgrad_pws_launch(d, a, g.nciT, g.ncoT, st);
    } else if (d->phase != 3 && p.id == P_PW) {
        const long long npix = (long long)d->n * d->h * d->w;
        REFID_CHECK(d->c_b == 0 || d->c_a % 32 == 0, "wgrad: pointwise tile needs c_a %% 32 == 0 for two sources");
        REFID_CHECK(npix * d->ld_g * 4 < 0x7fffffffLL && npix * d->ld_a * 4 < 0x7fffffffLL &&
                        (d->c_b == 0 || npix * d->ld_b * 4 < 0x7fffffffLL),
                    "wgrad: tensor too large for the pointwise tile's 32-bit offsets");
        WpArgs w;
        w.g = d->g; w.ldG = d->ld_g; w.Co = d->c_o;
        w.inA = d->in_a; w.inB = d->in_b; w.ldA = d->ld_a; w.ldB = d->ld_b; w.Ca = d->c_a; w.Ctot = d->c_a + d->c_b;
        w.slabs = a.slabs; w.bslabs = a.bslabs;
        w.npix = npix;
        long long chunk = (npix + g.nsplit - 1) / g.nsplit;
        w.chunk = (int)((chunk + 1) & ~1ll);
        w.CoP = g.CoP; w.CiP = g.CiP; w.WR = p.cot / 32; w.accum = a.accum;
        const int sn = p.cit / (32 * (4 / w.WR));
        dim3 grid(g.nsplit, g.nciT, g.ncoT);
        if (sn == 2) hipLaunchKernelGGL(wgrad_pw_kernel<2>, grid, dim3(256), 0, st, w);
        else hipLaunchKernelGGL(wgrad_pw_kernel<1>, grid, dim3(256), 0, st, w);
        REFID_LAUNCH_CHECK("wgrad_pw");
        rc = 0;
    } else if (d->phase != 3 && p.id == P_W3_32x32 && thinout_ok(d)) {
        for (int k = 0; k < ngrp; ++k)
            REFID_CHECK((uintptr_t)a.g[k] % 16 == 0, "wgrad (thin output): the gradient tensor must be 16-byte aligned (group %d)", k);
        dim3 grid(g.nsplit);
        switch (d->o_real) {
            case 1: hipLaunchKernelGGL(wgrad_thinout_kernel<1>, grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(wgrad_thinout_kernel<2>, grid, dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL(wgrad_thinout_kernel<3>, grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(wgrad_thinout_kernel<4>, grid, dim3(256), 0, st, a); break;
        }
        REFID_LAUNCH_CHECK("wgrad_thinout");
        rc = 0;
    } else if (d->phase != 3 && d->algo == 2) {
        rc = refid_wgrad_bf16_launch(a, g.nciT, g.ncoT, st);
    } else if (d->phase != 3) switch (p.id) {
        case P_W3: rc = launch_w<W3>(a, g, st); break;
        case P_W3_64x32: rc = launch_w<W3_64x32>(a, g, st); break;
        case P_W3_32x64: rc = launch_w<W3_32x64>(a, g, st); break;
        case P_W3_32x32: rc = launch_w<W3_32x32>(a, g, st); break;
        case P_W1: rc = launch_w<W1>(a, g, st); break;
        case P_W4S2: rc = launch_w<W4S2>(a, g, st); break;
        case P_W5: rc = launch_w<W5>(a, g, st); break;
        case P_W2S2: rc = launch_w<W2S2>(a, g, st); break;
        default: break;
    }
    if (rc) return rc;
    if (d->phase == 1 || d->phase == 2) return 0;          // reduction deferred (phase 3)
    RedArgs r;
    r.slabs = a.slabs; r.bslabs = a.bslabs; r.dw = d->dw; r.db = d->db;
    // channel-padded operands (e.g. 26 -> 28 image channels, 3 -> 4 output channels): only the
    // real rows / columns of the parameter-layout gradient exist
    r.nsplit = g.nsplit; r.ntaps = p.ntaps; r.Co = d->o_real;
    r.nsplitW = g.nsplit;
    r.permCo = permCo;
    {   // streaming first stage: S partial slabs (wgrad_wino24.hip::refid_launch_slab_fold; the element-wise stage below read
        // 100 MB of slabs at 0.09 of HBM)
        const long long slab = (long long)p.ntaps * g.CoP * g.CiP;
        if (const int S = refid_slab_fold_count(slab, g.nsplit)) {
            float* part = d->slabs + (size_t)g.nsplit * slab + (size_t)g.nsplit * g.CoP;
            if (int rc2 = refid_launch_slab_fold(a.slabs, part, slab, g.nsplit, S, st)) return rc2;
            r.slabs = part;
            r.nsplitW = S;
        }
    }
    r.Ci = (d->phase == 0 && a.Ctot < d->i_total - d->i_base) ? a.Ctot : d->i_total - d->i_base;
    r.CoP = g.CoP; r.CiP = g.CiP;
    r.iBase = d->i_base; r.iTotal = d->i_total;
    const long long total4 = (long long)p.ntaps * g.CoP * (g.CiP / 4);
    const int nb = (int)((total4 + 255) / 256);
    // lanes per element: only where one thread per float4 would leave the chip idle (small weight tensors)
    int lpe = 1;
    while (lpe < 8 && (long long)lpe * 2 * total4 <= 65536 && lpe * 2 <= r.nsplitW) lpe *= 2;
    r.perGroup = lpe;
    const int nblocks = (int)((total4 * lpe + 255) / 256);
    if (defer_now) {
        // two queued jobs must not add into the same gradient block (they would run concurrently): flush first
        for (const RedQueued& q : red_queue)
            if (q.r.dw == r.dw && q.r.iBase == r.iBase) {
                if (int rc2 = refid_slab_fold_flush(st)) return rc2;
                if (int rc2 = red_flush(st)) return rc2;
                break;
            }
        red_queue.push_back({r, nblocks});
        return 0;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblocks), dim3(256), 0, st, r);
    REFID_LAUNCH_CHECK("wgrad_reduce");
    return 0;
}
