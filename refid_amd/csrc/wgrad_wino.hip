// Winograd F(2x2,3x3) weight gradient of the 3x3 / stride-1 convolutions on the fp32 matrix cores.
//
//   dg[o][i] = G^T [ sum_{tiles} (A dY A^T)_xi[o] * (B^T d B)_xi[i] ] G          (xi = 0..15)
//
// i.e. 16 transform-domain GEMMs  dU_xi[o][i] += Z_xi[tile][o] * V_xi[tile][i]  over 2x2-pixel
// tiles (K = tiles) instead of 9 tap GEMMs over pixels (K = pixels): 16 MFMAs per
// (32o x 32i x 2 tiles) instead of 36.  The inverse transform G^T dU G (16 -> 9 values) is applied
// once per call by the slab reduction.
//
// Mapping: workgroup = 256 threads = 4 waves; wave w owns transform row i = w (xi = 4w..4w+3) for
// the whole 64(o) x 32(i) channel tile: 8 accumulators = 128 AGPRs.  Because a wave only needs ITS
// row of Z and V, both transforms are done on the fly in registers straight from the raw NHWC LDS
// tiles (gradient tile 4x32 pixels x 64 o, input halo 6x34 pixels x 32 i; conflict-free
// ds_read_b32 of 32 consecutive channels) -- no transformed copy is ever stored.  Split-K over pixel
// tiles into private slabs [split][xi][o][i]; bias gradient rides along as in conv_wgrad.hip.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int TH = 4, TW = 32;                 // output pixels per K tile (2 x 16 Winograd tiles)
constexpr int TH_FP32 = TH;
constexpr int COT = 64, CIT = 32;
constexpr int PX = TH * TW;
constexpr int HWD = TW + 2, HP = (TH + 2) * HWD;
constexpr int G4 = COT / 4, X4 = CIT / 4;
constexpr int G_TOTAL = PX * G4, X_TOTAL = HP * X4;
constexpr int G_ITEMS = G_TOTAL / 256, X_ITEMS = (X_TOTAL + 255) / 256;
constexpr int LDS_BYTES = (G_TOTAL + X_TOTAL) * 16 + COT * 4;

struct WwArgs {
    // up to REFID_WGRAD_MAX_GROUPS time steps of the same convolution (same geometry): their tiles are one K range
    const float* g[REFID_WGRAD_MAX_GROUPS]; const float* inA[REFID_WGRAD_MAX_GROUPS]; const float* inB[REFID_WGRAD_MAX_GROUPS];
    int groups;
    int ldG, Co;
    int ldA, ldB, Ca, Ctot;
    float* slabs; float* bslabs;
    int N, H, W, Ho, Wo, pad;
    int tilesX, tilesY, ntiles, nsplit;
    int CoP, CiP;
    int accum;
};

__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sG4 = reinterpret_cast<f32x4*>(smem);
    f32x4* sX4 = sG4 + G_TOTAL;
    float* sBias = reinterpret_cast<float*>(sX4 + X_TOTAL);
    const float* sG = reinterpret_cast<const float*>(sG4);
    const float* sX = reinterpret_cast<const float*>(sX4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row owned by this wave
    const int co0 = blockIdx.z * COT, ci0 = blockIdx.y * CIT;
    const int split = blockIdx.x;

    // B^T rows: i=0: d0-d2  i=1: d1+d2  i=2: d2-d1  i=3: d1-d3 ;  A rows: X_b = ca*dY0b + cb*dY1b
    const int ra = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rb = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const float ca = (ti == 3) ? 0.f : 1.f;
    const float cb = (ti == 0) ? 0.f : ((ti == 1) ? 1.f : -1.f);

    const int gq = tid % G4, xq = tid % X4;
    const int gco = co0 + gq * 4;
    const bool gcok = gco < a.Co;
    const int xc = ci0 + xq * 4;
    const bool xcok = xc < a.Ctot;
    // the input-channel tile lies in one source (host: c_a % 32 == 0 for two sources), so the descriptor is
    // workgroup-uniform; a tile beyond the sources (first recurrent step) keeps a valid descriptor, all lanes out of range
    const bool xFromA = ci0 < a.Ca || ci0 >= a.Ctot;
    const int xld = xFromA ? a.ldA : a.ldB;
    const int xcc = xFromA ? xc : xc - a.Ca;

    f32x16 acc[4][2];                                      // [j][o sub-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][sm][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 rg[G_ITEMS], rx[X_ITEMS];

    // Tile loads: buffer loads with 32-bit offsets.  A thread's pixel inside the tile never changes, so its offset is
    // (tile origin pixel) * pitch + a per-thread constant; out-of-image pixels / channels get the out-of-range offset
    // and come back as zeros (no per-lane branches around loads, no 64-bit address arithmetic per item).
    const long long gpixAll = (long long)a.N * a.Ho * a.Wo, xpixAll = (long long)a.N * a.H * a.W;
    const int limG = (int)min(gpixAll * a.ldG * 4, 0x7fffffffLL), limX = (int)min(xpixAll * xld * 4, 0x7fffffffLL);
    const int ntAll = a.ntiles * a.groups;                 // tiles of all grouped time steps
    auto load_tile = [&](int pt) {
        const int grp = pt / a.ntiles;                     // workgroup-uniform: which time step this tile belongs to
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[grp]), 0, limG, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xFromA ? a.inA[grp] : a.inB[grp]), 0, limX, 0x00020000);
        int t = pt - grp * a.ntiles;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        const int n = t / a.tilesY;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
        const int gbase = ((n * a.Ho + oy0) * a.Wo + ox0) * a.ldG * 4 + gco * 4;          // bytes, < 2^31 (host check)
        const int xbase = ((n * a.H + iy0) * a.W + ix0) * xld * 4 + xcc * 4;
#pragma unroll
        for (int it = 0; it < G_ITEMS; ++it) {
            const int p = tid / G4 + it * (256 / G4);
            const int dy = p / TW, dx = p % TW;
            const bool ok = gcok && oy0 + dy < a.Ho && ox0 + dx < a.Wo;
            const int vo = ok ? gbase + (dy * a.Wo + dx) * a.ldG * 4 : -1;
            rg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < X_ITEMS; ++it) {
            const int hp = tid / X4 + it * (256 / X4);
            const int dy = hp / HWD, dx = hp % HWD;
            const int iy = iy0 + dy, ix = ix0 + dx;
            const bool ok = xcok && hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int vo = ok ? xbase + (dy * a.W + dx) * xld * 4 : -1;
            rx[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, 0, 0));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G_ITEMS; ++it) {
            const int p = tid / G4 + it * (256 / G4);
            sG4[p * G4 + gq] = rg[it];
            bsum += rg[it];          // bias partial: summed HERE, not at load time -- using a prefetched register right
                                     // after its load was issued forced a vmcnt(0) before the MFMA section
        }
#pragma unroll
        for (int it = 0; it < X_ITEMS; ++it) {
            const int hp = tid / X4 + it * (256 / X4);
            if (hp < HP) sX4[hp * X4 + xq] = rx[it];
        }
    };

    int pt = split;
    if (pt < ntAll) {
        load_tile(pt);
        store_tile();
    }
    __syncthreads();

    for (; pt < ntAll; pt += a.nsplit) {
        const bool more = pt + a.nsplit < ntAll;
        if (more) load_tile(pt + a.nsplit);

        // 16 K steps per K tile, fully unrolled (every LDS address is base + immediate).  A step pairs the two tile
        // ROWS of one tile column (MFMA K half kh = tile row), so consecutive steps of a lane walk along a row and
        // the 4-column input window slides by 2: only 2 new columns per row are read per step (12 instead of 16
        // LDS floats per 8 MFMAs -- LDS bytes and VALU instructions do not hide under MFMAs on gfx950).
        // The raw operands of step s+1 are fetched before the 8 MFMAs of step s issue.
        const float* xA = sX + ((2 * kh + ra) * HWD) * CIT + li;     // + (2*s + b) * CIT
        const float* xB = sX + ((2 * kh + rb) * HWD) * CIT + li;
        const float* gP = sG + (2 * kh * TW) * COT + li;             // + (2*s) * COT
        // T_b = d[ra][b] + sgn d[rb][b] for the 34 window columns of this lane's row pair, consumed two per step; the
        // step loop is fully unrolled, so tw[] / gbuf[] are plain registers (no copies between steps).
        float tw[4];
        float gbuf[2][2][4];
        auto fetch_g = [&](int s, float (&pg)[2][4]) {
            const int go = (2 * s) * COT;
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
                pg[sm][0] = gP[go + sm * 32];            pg[sm][1] = gP[go + sm * 32 + COT];
                pg[sm][2] = gP[go + sm * 32 + TW * COT]; pg[sm][3] = gP[go + sm * 32 + TW * COT + COT];
            }
        };
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) tw[b4] = xA[b4 * CIT] + sgn * xB[b4 * CIT];
        fetch_g(0, gbuf[0]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float n2 = 0.f, n3 = 0.f;
            if (s + 1 < 16) {
                fetch_g(s + 1, gbuf[(s + 1) & 1]);
                n2 = xA[(2 * s + 4) * CIT] + sgn * xB[(2 * s + 4) * CIT];
                n3 = xA[(2 * s + 5) * CIT] + sgn * xB[(2 * s + 5) * CIT];
            }
            const float (&cg)[2][4] = gbuf[s & 1];
            // column transform; the 4th operand carries the sign of Z's 4th column (z3 = -x1), so Z needs no negation
            const float v[4] = {tw[0] - tw[2], tw[1] + tw[2], tw[2] - tw[1], tw[3] - tw[1]};
            // Z row ti:  X_b = ca dY[0][b] + cb dY[1][b];  z = {x0, x0 + x1, x0 - x1, (-)x1}
            float z[2][4];
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
                const float x0 = ca * cg[sm][0] + cb * cg[sm][2];
                const float x1 = ca * cg[sm][1] + cb * cg[sm][3];
                z[sm][0] = x0; z[sm][1] = x0 + x1; z[sm][2] = x0 - x1; z[sm][3] = x1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
                    acc[j][sm] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], z[sm][j], acc[j][sm], 0, 0, 0);
            tw[0] = tw[2]; tw[1] = tw[3]; tw[2] = n2; tw[3] = n3;
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- slab: [split][xi][co][ci]; D[ci][co]: lane li = output channel, register quad = 4 ci ------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* sl = a.slabs + ((long long)(split * 16 + ti * 4 + j) * a.CoP) * a.CiP;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const int co = co0 + sm * 32 + li;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ci = ci0 + 8 * qd + 4 * kh;
                f32x4 vv;
#pragma unroll
                for (int k = 0; k < 4; ++k) vv[k] = acc[j][sm][4 * qd + k];
                f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
                if (a.accum) vv += *dst;
                *dst = vv;
            }
        }
    }
    if (a.bslabs != nullptr && blockIdx.y == 0) {
        // fixed order (no LDS atomics): xor-shuffle tree over the wave's threads that share gq, then the four waves
        // in sequence; the operand tiles at the start of the LDS are dead here (every wave is past its last read)
        float* sred = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = refid_wave_rows_sum<G4>(bsum[k]);
            if ((tid & 63) < G4) sred[(tid >> 6) * COT + gq * 4 + k] = v;
        }
        __syncthreads();
        if (tid < COT) {
            const float tot = ((sred[tid] + sred[COT + tid]) + sred[2 * COT + tid]) + sred[3 * COT + tid];
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + tid;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}

#ifdef REFID_EXPERIMENTAL_TILES
// ------------------------------------------------------------------------------------------------------------------
// The same transform-domain GEMMs on the bf16 matrix cores, six exact-split bf16 products per fp32 product
// (refid_wgrad_desc.algo = 3; the operand split and the product list of conv_wino6.hip):
//     dU_xi[i][o] += sum_tiles V_xi[tile][i] * Z_xi[tile][o],   v = vh + vm + vl,  z = zh + zm + zl  (bf16 each, exact)
// v_mfma_f32_32x32x16_bf16 takes K = 16 TILES per instruction and wants, per lane, eight consecutive K values of one
// channel: lane (li = channel, kh) owns the tile columns 8kh .. 8kh+7 of one tile row.  A K tile is therefore ONE tile row
// (2 output rows x 32 columns = 16 Winograd tiles): 6 bf16 MFMAs (192 matrix-pipe cycles) per accumulator and K tile
// where the fp32 tile needs 8 x 64 = 512.  BOTH operands are transformed and split on the fly -- about 19 VALU per MFMA:
// the kernel is VALU-bound, so it is built for VALU throughput: a 32(o) x 32(i) channel tile per workgroup (a wave =
// one transform row, 4 accumulators) keeps a wave under 168 registers, three workgroups = three waves per SIMD.  The raw
// tiles (gradient 2x32 px x 32 o, input halo 4x34 px x 32 i, fp32 NHWC as in memory) arrive by LDS-DMA
// (buffer_load ... lds: no staging registers, no ds_write pass, hardware zero fill outside the image), double buffered,
// one barrier per K tile; fragments are gathered with conflict-free ds_read_b32 (lane = channel).
//
// MEASURED (tools/bench_wgrad6.py, B=8, 8 grouped steps; profiles/r03_wgrad6_bench.txt): correct (same tests as the fp32
// tile, 1e-5 relative to it) and SLOWER -- 0.72-0.81x the fp32 tile at two waves per SIMD (198 registers), 0.40-0.50x at
// three (168 registers, spills in the loop): ~450 VALU + 36 LDS instructions per 24 MFMAs make it VALU-issue bound at
// ~9 cycles per VALU instruction and SIMD, far from the matrix pipe's 768 cycles.  The 64(o) x 32(i) form (two sub-tiles
// per wave, 14 VALU per MFMA) does not fit 256 registers.  Only in libraries built with REFID_EXPERIMENTAL_TILES=1.
constexpr int TH6 = 2, COT6 = 32, GQ6 = COT6 / 4;
constexpr int G6_F4 = TH6 * TW * GQ6;                     // 512 float4: [pixel][32 o]
constexpr int HP6 = (TH6 + 2) * HWD;                      // 136 halo pixels
constexpr int X6_F4 = HP6 * X4;                           // 1088 float4: [pixel][32 i]
constexpr int BUF6_F4 = G6_F4 + X6_F4;
constexpr int LDS6_BYTES = 2 * BUF6_F4 * 16;              // 51,200: three workgroups per CU
constexpr int G6_PIECES = G6_F4 / 64, X6_PIECES = X6_F4 / 64;     // 8 / 17 one-KB pieces
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr6;

__device__ __forceinline__ void split8f(const float (&v)[8], f32x4 (&pl)[3]) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const __bf16 h = (__bf16)v[k];
        p0[k] = h;
        const float r = v[k] - (float)h;
        const __bf16 m = (__bf16)r;
        p1[k] = m;
        p2[k] = (__bf16)(r - (float)m);
    }
    pl[0] = __builtin_bit_cast(f32x4, p0);
    pl[1] = __builtin_bit_cast(f32x4, p1);
    pl[2] = __builtin_bit_cast(f32x4, p2);
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void wgrad_wino6_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row owned by this wave
    const int co0 = blockIdx.z * COT6, ci0 = blockIdx.y * CIT;
    const int split = blockIdx.x;

    const int ra = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rb = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const float ca = (ti == 3) ? 0.f : 1.f;
    const float cb = (ti == 0) ? 0.f : ((ti == 1) ? 1.f : -1.f);

    const bool xFromA = ci0 < a.Ca || ci0 >= a.Ctot;       // workgroup-uniform source (host: c_a % 32 == 0 for two sources)
    const int xld = xFromA ? a.ldA : a.ldB;
    const long long gpixAll = (long long)a.N * a.Ho * a.Wo, xpixAll = (long long)a.N * a.H * a.W;
    const int limG = (int)min(gpixAll * a.ldG * 4, 0x7fffffffLL), limX = (int)min(xpixAll * xld * 4, 0x7fffffffLL);
    const int ntAll = a.ntiles * a.groups;

    // DMA pieces of this wave: a piece = 64 lanes x 16 bytes = 1 KB of the LDS image, which IS the memory layout
    // (8 pixels x 32 channels for both tensors); lane -> (pixel, channel quad)
    constexpr int GPW = G6_PIECES / 4;                     // 2 gradient pieces per wave
    constexpr int XPW = (X6_PIECES + 3) / 4;               // 5 input pieces per wave (the 20 cover 17: the surplus repeats the last)
    auto dma_tile = [&](int pt, int buf) {
        // every per-lane offset is recomputed per tile from an opaque copy of the lane id: hoisted out of the K loop they
        // would sit in (spilled) registers all along; a few dozen VALU per tile instead
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int cq = ln & 7, prow = ln >> 3;
        const int gco = co0 + cq * 4, xc = ci0 + cq * 4;
        const bool gcok = gco < a.Co, xcok = xc < a.Ctot;
        const int xcc = xFromA ? xc : xc - a.Ca;
        const int grp = pt / a.ntiles;                     // workgroup-uniform: the time step this tile belongs to
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[grp]), 0, limG, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xFromA ? a.inA[grp] : a.inB[grp]), 0, limX, 0x00020000);
        int t = pt - grp * a.ntiles;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        const int n = t / a.tilesY;
        const int oy0 = ty * TH6, ox0 = tx * TW;
        const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
        char* gdst = smem + (buf * BUF6_F4 + wave * GPW * 64) * 16;
        char* xdst = smem + (buf * BUF6_F4 + G6_F4) * 16;
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int p = (wave * GPW + k) * 8 + prow;
            const int oy = oy0 + p / TW, ox = ox0 + p % TW;
            // out of the image / channel range: the offset is forced out of range with an OR (a select becomes a branch)
            const int bad = -(int)!(gcok && oy < a.Ho && ox < a.Wo);
            const int vo = ((((n * a.Ho + oy) * a.Wo + ox) * a.ldG + gco) * 4) | bad;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr6)(gdst + k * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < XPW; ++k) {
            const int piece = min(wave * XPW + k, X6_PIECES - 1);
            const int hp = piece * 8 + prow;
            const int iy = iy0 + hp / HWD, ix = ix0 + hp % HWD;
            const int bad = -(int)!(xcok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
            const int vo = ((((n * a.H + iy) * a.W + ix) * xld + xcc) * 4) | bad;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr6)(xdst + piece * 1024), 16, vo, 0, 0, 0);
        }
    };

    f32x16 acc[4];                                         // [j]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bs = 0.f;                                        // bias partial of channel li: this wave's quarter of the tile columns
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};              // products kept: (V plane, Z plane), largest first
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};

    int pt = split, it = 0;
    if (pt < ntAll) dma_tile(pt, 0);
    for (; pt < ntAll; pt += a.nsplit, ++it) {
        const int cur = it & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's DMA pieces of buffer `cur` have landed
        __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody is done with the other buffer
        __builtin_amdgcn_sched_barrier(0);
        if (pt + a.nsplit < ntAll) dma_tile(pt + a.nsplit, cur ^ 1);
        const float* sG = reinterpret_cast<const float*>(smem) + cur * BUF6_F4 * 4;
        const float* sX = sG + G6_F4 * 4;
        // ---- V side: row transform of this lane's 18 window columns (tile columns 8kh .. 8kh+7 of input channel li) ----
        const float* xA = sX + (ra * HWD + 16 * kh) * CIT + li;
        const float* xB = sX + (rb * HWD + 16 * kh) * CIT + li;
        float tw[18];
#pragma unroll
        for (int c = 0; c < 18; ++c) tw[c] = xA[c * CIT] + sgn * xB[c * CIT];
        // ---- Z side: X_b = ca dY[0][b] + cb dY[1][b] of the 8 tiles of output channel li ----
        __builtin_amdgcn_sched_barrier(0);
        float x0[8], x1[8];
        {
            const float* g0 = sG + (16 * kh) * COT6 + li;
            const float* g1 = g0 + TW * COT6;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float a0 = g0[(2 * k) * COT6], a1 = g0[(2 * k + 1) * COT6];
                const float b0 = g1[(2 * k) * COT6], b1 = g1[(2 * k + 1) * COT6];
                x0[k] = ca * a0 + cb * b0;
                x1[k] = ca * a1 + cb * b1;
                // bias: every gradient value is counted by exactly one wave (tile columns k with k % 4 == this wave)
                if ((k & 3) == ti) bs += (a0 + a1) + (b0 + b1);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);             // one column at a time (register budget: 3 waves per SIMD);
                                                           // the other waves of the SIMD fill the matrix pipe meanwhile
            float v[8], z[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = (j == 0) ? tw[2 * k] - tw[2 * k + 2] : (j == 1) ? tw[2 * k + 1] + tw[2 * k + 2]
                     : (j == 2) ? tw[2 * k + 2] - tw[2 * k + 1] : tw[2 * k + 3] - tw[2 * k + 1];
                z[k] = (j == 0) ? x0[k] : (j == 1) ? x0[k] + x1[k] : (j == 2) ? x0[k] - x1[k] : x1[k];
            }
            f32x4 pv[3], pz[3];
            split8f(v, pv);
            split8f(z, pz);
#pragma unroll
            for (int e = 0; e < 6; ++e)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, pv[TA[e]]), __builtin_bit_cast(bf16x8, pz[TB[e]]), acc[j], 0, 0, 0);
        }
    }

    // ---- slab: [split][xi][co][ci]; D[ci][co]: lane li = output channel, register quad = 4 ci (as the fp32 tile) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* sl = a.slabs + ((long long)(split * 16 + ti * 4 + j) * a.CoP) * a.CiP;
        const int co = co0 + li;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ci = ci0 + 8 * qd + 4 * kh;
            f32x4 vv;
#pragma unroll
            for (int k = 0; k < 4; ++k) vv[k] = acc[j][4 * qd + k];
            f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
            if (a.accum) vv += *dst;
            *dst = vv;
        }
    }
    if (a.bslabs != nullptr && blockIdx.y == 0) {
        // fixed order: the two column halves (kh) by one shuffle, then the four waves in sequence through LDS
        float* sred = reinterpret_cast<float*>(smem);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                                   // every wave is past its last tile read; no DMA in flight
        const float v = bs + __shfl_xor(bs, 32, 64);
        if (kh == 0) sred[wave * COT6 + li] = v;
        __syncthreads();
        if (tid < COT6) {
            const float tot = ((sred[tid] + sred[COT6 + tid]) + sred[2 * COT6 + tid]) + sred[3 * COT6 + tid];
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + tid;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}

#else
constexpr int TH6 = 2, COT6 = 32;              // (geometry of the experimental six-product kernel: workspace sizing only)
#endif

struct WrArgs {
    const float* slabs; const float* bslabs; float* dw; float* db;
    int nsplit, Co, Ci, CoP, CiP, iBase, iTotal, perGroup;
};

// slab reduction + inverse weight transform: dg = G^T dU G, accumulated into OIHW (9 contiguous floats).  Deterministic:
// `perGroup` = LPE (power of two <= 16) adjacent lanes share one (co, ci) element, lane `sub` adds slabs sub, sub + LPE, ...
// in order, a fixed xor-shuffle tree combines them, lane 0 owns the gradient element (no atomics).
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const WrArgs a) {
    const long long plane = (long long)a.CoP * a.CiP;
    const long long slabStride = 16 * plane;
    const int lpe = a.perGroup;
    const long long gid = blockIdx.x * 256ll + threadIdx.x;
    const long long e = gid / lpe;                                 // (co, ci), ci fastest
    const int sub = (int)(gid % lpe);
    {
        const bool live = e < (long long)a.Co * a.Ci;
        const int ci = live ? (int)(e % a.Ci) : 0, co = live ? (int)(e / a.Ci) : 0;
        const float* p = a.slabs + (long long)co * a.CiP + ci;
        float u[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) u[x] = 0.f;
        if (live) {
            for (int k = sub; k < a.nsplit; k += lpe) {
#pragma unroll
                for (int x = 0; x < 16; ++x) u[x] += p[k * slabStride + x * plane];
            }
        }
        for (int o = 1; o < lpe; o <<= 1) {
#pragma unroll
            for (int x = 0; x < 16; ++x) u[x] += __shfl_xor(u[x], o, 64);
        }
        // t[a][j] = sum_i G[i][a] u[i][j] ;  dg[a][b] = sum_j t[a][j] G[j][b]
        float dg[9];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = 0.5f * (u[4 + j] + u[8 + j]), d = 0.5f * (u[4 + j] - u[8 + j]);
                t[j] = (aa == 0) ? u[j] + m : ((aa == 1) ? d : m + u[12 + j]);
            }
            const float m = 0.5f * (t[1] + t[2]), d = 0.5f * (t[1] - t[2]);
            dg[aa * 3 + 0] = t[0] + m;
            dg[aa * 3 + 1] = d;
            dg[aa * 3 + 2] = m + t[3];
        }
        if (live && sub == 0) {
            float* dst = a.dw + ((long long)co * a.iTotal + a.iBase + ci) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] += dg[k];
        }
    }
    const int s0 = 0, s1 = a.nsplit;
    if (a.db != nullptr && blockIdx.x == 0) {
        for (int co = threadIdx.x; co < a.Co; co += 256) {
            float s = 0.f;
            for (int k = s0; k < s1; ++k) s += a.bslabs[(long long)k * a.CoP + co];
            a.db[co] += s;
        }
    }
}

struct Geo { int ncoT, nciT, tilesX, tilesY, ntiles, nsplit, CoP, CiP; };

Geo geo_of(const refid_wgrad_desc* d) {
    const int TH = d->algo == 3 ? TH6 : TH_FP32;           // the six-product kernel's K tile is one tile row
    Geo g;
    const int cot = d->algo == 3 ? COT6 : COT;
    g.ncoT = cdiv(d->c_o, cot);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;   // stable across steps
    g.nciT = cdiv(ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b, CIT);
    g.tilesX = cdiv(d->wo, TW);
    g.tilesY = cdiv(d->ho, TH);
    g.ntiles = g.tilesX * g.tilesY * d->n;
    int want = cdiv(d->algo == 3 ? 768 : 512, g.ncoT * g.nciT);
    if (want < 1) want = 1;
    if (want > g.ntiles) want = g.ntiles;
    g.nsplit = want;
    g.CoP = g.ncoT * cot;
    g.CiP = g.nciT * CIT;
    return g;
}

}  // namespace

size_t refid_wgrad_wino_workspace_bytes(const refid_wgrad_desc* d) {
    const Geo g = geo_of(d);
    return ((size_t)g.nsplit * 16 * g.CoP * g.CiP + (size_t)g.nsplit * g.CoP) * sizeof(float);
}

int refid_wgrad_wino_launch(const refid_wgrad_desc* d, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0}, attr_done6{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_wino_kernel, LDS_BYTES, "wgrad_wino")) return rc;
#ifdef REFID_EXPERIMENTAL_TILES
    static std::atomic<unsigned long long> attr_done62{0};
    if (int rc = refid_lds_attr_once(attr_done6, &wgrad_wino6_kernel<3>, LDS6_BYTES, "wgrad_wino6")) return rc;
    if (int rc = refid_lds_attr_once(attr_done62, &wgrad_wino6_kernel<2>, LDS6_BYTES, "wgrad_wino6")) return rc;
#else
    REFID_CHECK(d->algo != 3, "wgrad: algo 3 (Winograd, six bf16 products) is an experiment that measured slower than algo 1; "
                              "build with REFID_EXPERIMENTAL_TILES=1 to run it");
#endif
    const Geo g = geo_of(d);
    REFID_CHECK(d->c_b == 0 || d->c_a % CIT == 0, "wgrad (Winograd): c_a must be a multiple of %d for two sources", CIT);
    {
        const long long lim = 0x7fffffffLL;
        REFID_CHECK((long long)d->n * d->ho * d->wo * d->ld_g * 4 < lim && (long long)d->n * d->h * d->w * d->ld_a * 4 < lim &&
                        (d->c_b == 0 || (long long)d->n * d->h * d->w * d->ld_b * 4 < lim),
                    "wgrad (Winograd): tensor too large for 32-bit buffer offsets (use algo 0)");
    }
    WwArgs a;
    const int ngrp = d->groups > 1 ? d->groups : 1;
    REFID_CHECK(ngrp <= REFID_WGRAD_MAX_GROUPS, "wgrad: at most %d grouped time steps", REFID_WGRAD_MAX_GROUPS);
    for (int k = 0; k < REFID_WGRAD_MAX_GROUPS; ++k) {
        const bool on = k > 0 && k < ngrp;
        a.g[k] = k == 0 ? d->g : (on ? d->g_more[k - 1] : d->g);
        a.inA[k] = k == 0 ? d->in_a : (on ? d->in_a_more[k - 1] : d->in_a);
        a.inB[k] = k == 0 ? d->in_b : (on ? d->in_b_more[k - 1] : d->in_b);
        REFID_CHECK(a.g[k] && a.inA[k] && (d->c_b == 0 || a.inB[k]), "wgrad: null tensor pointer in group %d", k);
    }
    a.groups = ngrp;
    a.ldG = d->ld_g; a.Co = d->c_o;
    a.ldA = d->ld_a; a.ldB = d->ld_b;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.slabs = d->slabs;
    a.bslabs = d->db ? d->slabs + (size_t)g.nsplit * 16 * g.CoP * g.CiP : nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pad = d->pad;
    a.tilesX = g.tilesX; a.tilesY = g.tilesY; a.ntiles = g.ntiles; a.nsplit = g.nsplit;
    a.CoP = g.CoP; a.CiP = g.CiP;
    a.accum = (d->phase == 2);
    if (d->phase != 3) {
        if (d->algo == 3) {
            REFID_CHECK(d->ld_g % 4 == 0 && d->ld_a % 4 == 0 && (d->c_b == 0 || d->ld_b % 4 == 0) && d->c_o % 4 == 0 &&
                            (d->c_a + d->c_b) % 4 == 0,
                        "wgrad (Winograd, six products): pitches and channel counts must be multiples of 4");
#ifdef REFID_EXPERIMENTAL_TILES
            static const int wps = []() { const char* e = getenv("REFID_WGRAD6_WPS"); return e ? atoi(e) : 2; }();
            if (wps == 2) hipLaunchKernelGGL(wgrad_wino6_kernel<2>, dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), LDS6_BYTES, st, a);
            else hipLaunchKernelGGL(wgrad_wino6_kernel<3>, dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), LDS6_BYTES, st, a);
#endif
        } else {
            hipLaunchKernelGGL(wgrad_wino_kernel, dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), LDS_BYTES, st, a);
        }
        REFID_LAUNCH_CHECK("wgrad_wino");
    }
    if (d->phase == 1 || d->phase == 2) return 0;          // reduction deferred (phase 3)
    WrArgs r;
    r.slabs = a.slabs; r.bslabs = a.bslabs; r.dw = d->dw; r.db = d->db;
    r.nsplit = g.nsplit; r.Co = d->o_real;
    r.Ci = (d->phase == 0 && a.Ctot < d->i_total - d->i_base) ? a.Ctot : d->i_total - d->i_base;
    r.CoP = g.CoP; r.CiP = g.CiP; r.iBase = d->i_base; r.iTotal = d->i_total;
    const long long total = (long long)r.Co * r.Ci;
    int lpe = 1;                           // lanes per element (small weight tensors only)
    while (lpe < 16 && (long long)lpe * 2 * total <= 65536 && lpe * 2 <= g.nsplit) lpe *= 2;
    r.perGroup = lpe;
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((int)((total * lpe + 255) / 256)), dim3(256), 0, st, r);
    REFID_LAUNCH_CHECK("wgrad_wino_reduce");
    return 0;
}
