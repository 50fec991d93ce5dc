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
#include <type_traits>

// Timing experiments only (tools/probes/wgrad_wino_ablate.py builds the variants; results are wrong for n != 0):
//   product tile:   1: tiles are not re-staged (no ds_write pass, no second barrier)   2: no global tile loads
//                   3: neither, no barrier   4: 3 + operands from registers -- 3 and 4 are NOT bounds: with nothing left in
//                   the loop that writes LDS or synchronises, the compiler hoists every LDS read out of it
//   LDS-DMA tile (experimental builds, algo 4):   5: no DMA in the K loop   6: no wait / barrier
//                   11: every request dead (DMA issued, all offsets out of range: zero fill, no memory traffic)
//                   12: no DMA, no wait / barrier, scheduling fences kept (the LDS-fed loop by itself)
#ifndef REFID_WW_ABLATE
#define REFID_WW_ABLATE 0
#endif
#define WW_DMA_ON (REFID_WW_ABLATE == 0 || REFID_WW_ABLATE == 6 || REFID_WW_ABLATE == 11)
#define WW_BAR_ON (REFID_WW_ABLATE == 0 || REFID_WW_ABLATE == 5 || REFID_WW_ABLATE == 11)

namespace {

constexpr int TH = 4, TW = 32;                 // output pixels per K tile (2 x 16 Winograd tiles)
constexpr int TH_FP32 = TH;
constexpr int COT = 64, CIT = 32;
constexpr int PX = TH * TW;
constexpr int HWD = TW + 2, HP = (TH + 2) * HWD;
constexpr int G4 = COT / 4, X4 = CIT / 4;
constexpr int G_TOTAL = PX * G4, X_TOTAL = HP * X4;
constexpr int X_ITEMS = (X_TOTAL + 255) / 256;
constexpr int lds_bytes_ww(int iw) { return (G_TOTAL + iw * X_TOTAL) * 16 + COT * 4; }
constexpr int LDS_BYTES = lds_bytes_ww(1);

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

// IW = 32-channel input sub-tiles per workgroup.  IW = 1: 4 waves, two workgroups per CU (the product form).  IW = 2 (round 4
// experiment, REFID_WGRAD_WINO_IW=2): 8 waves, one workgroup per CU -- waves 0-3 and 4-7 are two copies of the tile above for two
// NEIGHBOURING input-channel tiles that share ONE staged gradient tile (the 64 x 32-pixel dY tile is fetched once per
// input-channel tile: 2 x at 64 input channels, 8 x at 256): same waves per SIMD, same per-wave loop, 28 % fewer staged bytes
// per MFMA at 64 channels -- and measured 0-4 % slower (see the launcher).  Each half stages its own input halo with its own
// (wave-uniform) source descriptor, so a two-source conv may change source between the halves.
template <int IW>
__global__ __launch_bounds__(256 * IW, IW == 1 ? 2 : 1) void wgrad_wino_kernel(const WwArgs a) {
    constexpr int NT = 256 * IW;
    constexpr int GI = G_TOTAL / NT;                       // gradient-tile items per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x;
    const int half = IW == 2 ? __builtin_amdgcn_readfirstlane(tid0 >> 8) : 0;     // wave-uniform
    f32x4* sG4 = reinterpret_cast<f32x4*>(smem);
    f32x4* sX4 = sG4 + G_TOTAL + half * X_TOTAL;
    float* sBias = reinterpret_cast<float*>(sG4 + G_TOTAL + IW * X_TOTAL);
    const float* sG = reinterpret_cast<const float*>(sG4);
    const float* sX = reinterpret_cast<const float*>(sX4);

    const int tid = tid0 & 255;                            // thread inside its half (input-halo staging, compute roles)
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row owned by this wave
    const int co0 = blockIdx.z * COT, ci0 = (blockIdx.y * IW + half) * CIT;
    const int split = blockIdx.x;


    const int gq = tid0 % G4, xq = tid % X4;
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
    f32x4 rg[GI], rx[X_ITEMS];

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
        for (int it = 0; it < GI; ++it) {
            const int p = tid0 / G4 + it * (NT / G4);
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
        for (int it = 0; it < GI; ++it) {
            const int p = tid0 / G4 + it * (NT / G4);
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

    // The K loop exists four times, one per transform row (wave-uniform switch): with the row's coefficients as compile-time
    // constants the Z transform costs 0-2 instead of 4 VALU per column tile and step, rows 0 and 3 read only the gradient
    // row they use (8 instead of 12 LDS reads per step), and row 3's minus sign moves into the V window (d3 - d1).
    //   B^T rows: i=0: d0-d2  i=1: d1+d2  i=2: d2-d1  i=3: d1-d3 ;  A rows: X_b = ca dY[0][b] + cb dY[1][b],
    //   (ca, cb) = (1,0) (1,1) (1,-1) (0,-1)
    auto kloop = [&](auto TI) {
        constexpr int I = decltype(TI)::value;
        constexpr int ra = (I == 0) ? 0 : ((I == 2) ? 2 : 1);
        constexpr int rb = (I == 0) ? 2 : ((I == 1) ? 2 : ((I == 2) ? 1 : 3));
        auto comb = [](float pa, float pb) { return I == 1 ? pa + pb : (I == 3 ? pb - pa : pa - pb); };   // window row pair
        int pt = split;
        if (pt < ntAll) {
            load_tile(pt);
            store_tile();
        }
        __syncthreads();

        for (; pt < ntAll; pt += a.nsplit) {
            const bool more = pt + a.nsplit < ntAll;
            if (more && REFID_WW_ABLATE != 2 && REFID_WW_ABLATE < 3) load_tile(pt + a.nsplit);

            // 16 K steps per K tile, fully unrolled (every LDS address is base + immediate).  A step pairs the two tile
            // ROWS of one tile column (MFMA K half kh = tile row), so consecutive steps of a lane walk along a row and
            // the 4-column input window slides by 2: only 2 new columns per row are read per step (12 instead of 16
            // LDS floats per 8 MFMAs -- LDS bytes and VALU instructions do not hide under MFMAs on gfx950).
            // The raw operands of step s+1 are fetched before the 8 MFMAs of step s issue.
            const float* xA = sX + ((2 * kh + ra) * HWD) * CIT + li;     // + (2*s + b) * CIT
            const float* xB = sX + ((2 * kh + rb) * HWD) * CIT + li;
            const float* gP = sG + (2 * kh * TW) * COT + li;             // + (2*s) * COT
            // T_b = d[ra][b] +- d[rb][b] for the 34 window columns of this lane's row pair, consumed two per step; the
            // step loop is fully unrolled, so tw[] / gbuf[] are plain registers (no copies between steps).
            float tw[4];
            float gbuf[2][2][4];
            auto fetch_g = [&](int s, float (&pg)[2][4]) {
                const int go = (2 * s) * COT;
#pragma unroll
                for (int sm = 0; sm < 2; ++sm) {
                    if (REFID_WW_ABLATE == 4) {
                        pg[sm][0] = pg[sm][1] = pg[sm][2] = pg[sm][3] = __builtin_bit_cast(float, go + sm + (int)threadIdx.x);
                        continue;
                    }
                    if (I != 3) { pg[sm][0] = gP[go + sm * 32]; pg[sm][1] = gP[go + sm * 32 + COT]; }
                    if (I != 0) { pg[sm][2] = gP[go + sm * 32 + TW * COT]; pg[sm][3] = gP[go + sm * 32 + TW * COT + COT]; }
                }
            };
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) tw[b4] = comb(xA[b4 * CIT], xB[b4 * CIT]);
            fetch_g(0, gbuf[0]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float n2 = 0.f, n3 = 0.f;
                if (s + 1 < 16) {
                    fetch_g(s + 1, gbuf[(s + 1) & 1]);
                    if (REFID_WW_ABLATE == 4) {
                        n2 = tw[0] - tw[1]; n3 = tw[1] - tw[0];
                    } else {
                        n2 = comb(xA[(2 * s + 4) * CIT], xB[(2 * s + 4) * CIT]);
                        n3 = comb(xA[(2 * s + 5) * CIT], xB[(2 * s + 5) * CIT]);
                    }
                }
                const float (&cg)[2][4] = gbuf[s & 1];
                // column transform; the 4th operand carries the sign of Z's 4th column (z3 = -x1), so Z needs no negation
                const float v[4] = {tw[0] - tw[2], tw[1] + tw[2], tw[2] - tw[1], tw[3] - tw[1]};
                // Z row I:  X_b = ca dY[0][b] + cb dY[1][b] (row 3: +dY[1][b], its sign sits in the window);  z = {x0, x0 + x1, x0 - x1, (-)x1}
                float z[2][4];
#pragma unroll
                for (int sm = 0; sm < 2; ++sm) {
                    const float x0 = I == 0 ? cg[sm][0] : (I == 1 ? cg[sm][0] + cg[sm][2] : (I == 2 ? cg[sm][0] - cg[sm][2] : cg[sm][2]));
                    const float x1 = I == 0 ? cg[sm][1] : (I == 1 ? cg[sm][1] + cg[sm][3] : (I == 2 ? cg[sm][1] - cg[sm][3] : cg[sm][3]));
                    z[sm][0] = x0; z[sm][1] = x0 + x1; z[sm][2] = x0 - x1; z[sm][3] = x1;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int sm = 0; sm < 2; ++sm)
                        acc[j][sm] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], z[sm][j], acc[j][sm], 0, 0, 0);
                tw[0] = tw[2]; tw[1] = tw[3]; tw[2] = n2; tw[3] = n3;
            }
            if (REFID_WW_ABLATE < 3) __syncthreads();
            if (more && REFID_WW_ABLATE != 1 && REFID_WW_ABLATE < 3) {
                store_tile();
                __syncthreads();
            }
        }
    };
    switch (__builtin_amdgcn_readfirstlane(ti)) {
        case 0: kloop(std::integral_constant<int, 0>{}); break;
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        default: kloop(std::integral_constant<int, 3>{}); break;
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
            if ((tid0 & 63) < G4) sred[(tid0 >> 6) * COT + gq * 4 + k] = v;
        }
        __syncthreads();
        if (tid0 < COT) {
            float tot = ((sred[tid0] + sred[COT + tid0]) + sred[2 * COT + tid0]) + sred[3 * COT + tid0];
            if (IW == 2) tot += ((sred[4 * COT + tid0] + sred[5 * COT + tid0]) + sred[6 * COT + tid0]) + sred[7 * COT + tid0];
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + tid0;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}

#ifdef REFID_EXPERIMENTAL_TILES
// ------------------------------------------------------------------------------------------------------------------
// The same tile with its operands staged by LDS-DMA into two buffers (refid_wgrad_desc.algo = 4, experimental builds).
//
// Question: the kernel above stages single-buffered (load -> registers -> barrier -> ds_write -> barrier per K tile);
// tools/probes/wgrad_wino_ablate.py gives 107 TF/s as it is, 125 without the ds_write pass and its second barrier, 116
// without the global loads.  (The "MFMA loop alone" variants of that probe read 144-149 TF/s but are NOT a bound: with no
// barrier or store left in the loop the compiler hoists every LDS read out of it.)  Does a double-buffered, DMA-fed
// loop with one barrier per K tile recover the difference?  Here
//   * a K tile is ONE tile row (2 output rows x 32 columns = 16 Winograd tiles; gradient 64 px x 64 o = 16 KB, input halo
//     4 rows x 40 px x 32 i = 20 KB, rows padded from 34 to 40 pixels so that a halo row is exactly five 1 KB pieces), so
//     that TWO buffers fit twice per CU (2 x 36 KB per workgroup);
//   * the tiles travel global -> LDS by `buffer_load_dwordx4 ... lds` (the LDS image IS the NHWC memory layout, 1 KB per
//     wave instruction, hardware zero fill for out-of-range offsets): no staging registers, no ds_write pass.  Wave w
//     moves gradient row w / 2 (its half) and halo row w: row validity is a scalar, only the column test is per lane;
//   * every workgroup walks a CONTIGUOUS range of K tiles (tile coordinates advance by scalar selects; the strided walk
//     of the kernel above needs five integer divisions per tile), in PAIRS so that every LDS address is base register +
//     immediate (bases in registers the compiler cannot see through: left alone it re-bases with a v_add per 1 KB);
//   * ONE barrier per K tile, placed before the tile's LAST step: behind it tile t+1 is known to have landed (its DMA
//     was issued a whole tile earlier) and every wave's reads of tile t are complete, so the DMA of tile t+2 into tile
//     t's buffer is issued there (cut into three parts that ride in three consecutive steps), and the last step fetches
//     the first operands of tile t+1 -- the matrix pipe never drains at a tile boundary;
//   * the bias gradient is the transform point (1, 1) of A dY A^T (= the sum of the 2x2 tile): wave 1's operand, summed
//     as it goes by.
// MEASURED (tools/bench_wgrad_wino.py, profiles/r03_wgrad_wino_dma.txt): correct (same tests), 110-114 TF/s against
// 106-108 at the 64x64 / 128x128 layers, equal (93 / 105) at 256x256; the train step does not move (477.3 vs 477.7 ms).
// With real LDS reads the loop reaches 120 TF/s without any memory traffic (requests issued but all out of range) and
// loses 8-11 % as soon as the tiles really stream in -- the barrier, the DMA issue and the address arithmetic are free by
// then.  I.e. the fp32 tile is limited by its LDS-fed inner loop (0.8 LDS + 3 VALU instructions per 64-cycle MFMA do not
// fully hide) and by the memory system's share of the chip's power, not by how the tiles are staged.  Not kept in the
// product library.  MFMA K half kh = left / right half of the tile row: a lane walks along a row, its window slides by 2.
constexpr int TH2 = 2;
constexpr int HW2 = 40;                                    // halo row pitch in pixels (34 used)
constexpr int G2_F4 = TH2 * TW * G4;                       // 1024 float4: [pixel][64 o]
constexpr int X2_F4 = (TH2 + 2) * HW2 * X4;                // 1280 float4: [row][40 px][32 i]
constexpr int BUF2_F4 = G2_F4 + X2_F4;
constexpr int LDS2_BYTES = 2 * BUF2_F4 * 16;               // 73,728: two workgroups per CU
typedef __attribute__((address_space(3))) void* lds_ptr2;

__global__ __launch_bounds__(256, 2) void wgrad_wino_dma_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row owned by this wave
    const int co0 = blockIdx.z * COT, ci0 = blockIdx.y * CIT;
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
    const int chunk = (ntAll + a.nsplit - 1) / a.nsplit;
    const int p0 = split * chunk, p1 = min(p0 + chunk, ntAll);

    // ---- DMA roles (per-lane constants) ----
    // gradient: wave w moves pixels 16w .. 16w+15 of the 2 x 32 tile = row w/2, columns 16(w&1) + 4k + (lane>>4); lane&15 = o quad
    // input:    wave w moves halo row w, columns 8k + (lane>>3) (k = 0..4; columns >= 34 are padding); lane&7 = i quad
    const int gdy = wave >> 1, gdx = (wave & 1) * 16 + (lane >> 4);
    const int gco = co0 + (lane & 15) * 4;
    const int gconst = ((gdy * a.Wo + gdx) * a.ldG + gco) * 4;
    const int gbadc = (gco < a.Co) ? 0 : -1;
    const int xdx = lane >> 3;
    const int xc = ci0 + (lane & 7) * 4;
    const int xcc = xFromA ? xc : xc - a.Ca;
    const int xconst = ((wave * a.W + xdx) * xld + xcc) * 4;
    const int xbadc = (xc < a.Ctot) ? 0 : -1;
    const int xbad4 = (xdx >= 2) ? -1 : xbadc;             // piece 4: columns 32 + xdx, only 32 and 33 exist

    // coordinates of the next tile to request (uniform; advanced by increments)
    int qg, qn, qy, qx;
    {
        int t = p0 < ntAll ? p0 : 0;
        qg = t / a.ntiles; t -= qg * a.ntiles;
        qx = t % a.tilesX; t /= a.tilesX;
        qy = t % a.tilesY; qn = t / a.tilesY;
    }
    // tiles are processed in PAIRS (buffer 0, buffer 1) so that every LDS address of the loop body is base + immediate:
    // with the buffer chosen at run time the address arithmetic alone took the tile from 146 to 113 TFLOP/s
    // (tools/probes/wgrad_wino_ablate.py, variants 7 / 9) -- VALU instructions do not hide under fp32 MFMAs here.
    // An odd range is padded with a dead tile (all offsets out of range: the DMA fills its buffer with zeros).
    const int npairs = (p1 - p0 + 1) / 2;
    const int p1e = p0 + 2 * npairs;
    int issued = p0;
    // tensors of the time step the next requested tile belongs to (reloaded from the argument block only when the walk
    // crosses into the next step: an s_load + its wait inside the request sequence stalls the LDS counter as well)
    const float* gcur = a.g[min(qg, a.groups - 1)];
    const float* xcur = xFromA ? a.inA[min(qg, a.groups - 1)] : a.inB[min(qg, a.groups - 1)];
    // A tile request = 9 DMA instructions per wave + ~90 scalar / vector instructions of address arithmetic.  Issued in one
    // piece behind the barrier it costs 10 % (the matrix pipe waits for it), so it is cut into three parts that ride in
    // three consecutive steps (the tile's last step and the first two of the next tile), straight-line code without
    // branches so that each part stays inside its step's scheduling region.
    auto issue_part = [&](auto BUF, auto PART) {
        constexpr int buf = decltype(BUF)::value, part = decltype(PART)::value;
        const int oy0 = qy * TH2, ox0 = qx * TW;
        const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
        // out of the image / channel range: the offset is forced out of range with an OR (a select becomes a branch);
        // tiles past the end of the range are dead (all out of range: zero fill)
        const int dead = (issued < p1 && REFID_WW_ABLATE != 11) ? 0 : -1;
        if constexpr (part < 2) {
            const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gcur), 0, limG, 0x00020000);
            char* gdst = smem + (buf * BUF2_F4 + wave * 256) * 16;
            const int gsb = ((qn * a.Ho + oy0) * a.Wo + ox0) * a.ldG * 4;
            const int grow = (oy0 + gdy < a.Ho) ? dead : -1;
#pragma unroll
            for (int k = (part == 0 ? 0 : 3); k < (part == 0 ? 3 : 4); ++k) {
                const int cbad = (ox0 + gdx + 4 * k < a.Wo) ? 0 : -1;
                const int vo = (gsb + k * 16 * a.ldG + gconst) | gbadc | grow | cbad;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr2)(gdst + k * 1024), 16, vo, 0, 0, 0);
            }
        }
        if constexpr (part > 0) {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xcur), 0, limX, 0x00020000);
            char* xdst = smem + (buf * BUF2_F4 + G2_F4 + wave * (HW2 * X4)) * 16;
            const int xsb = ((qn * a.H + iy0) * a.W + ix0) * xld * 4;
            const int xrow = ((unsigned)(iy0 + wave) < (unsigned)a.H) ? dead : -1;
#pragma unroll
            for (int k = (part == 1 ? 0 : 2); k < (part == 1 ? 2 : 5); ++k) {
                const int cbad = ((unsigned)(ix0 + xdx + 8 * k) < (unsigned)a.W) ? 0 : -1;
                const int vo = (xsb + k * 32 * xld + xconst) | (k == 4 ? xbad4 : xbadc) | xrow | cbad;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr2)(xdst + k * 1024), 16, vo, 0, 0, 0);
            }
        }
        if constexpr (part == 2) {
            ++issued;
            // advance (scalar selects, no branches)
            qx += 1;
            const int wx = (qx == a.tilesX) ? 1 : 0;
            qx = wx ? 0 : qx;
            qy += wx;
            const int wy = (qy == a.tilesY) ? 1 : 0;
            qy = wy ? 0 : qy;
            qn += wy;
            const int wn = (qn == a.N) ? 1 : 0;
            qn = wn ? 0 : qn;
            qg += wn;
            // (unconditional: a branch here would end the scheduling region)
            const int qgc = min(qg, a.groups - 1);
            gcur = a.g[qgc];
            xcur = xFromA ? a.inA[qgc] : a.inB[qgc];
        }
    };
    auto issue = [&](auto BUF) {
        issue_part(BUF, std::integral_constant<int, 0>{});
        issue_part(BUF, std::integral_constant<int, 1>{});
        issue_part(BUF, std::integral_constant<int, 2>{});
    };

    f32x16 acc[4][2];                                      // [j][o sub-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][sm][r] = 0.f;
    // bias: the transform point (1, 1) of A dY A^T IS the sum of the 2x2 gradient tile, so wave 1's operand z[sm][1] is
    // summed as it goes by (2 VALU per step in every wave -- only wave 1's sum is used; a per-wave branch would end the
    // scheduling region, separate reads of the gradient tile cost 8 % of the kernel)
    float bs[2] = {0.f, 0.f};

    // lane (li, kh): tile columns 8kh .. 8kh+7 of the tile row, one per step; window columns 16kh + 2s .. + 3.
    // LDS addresses: every read is base register + immediate.  ds_read2_b32 reaches 1 KB, ds_read2st64_b32 pairs addresses
    // that are multiples of 256 B apart over 64 KB; left to itself the compiler re-bases with a v_add_u32 per 1 KB window
    // (40 per tile pair: 8 % of the kernel -- VALU instructions do not hide under fp32 MFMAs here).  So the bases live in
    // registers the compiler cannot see through: one per 32-channel half for the gradient tile (all its offsets are
    // multiples of 256 B, both buffers), one per (buffer, halo row, 8-column window) for the input tile.
    typedef __attribute__((address_space(3))) const float lds_cf;      // (typed LDS pointers: 32-bit, ds_read instructions)
    lds_cf* lds = (lds_cf*)smem;
    lds_cf* gS[2];
    lds_cf* xAw[2][3];
    lds_cf* xBw[2][3];
#pragma unroll
    for (int sm = 0; sm < 2; ++sm) {
        gS[sm] = lds + (16 * kh) * COT + sm * 32 + li;
        asm volatile("" : "+v"(gS[sm]));
    }
#pragma unroll
    for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            xAw[bf][w] = lds + bf * (BUF2_F4 * 4) + G2_F4 * 4 + (ra * HW2 + 16 * kh + 8 * w) * CIT + li;
            xBw[bf][w] = lds + bf * (BUF2_F4 * 4) + G2_F4 * 4 + (rb * HW2 + 16 * kh + 8 * w) * CIT + li;
            asm volatile("" : "+v"(xAw[bf][w]), "+v"(xBw[bf][w]));
        }
    float tw[4];
    float gbuf[2][2][4];
    auto fetch_g = [&](int bf, int s, float (&pg)[2][4]) {
        const int go = bf * (BUF2_F4 * 4) + (2 * s) * COT;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            pg[sm][0] = gS[sm][go];            pg[sm][1] = gS[sm][go + COT];
            pg[sm][2] = gS[sm][go + TW * COT]; pg[sm][3] = gS[sm][go + TW * COT + COT];
        }
    };
    auto xval = [&](int bf, int c) { return xAw[bf][c >> 3][(c & 7) * CIT] + sgn * xBw[bf][c >> 3][(c & 7) * CIT]; };
    // one K tile out of buffer CUR
    auto tile = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
            if (s + 1 < 8) {
                // parts two and three of the request the previous tile's last step began (into ITS buffer, cur ^ 1)
                if (WW_DMA_ON) {
                    if (s == 0) issue_part(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, 1>{});
                    if (s == 1) issue_part(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, 2>{});
                }
                fetch_g(cur, s + 1, gbuf[(s + 1) & 1]);
                n2 = xval(cur, 2 * s + 4);
                n3 = xval(cur, 2 * s + 5);
            } else {
                // the tile's last step: behind this barrier the next tile has landed (buffer cur^1) and nobody reads buffer
                // cur any more (its last operands are in registers: lgkmcnt(0)), so the tile after it is requested into cur
                if (WW_BAR_ON) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                }
                if (REFID_WW_ABLATE == 12) { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); }
                // request the tile after the next into buffer cur, first part (past the range: dead tiles -- zero fill, never read)
                if (WW_DMA_ON) issue_part(CUR, std::integral_constant<int, 0>{});
                fetch_g(cur ^ 1, 0, gbuf[0]);              // first operands of the next tile (stale data after the last)
                n0 = xval(cur ^ 1, 0);
                n1 = xval(cur ^ 1, 1);
                n2 = xval(cur ^ 1, 2);
                n3 = xval(cur ^ 1, 3);
            }
            const float (&cg)[2][4] = gbuf[s & 1];
            const float v[4] = {tw[0] - tw[2], tw[1] + tw[2], tw[2] - tw[1], tw[3] - tw[1]};
            float z[2][4];
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
                const float x0 = ca * cg[sm][0] + cb * cg[sm][2];
                const float x1 = ca * cg[sm][1] + cb * cg[sm][3];
                z[sm][0] = x0; z[sm][1] = x0 + x1; z[sm][2] = x0 - x1; z[sm][3] = x1;
                bs[sm] += z[sm][1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
                    acc[j][sm] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], z[sm][j], acc[j][sm], 0, 0, 0);
            if (s + 1 < 8) { tw[0] = tw[2]; tw[1] = tw[3]; tw[2] = n2; tw[3] = n3; }
            else { tw[0] = n0; tw[1] = n1; tw[2] = n2; tw[3] = n3; }
        }
    };

    if (npairs > 0) {
        issue(std::integral_constant<int, 0>{});
        issue_part(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});    // the first tile's steps 0 and 1 finish it
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) tw[b4] = xval(0, b4);
        fetch_g(0, 0, gbuf[0]);
    }
    for (int i = 0; i < npairs; ++i) {
        tile(std::integral_constant<int, 0>{});
        tile(std::integral_constant<int, 1>{});
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
        // wave 1 holds the tile sums: the two row halves (kh) by one shuffle; no cross-wave reduction
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const float tot = bs[sm] + __shfl_xor(bs[sm], 32, 64);
            if (ti == 1 && kh == 0) {
                float* dst = a.bslabs + (long long)split * a.CoP + co0 + sm * 32 + li;
                *dst = a.accum ? *dst + tot : tot;
            }
        }
    }
}

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
constexpr int TH6 = 2, COT6 = 32, TH2 = 2;     // (geometry of the experimental kernels: workspace sizing only)
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
    const int TH = d->algo == 3 ? TH6 : (d->algo == 4 ? TH2 : TH_FP32);   // experimental kernels: K tile = one tile row
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
    static std::atomic<unsigned long long> attr_done{0}, attr_done6{0}, attr_doneW{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_wino_kernel<1>, lds_bytes_ww(1), "wgrad_wino")) return rc;
    if (int rc = refid_lds_attr_once(attr_doneW, &wgrad_wino_kernel<2>, lds_bytes_ww(2), "wgrad_wino/8 waves")) return rc;
#ifdef REFID_EXPERIMENTAL_TILES
    static std::atomic<unsigned long long> attr_done62{0}, attr_done2{0};
    if (int rc = refid_lds_attr_once(attr_done2, &wgrad_wino_dma_kernel, LDS2_BYTES, "wgrad_wino_dma")) return rc;
    if (int rc = refid_lds_attr_once(attr_done6, &wgrad_wino6_kernel<3>, LDS6_BYTES, "wgrad_wino6")) return rc;
    if (int rc = refid_lds_attr_once(attr_done62, &wgrad_wino6_kernel<2>, LDS6_BYTES, "wgrad_wino6")) return rc;
#else
    REFID_CHECK(d->algo != 3 && d->algo != 4,
                "wgrad: algo 3 (Winograd, six bf16 products) and algo 4 (Winograd, LDS-DMA staging) are experiments that did not "
                "beat algo 1; build with REFID_EXPERIMENTAL_TILES=1 to run them");
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
        } else if (d->algo == 4) {
            REFID_CHECK(d->ld_g % 4 == 0 && d->ld_a % 4 == 0 && (d->c_b == 0 || d->ld_b % 4 == 0),
                        "wgrad (Winograd, LDS-DMA): pitches must be multiples of 4 floats");
            for (int k = 0; k < ngrp; ++k)
                REFID_CHECK(((uintptr_t)a.g[k] | (uintptr_t)a.inA[k] | (uintptr_t)(d->c_b ? a.inB[k] : nullptr)) % 16 == 0,
                            "wgrad (Winograd, LDS-DMA): tensors must be 16-byte aligned (group %d)", k);
#ifdef REFID_EXPERIMENTAL_TILES
            hipLaunchKernelGGL(wgrad_wino_dma_kernel, dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), LDS2_BYTES, st, a);
#endif
        } else {
            // REFID_WGRAD_WINO_IW=2: two neighbouring input-channel tiles per workgroup (8 waves, the gradient tile staged once
            // for both) whenever the tile count is even.  Measured (tools/bench_wgrad_wino.py regs, round 4): 0-4 % SLOWER than
            // the 4-wave form on every config-2 shape (95.6 vs 95.3 TF/s at 64->64 @256^2, 105-106 vs 109-110 at the 128- and
            // 256-channel layers; train step 472-474 vs 471 ms) -- the staged bytes it saves were not what limits this tile,
            // the eight-wave barriers cost more.  Off; same weight-gradient bits, bias partials in a different fixed order.
            static const int iw_max = []() { const char* e = getenv("REFID_WGRAD_WINO_IW"); return e ? atoi(e) : 1; }();
            if (iw_max >= 2 && g.nciT % 2 == 0)
                hipLaunchKernelGGL(wgrad_wino_kernel<2>, dim3(g.nsplit, g.nciT / 2, g.ncoT), dim3(512), lds_bytes_ww(2), st, a);
            else
                hipLaunchKernelGGL(wgrad_wino_kernel<1>, dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), lds_bytes_ww(1), st, a);
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
