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
    const int TH = TH_FP32;
    Geo g;
    const int cot = COT;
    g.ncoT = cdiv(d->c_o, cot);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;   // stable across steps
    g.nciT = cdiv(ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b, CIT);
    g.tilesX = cdiv(d->wo, TW);
    g.tilesY = cdiv(d->ho, TH);
    g.ntiles = g.tilesX * g.tilesY * d->n;
    int want = cdiv(512, g.ncoT * g.nciT);
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
    static std::atomic<unsigned long long> attr_done{0}, attr_doneW{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_wino_kernel<1>, lds_bytes_ww(1), "wgrad_wino")) return rc;
    if (int rc = refid_lds_attr_once(attr_doneW, &wgrad_wino_kernel<2>, lds_bytes_ww(2), "wgrad_wino/8 waves")) return rc;
    REFID_CHECK(d->algo != 3 && d->algo != 4,
                "wgrad: algo 3 (Winograd, six bf16 products) and algo 4 (Winograd, LDS-DMA staging) were experiments that did not "
                "beat algo 1 (DESIGN.md section 7); they are no longer built");
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
        {
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
