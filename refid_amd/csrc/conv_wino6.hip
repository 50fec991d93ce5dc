// Winograd F(2x2, 3x3) convolution tile with the transform-domain GEMMs on the gfx950 bf16 matrix cores:
// every fp32 operand of  M_xi[cout][tile] += U_xi[cout][c] * V_xi[c][tile]  is the EXACT sum of three bf16 numbers
// (h = rne(v), m = rne(v - h), l = v - h - m: 8 + 8 + 8 significand bits), and six bf16 MFMAs
//     u*v = uh*vh + (uh*vm + um*vh) + (uh*vl + ul*vh + um*vm)   + O(2^-24 |uv|)
// reproduce the fp32 product to one rounding with the same fp32 accumulator.
//
// Why both at once ("Winograd x six products"): the fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles for
// K = 2) runs at 1/16 of the bf16 one (v_mfma_f32_32x32x16_bf16, 32 cycles for K = 16).  Per 16 input channels and
// wave the fp32 Winograd tile issues 64 fp32 MFMAs = 4096 matrix-pipe cycles; six bf16 products on the DIRECT conv
// (conv_split.hip) need 6 * 10/9 * 36/16 as many bf16 MFMAs as Winograd does and only tie it; six products in the
// WINOGRAD domain are 48 bf16 MFMAs = 1536 cycles -- 2.67x fewer matrix-pipe cycles than either, in the same error
// class (the transforms B^T d B and G g G^T stay exact +-1 / 0.5 arithmetic in fp32; only the products change).
// The price is VALU: V is split on the fly (44 VALU per 8 values, ~5 VALU per MFMA) -- tools/probes/wino6_loop.hip
// measured that two waves per SIMD hide it (0.52 of the bf16 peak with the split, 0.58 without, random data).
//
// THREE fp16 products (template parameter F16, refid_conv_desc.mfma_terms = 3; round 6): fp16 has 11 significand bits, so TWO
// planes h = rne16(v), l = rne16(v - h) carry 22 and  u*v = uh*vh + uh*vl + ul*vh + O(2^-22 |uv|)  takes three
// v_mfma_f32_32x32x16_f16 and two U planes -- half the matrix-pipe cycles and two thirds of the U bytes (the kernel's two
// measured costs: profiles/r05_wino6_ablation.txt, r06_wino6_f16_ceiling.txt) -- for a per-product error of ~2^-22 instead of
// ~2^-24, both below what the fp32 accumulation of a K >= 144 dot product adds.  The price is fp16's RANGE (normal numbers
// 2^-14 .. 2^16: the low plane of a value below 2^-3 is subnormal), so both operands travel scaled by exact powers of two:
//   * U by 2^eU per packing (max |U| 2^eU in [2^12, 2^15); refid_pack_conv_weights_wino3h, exponent in the packing's header);
//   * V per LANE = per Winograd tile and transform row, ONLINE along K: a lane's accumulators all belong to its own tile
//     (MFMA column), so its scale need only be constant along K -- the lane keeps a reference exponent E (of the largest
//     |B^T d| it has seen), scales a chunk's row-transformed values by 2^(7 - E) (the chunk's largest lands in [2^7, 2^8), V in
//     < 2^9), and when a later chunk exceeds 2^6 times the reference (V would pass 2^15) it multiplies its accumulators by
//     2^(E - E') and goes on with E' -- the flash-attention rescale, with powers of two: exact.  Values 2^10 below a chunk's
//     largest keep all 22 bits; smaller ones an ABSOLUTE error of 2^-32 of the largest, far below the 2^-22 of the large terms
//     they are added to.  The output transform undoes 2^(7 - E + eU) per lane with v_ldexp before the rows meet in LDS.
//
//   out = mask( post( pre(conv3x3(src) + bias) + res ) ),  src = in_a or [in_a | in_b]
// (same contract, epilogue, split-K form and XCD-aware work mapping as conv_wino.hip; serves the forward conv and,
// on the flipped/transposed weights, the input gradient.)
//
// Mapping (one workgroup = 256 threads = 4 waves, 2 workgroups per CU):
//   * workgroup tile = 4x32 output pixels = 32 Winograd tiles x 64 output channels; K walks input channels in
//     chunks of 16 (the bf16 MFMA's K): lane (li, kh) holds channels 8kh .. 8kh+7 of tile / output channel li.
//   * wave w owns transform ROW i = w (xi = 4i .. 4i+3) for both 32-channel column tiles: 8 accumulators.
//   * per chunk a wave reads its two rows of the raw 6x34 halo (16 ds_read_b128, conflict-free image: even / odd
//     pixel columns de-interleaved, row pitch 40 slots), forms t = row transform (32 VALU) and then, per column j:
//     v_j (8 VALU) -> three bf16 planes (44 VALU) -> 12 MFMAs against the U fragments of (xi = 4i+j, 3 planes, 2 tiles).
//   * U = G g G^T comes pre-split from the pack kernel ([chunk16][xi][plane][cout][16] bf16); a lane's fragment is ONE
//     16-byte buffer load, a wave reads 1 KB contiguous; fragments of column j+1 are requested before column j's MFMAs
//     (two register sets), never through LDS.
//   * the raw halo of chunk c+2 is requested at the top of chunk c (global -> VGPR -> LDS, double buffered).
#include "common.h"
#include "conv_args.h"
#include <algorithm>
#include <cstdlib>

#ifndef REFID_WINO6_ABLATE
#define REFID_WINO6_ABLATE 0             // tools/probes/wino6_ablate.py only (see the loaders below)
#endif

namespace {

constexpr int TW = 32;                  // output pixels per workgroup row
constexpr int KC = 16;                  // input channels per chunk = K of one bf16 MFMA
constexpr int HWD = TW + 2;             // halo pixels per row
constexpr int TH = 4;                   // (output channels per workgroup: 32 * NT, NT = 32-channel column tiles per wave)
constexpr int ROWP = 40;                // LDS slots per halo row: even columns at 0..16, odd columns at 20..36
constexpr int PLANE = 6 * ROWP + 5;     // slots per channel-quad plane (245: plane pitch 980 dwords = 20 mod 32)
constexpr int R_F4 = 4 * PLANE;         // one raw halo buffer: [quad][row][slot] float4
constexpr int HP = (TH + 2) * HWD;      // 204 halo pixels
constexpr int R_ITEMS = (4 * HP + 255) / 256;
// 2 raw buffers (31 KB) in the K loop; afterwards the row exchange: 66 KB for 64 output channels, 33 KB for 32
constexpr int lds_bytes(int nt) { return nt == 2 ? 64 * 66 * 16 : (32 * 66 * 16 > 2 * R_F4 * 16 ? 32 * 66 * 16 : 2 * R_F4 * 16); }
constexpr int OOB = -1;                 // voffset 0xFFFFFFFF: buffer loads return 0 (hardware range check)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int W3H_HEADER = 64;          // bytes in front of the fp16 planes: int eU (runtime.hip)
constexpr int F16_TGT = 7;              // a freshly scaled chunk's largest |B^T d| lands in [2^7, 2^8)
constexpr int F16_SLACK = 6;            // binades the largest value may grow over the reference before the accumulators are rescaled
constexpr int F16_E0 = F16_TGT + 1;     // initial (biased) reference exponent: any real data is larger

// v (8 fp32 channels) -> three bf16 planes that sum to v exactly
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[3]) {
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

// v (8 fp32 channels, already scaled into fp16's range) -> two fp16 planes, h + l = v to 22 bits (20 VALU)
__device__ __forceinline__ void split8h(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[3]) {
    f16x8 h, l;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 ab = {k < 4 ? v0[k] : v1[k - 4], k < 4 ? v0[k + 1] : v1[k - 3]};
        const f16x2 hh = __builtin_convertvector(ab, f16x2);                  // v_cvt_pk_f16_f32 (RNE)
        const f32x2 r = ab - __builtin_convertvector(hh, f32x2);              // exact
        const f16x2 ll = __builtin_convertvector(r, f16x2);
        h[k] = hh[0]; h[k + 1] = hh[1];
        l[k] = ll[0]; l[k + 1] = ll[1];
    }
    pl[0] = __builtin_bit_cast(f32x4, h);
    pl[1] = __builtin_bit_cast(f32x4, l);
}

// NT = 2: 64 output channels per workgroup (8 accumulators per wave, two workgroups per CU).
// NT = 1: the 32-output-channel layers (decoder 2's trunk, the input gradient of level 0's first conv; round 4 -- they ran
//         on the fp32 Winograd tile at 0.36 of the fp32 roof): 4 accumulators per wave, three workgroups per CU; every V
//         split feeds 6 instead of 12 MFMAs, so this form is vector-issue bound -- and still well ahead of 64 fp32 MFMAs.
// F16: the three-fp16-product form (two planes per operand, online power-of-two scaling: see the top of the file)
template <int NT, bool F16>
__global__ __launch_bounds__(256, NT == 2 ? 2 : 3) void conv_wino6_kernel(const ConvKArgs a) {
    constexpr int BN = 32 * NT;
    constexpr int NPL = F16 ? 2 : 3;                        // planes per operand
    constexpr int NPR = (REFID_WINO6_ABLATE == 14 || REFID_WINO6_ABLATE == 15) ? 3 : (F16 ? 3 : 6);   // products kept
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sR = reinterpret_cast<f32x4*>(smem);            // two raw halo buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row i owned by this wave (xi = 4i .. 4i+3)
    if (REFID_WINO6_ABLATE == 20 && a.N > 0) return;       // (every workgroup leaves at once, registers / LDS as the product: dispatch only)
    if (REFID_WINO6_ABLATE == 21 && a.N > 0) { __syncthreads(); if (tid == 12345) a.out[0] = 1.f; return; }   // + one barrier

#if defined(REFID_WINO6_ABLATE) && (REFID_WINO6_ABLATE == 9 || REFID_WINO6_ABLATE == 11)
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if ((hw & 1) && blockIdx.x < 512 * 2) {          // first generation only: later workgroups inherit the phase
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < (REFID_WINO6_ABLATE == 9 ? 400 : 800)) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
    // XCD-aware work mapping (as conv_wino.hip): the channel tiles of one pixel tile are consecutive on one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int bt = (slot / a.ncot) * 8 + xcd;                    // pixel-tile index
    if (bt >= a.tilesX * a.tilesY * a.N) return;
    const int n0 = (slot % a.ncot) * BN;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int n = bt / a.tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // ---- loaders -----------------------------------------------------------------------------------------------
    const int limA = (int)min((long long)a.N * a.H * a.W * a.ldA * 4, 0x7fffffffLL);
    const int limB = a.inB ? (int)min((long long)a.N * a.H * a.W * a.ldB * 4, 0x7fffffffLL) : 0;
    const int uPlane = a.CoutPad * KC * 2;                 // bytes between two planes of one xi
    const int uXi = NPL * uPlane;                          // bytes between xi and xi+1
    const int uChunk = 16 * uXi;                           // bytes per K chunk
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w) + (F16 ? W3H_HEADER / 4 : 0), 0, (int)min((long long)a.nchunks * uChunk, 0x7fffffffLL), 0x00020000);
    const int eU = F16 ? *reinterpret_cast<const int*>(a.w) : 0;      // U travels as U 2^eU (wave-uniform scalar load)
    int eRef = F16_E0;                                      // this lane's reference exponent (biased), F16 only
    // raw halo: thread -> (pixel, channel quad); four lanes read one pixel's 64 contiguous bytes
    const int q = tid & 3;
    int pixo[R_ITEMS], sdst[R_ITEMS];
#pragma unroll
    for (int it = 0; it < R_ITEMS; ++it) {
        const int hp = (tid >> 2) + it * 64;
        const int row = hp / HWD, col = hp % HWD;
        const int iy = oy0 - a.pad + row, ix = ox0 - a.pad + col;
        const bool ok = hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pixo[it] = ok ? (n * a.H + iy) * a.W + ix : OOB;                       // pixel index; bytes = pixo * ld * 4 + q * 16
        // (threads past the halo store their zeros into the plane's padding slots: no branch around the store)
        sdst[it] = hp < HP ? q * PLANE + row * ROWP + (col >> 1) + (col & 1) * 20 : q * PLANE + 6 * ROWP + (tid & 3);
    }
    // U fragments of this lane: rows (cout) n0 + nt*32 + li, channels 8kh .. 8kh+7
    int voU[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int urow = a.coBase + n0 + nt * 32 + li;
        voU[nt] = (urow < a.CoutPad) ? (urow * KC + kh * 8) * 2 + ti * 4 * uXi : OOB;
        if (REFID_WINO6_ABLATE == 12) voU[nt] = ti * 4 * uXi;
    }
    // split-K (small grids only): this workgroup reduces chunks [kc0, kc1) and writes a raw partial output
    const int kper = (a.nchunks + a.ksplit - 1) / a.ksplit;
    const int kc0 = blockIdx.y * kper, kc1 = min(a.nchunks, kc0 + kper);

    f32x4 rr[R_ITEMS];
    // chunks past kc1 are requested with out-of-range offsets (zeros, no memory traffic): no branches around loads,
    // so the compiler's vmcnt bookkeeping stays exact
    // tools/probes/wino6_ablate.py builds this file with -DREFID_WINO6_ABLATE=n (one piece of the kernel removed, results
    // wrong) to price the pieces: 1 = U fragments always from chunk 0 (cache resident), 2 = no three-plane split,
    // 3 = no K loop, 4 = no residual / mask loads and no stores, 5 = raw halo always from chunk 0, 6 = no U loads in the K loop,
    // 7 = no raw loads / LDS stores in the K loop, 8 = 6 + 7, 9 / 11 = the workgroup in the odd wave slot of its SIMD starts
    // 4 / 8 us late (anti-phased pair per CU), 10 = 3 + 4, 12 = every lane of a U load reads the same 16 bytes (same
    // instruction count, no bandwidth), 13 = half the U loads (column tile 1 reuses tile 0's fragments), 14 = three products on
    // two planes (the MFMA / U-byte count of a two-plane fp16 form), 15 = 14 without the split; fp16 form (WINO6_TERMS=3):
    // 16 = no max / rescale / scale, 17 = a one-instruction max, 18 = no split, 19 = the prologue's halo loads from a cache-resident
    // tile, 20 = every workgroup returns at once (dispatch cost of the grid), 21 = 20 + one barrier.  Never in the product.
    auto load_raw = [&](int ch, f32x4 (&dst)[R_ITEMS]) {
        const int c0 = (REFID_WINO6_ABLATE == 5 ? 0 : ch) * KC;   // chunk-uniform source: Ca % 16 == 0 for two sources
        const bool fromA = c0 < a.Ca;
        const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
        const int qmask = (ch < kc1 && c0 + q * 4 < a.Ctot) ? 0 : OOB;   // partial last chunk: upper quads are zeros
        const int ld4 = (fromA ? a.ldA : a.ldB) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(fromA ? a.inA : a.inB), 0, fromA ? limA : limB, 0x00020000);
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            // pixo = -1 (outside the image): 2^32 - ld4 + q*16 is beyond any buffer, no compare needed; quads past the last
            // channel are forced out of range with an OR (a select here becomes a branch around the load + vmcnt(0))
            const int vo = (pixo[it] * ld4 + q * 16) | qmask;
            dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
        }
    };
    auto store_raw = [&](int buf, const f32x4 (&src)[R_ITEMS]) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) sR[buf * R_F4 + sdst[it]] = src[it];
    };
    // fragments of column j of chunk ch: [plane][nt]
    auto load_u = [&](int ch, int j, f32x4 (&dst)[NPL][NT]) {
        // (the hardware range check covers the vector offset only: a chunk past the range must not travel as a scalar offset)
        const bool in = ch < kc1;
        const int so = (REFID_WINO6_ABLATE == 1 ? 0 : ch) * uChunk + j * uXi;
#pragma unroll
        for (int p = 0; p < ((REFID_WINO6_ABLATE == 14 || REFID_WINO6_ABLATE == 15) ? 2 : NPL); ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (REFID_WINO6_ABLATE == 13 && nt == 1) { dst[p][nt] = dst[p][0]; continue; }
                dst[p][nt] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, in ? voU[nt] : OOB, in ? so + p * uPlane : 0, 0));
            }
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    // B^T row i of the 4x4 input patch of this lane's tile: t = d[P] + sgn * d[M]
    //   i=0: d0 - d2   i=1: d1 + d2   i=2: d2 - d1   i=3: d1 - d3
    const int rowP = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rowM = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    // patch columns b = 0..3 of tile column k = li & 15 are pixels 2k+b: slots k, 20+k, k+1, 21+k of the row
    const int tbase = 2 * kh * PLANE + (2 * (li >> 4)) * ROWP + (li & 15);
    const int offP = tbase + rowP * ROWP, offM = tbase + rowM * ROWP;
    constexpr int BOFF[4] = {0, 20, 1, 21};

    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};              // products kept: (V plane, U plane), largest first
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};

    f32x4 uA[NPL][NT], uB[NPL][NT];
    auto phase = [&](int ch) {
        const int lc = ch - kc0;
        if (REFID_WINO6_ABLATE != 7 && REFID_WINO6_ABLATE != 8) {
            store_raw((lc + 1) & 1, rr);                   // raw(ch+1): requested one chunk ago
            load_raw(ch + 2, rr);
#ifdef REFID_WINO6_PIN
            __builtin_amdgcn_sched_barrier(0x38F);
#endif
        }
        const f32x4* r = sR + (lc & 1) * R_F4;
        f32x4 t[2][4];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int b = 0; b < 4; ++b) t[qq][b] = r[qq * PLANE + offP + BOFF[b]] + r[qq * PLANE + offM + BOFF[b]] * sgn;
        if constexpr (F16 && REFID_WINO6_ABLATE != 16) {
            // largest |t| of this lane's tile in this chunk (both K halves: lanes l and l ^ 32 hold the same tile)
            float m = 0.f;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(t[qq][b][0]), fabsf(t[qq][b][1]))), fmaxf(fabsf(t[qq][b][2]), fabsf(t[qq][b][3])));
            if (REFID_WINO6_ABLATE == 17) m = fabsf(t[0][0][0]);            // (a one-instruction "max": prices the max tree)
            const unsigned mb = __float_as_uint(m);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            const int eb = (int)(max(sw[0], sw[1]) >> 23);
            const bool grow = eb > eRef + F16_SLACK;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // rare: at a tile's first chunk with data, then only when the data grow 64x
                const int en = grow ? eb : eRef;
                const int fe = 127 + eRef - en;             // accumulators *= 2^(eRef - en)  (more than 2^-126: they no longer count)
                const float f = fe >= 1 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int rr2 = 0; rr2 < 16; ++rr2) acc[j][nt][rr2] *= f;
                eRef = en;
            }
            const float sc = __uint_as_float((unsigned)(F16_TGT + 254 - eRef) << 23);      // 2^(TGT - (eRef - 127)), eRef in [8, 255]
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int b = 0; b < 4; ++b) t[qq][b] *= sc;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 (&cur)[NPL][NT] = (j & 1) ? uB : uA;
            f32x4 (&nxt)[NPL][NT] = (j & 1) ? uA : uB;
            if (REFID_WINO6_ABLATE != 6 && REFID_WINO6_ABLATE != 8) load_u(j == 3 ? ch + 1 : ch, (j + 1) & 3, nxt);
#ifdef REFID_WINO6_PIN
            __builtin_amdgcn_sched_barrier(0x38F);         // vector-memory instructions stay where they are written
#endif
            f32x4 v[2], pl[3];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                v[qq] = (j == 0) ? t[qq][0] - t[qq][2] : (j == 1) ? t[qq][1] + t[qq][2]
                      : (j == 2) ? t[qq][2] - t[qq][1] : t[qq][1] - t[qq][3];
            if (REFID_WINO6_ABLATE == 2 || REFID_WINO6_ABLATE == 15) { pl[0] = v[0]; pl[1] = v[1]; pl[2] = t[0][j]; }
            else if (F16 && REFID_WINO6_ABLATE == 18) { pl[0] = v[0]; pl[1] = v[1]; }
            else if constexpr (F16) split8h(v[0], v[1], pl);
            else split8(v[0], v[1], pl);
#pragma unroll
            for (int e = 0; e < NPR; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {          // consecutive MFMAs hit different accumulators
                    if constexpr (F16)
                        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, cur[TB[e]][nt]), __builtin_bit_cast(f16x8, pl[TA[e]]), acc[j][nt], 0, 0, 0);
                    else
                        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, cur[TB[e]][nt]), __builtin_bit_cast(bf16x8, pl[TA[e]]), acc[j][nt], 0, 0, 0);
                }
        }
        __syncthreads();                                   // raw(ch) consumed by every wave; raw(ch+1) visible
    };

    // prologue: raw(kc0) -> LDS; raw(kc0+1) and U(kc0, column 0) in flight
    {
        f32x4 rr0[R_ITEMS];
#if REFID_WINO6_ABLATE == 19
        int keep[R_ITEMS];                                 // the prologue's two halo chunks from ONE cache-resident tile (tile 0 of
#pragma unroll                                             // sample 0): prices the HBM latency a workgroup's start waits for
        for (int it = 0; it < R_ITEMS; ++it) {
            keep[it] = pixo[it];
            const int hp = (tid >> 2) + it * 64;
            const int row = hp / HWD, col = hp % HWD;
            pixo[it] = (hp < HP && row >= a.pad && col >= a.pad) ? (row - a.pad) * a.W + (col - a.pad) : OOB;
        }
#endif
        load_raw(kc0, rr0);
        load_raw(kc0 + 1, rr);
#if REFID_WINO6_ABLATE == 19
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) pixo[it] = keep[it];
#endif
        load_u(kc0, 0, uA);
        if (REFID_WINO6_ABLATE == 6 || REFID_WINO6_ABLATE == 8) load_u(kc0, 1, uB);
        store_raw(0, rr0);
    }
    __syncthreads();

    if (REFID_WINO6_ABLATE != 3 && REFID_WINO6_ABLATE != 10)
        for (int ch = kc0; ch < kc1; ++ch) phase(ch);

    // ---- output transform (as conv_wino.hip) -----------------------------------------------------------------
    // lane: tile li, channels (r&3)+8(r>>2)+4kh of a 32-channel tile;  acc[j][t] = M[i][j]
    //   R_i[b] = sum_j M[i][j] A[j][b] :  b=0: M0+M1+M2   b=1: M1-M2-M3          (in registers)
    //   Y[a][b] = sum_i A^T[a][i] R_i[b]:  a=0: R0+R1+R2   a=1: R1-R2-R3          (across the 4 waves, through LDS)
    constexpr int XL = 66;
    f32x4* xch = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 r0, r1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * rq + k;
                r0[k] = acc[0][t][r] + acc[1][t][r] + acc[2][t][r];
                r1[k] = acc[1][t][r] - acc[2][t][r] - acc[3][t][r];
                if constexpr (F16) {                        // undo U's and this lane's V scale: exact (v_ldexp_f32)
                    r0[k] = ldexpf(r0[k], eRef - 127 - F16_TGT - eU);
                    r1[k] = ldexpf(r1[k], eRef - 127 - F16_TGT - eU);
                }
            }
            xch[(((ti * 2 + 0) * NT + t) * 4 + rq) * XL + kh * 33 + li] = r0;
            xch[(((ti * 2 + 1) * NT + t) * 4 + rq) * XL + kh * 33 + li] = r1;
        }
    // ---- fused epilogue, COALESCED: thread -> (output pixel, channel quad) in memory order.  A pass of the 256 threads covers
    // PPT = 256 / (BN / 4) consecutive pixels of a row x all BN channels (16 pixels x 64 channels, or 32 x 32); item it of a
    // thread is pixel (row it / CPR, column PPT (it % CPR) + p0) and ALWAYS the thread's own channel quad: the addresses are
    // one 64-bit base per tensor and thread plus workgroup-uniform steps (a generic (n, oy, ox) -> offset product per item
    // was 160 quarter-rate integer multiplies per tile -- as much vector-issue time as the K loop at 64 input channels).
    constexpr int QUADS = BN / 4, PPT = 256 / QUADS, CPR = TW / PPT;        // 16 / 16 / 2   or   8 / 32 / 1
    constexpr int NIT = TH * CPR;                                           // 8 or 4
    static_assert(TW == 32 && (NT == 1 || NT == 2) && PPT * CPR == TW, "epilogue item mapping");
    constexpr int TI_STRIDE = 2 * NT * 4 * XL;             // exchange slots between transform rows i and i + 1
    const int p0 = tid / QUADS, c = (tid % QUADS) * 4;
    const int j0 = n0 + c;
    const int nt = c >> 5, rq = (c & 31) >> 3, ckh = (c & 7) >> 2;
    const long long opb = (long long)(n * a.Ho + oy0) * a.Wo + ox0 + p0;
    const bool jok = j0 < a.Cout;
    const bool vec = a.vecOK && (j0 + 3 < a.Cout);
    bool cok[CPR];
#pragma unroll
    for (int q2 = 0; q2 < CPR; ++q2) cok[q2] = ox0 + q2 * PPT + p0 < a.Wo;
    float* const outB = a.out + (a.ksplit > 1 ? blockIdx.y * a.wsStride : 0) + opb * a.ldO + j0;
    const float* const resB = a.res ? a.res + opb * a.ldR + j0 : nullptr;
    const float* const mskB = a.mask ? a.mask + opb * a.ldM + j0 : nullptr;
    const int stepO[2] = {a.Wo * a.ldO, PPT * a.ldO}, stepR[2] = {a.Wo * a.ldR, PPT * a.ldR}, stepM[2] = {a.Wo * a.ldM, PPT * a.ldM};
    // second output out2 = out + add2 (a skip sum leaves with the tile): same item addressing as the other epilogue tensors
    const bool two = a.out2 != nullptr && a.ksplit == 1;
    float* const out2B = two ? a.out2 + opb * a.ldO2 + j0 : nullptr;
    const float* const add2B = two ? a.add2 + opb * a.ldA2 + j0 : nullptr;
    const int stepO2[2] = {a.Wo * a.ldO2, PPT * a.ldO2}, stepA2[2] = {a.Wo * a.ldA2, PPT * a.ldA2};
    const bool pre = REFID_WINO6_ABLATE != 4 && REFID_WINO6_ABLATE != 10 && vec && a.ksplit == 1 && (a.res != nullptr || a.mask != nullptr);
    f32x4 pres[NIT], pmask[NIT];
    if (pre) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool ok = (oy0 + it / CPR < a.Ho) && cok[it % CPR];
            pres[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            pmask[it] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (ok && a.res) pres[it] = *reinterpret_cast<const f32x4*>(resB + (it / CPR) * stepR[0] + (it % CPR) * stepR[1]);
            if (ok && a.mask) pmask[it] = *reinterpret_cast<const f32x4*>(mskB + (it / CPR) * stepM[0] + (it % CPR) * stepM[1]);
        }
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};                        // the same four output channels for all of a thread's items
    if (a.bias && a.ksplit == 1 && jok) {
        const float* bp = a.bias + a.coBase + j0;
        if (vec) bv = *reinterpret_cast<const f32x4*>(bp);
        else
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j0 + k < a.Cout) bv[k] = bp[k];
    }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it / CPR, colb = it % CPR;
        const int col = colb * PPT + p0;
        const int oa = row & 1, tile = ((row & 3) >> 1) * 16 + (col >> 1), ob = col & 1;
        if (oy0 + row >= a.Ho || !cok[colb] || !jok) continue;
        const f32x4* xp = xch + (((oa * 2 + ob) * NT + nt) * 4 + rq) * XL + ckh * 33 + tile;   // row i0 = oa
        const float sg = oa ? -1.f : 1.f;                   // a=0: R0+R1+R2 ; a=1: R1-R2-R3
        f32x4 v = xp[0] + (xp[TI_STRIDE] + xp[2 * TI_STRIDE]) * sg;
        float* const outp = outB + row * stepO[0] + colb * stepO[1];
        if (a.ksplit > 1) {                                 // raw partial sums; the finishing pass applies the epilogue
            *reinterpret_cast<f32x4*>(outp) = v;
            continue;
        }
        v += bv;
        lrelu4(v, a.slopePre, a.slopePre != 1.f);
        if (REFID_WINO6_ABLATE == 4 || REFID_WINO6_ABLATE == 10) {
            if (v[0] == 12345.678f) outp[0] = v[1];
            continue;
        }
        if (vec) {
            if (a.res) v += pre ? pres[it] : *reinterpret_cast<const f32x4*>(resB + row * stepR[0] + colb * stepR[1]);
            lrelu4(v, a.slopePost, a.slopePost != 1.f);
            if (a.mask) {
                const f32x4 mv = pre ? pmask[it] : *reinterpret_cast<const f32x4*>(mskB + row * stepM[0] + colb * stepM[1]);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
            }
            *reinterpret_cast<f32x4*>(outp) = v;
            if (two)
                *reinterpret_cast<f32x4*>(out2B + row * stepO2[0] + colb * stepO2[1]) =
                    v + *reinterpret_cast<const f32x4*>(add2B + row * stepA2[0] + colb * stepA2[1]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= a.Cout) break;
                float tv = v[k];
                if (a.res) tv += resB[row * stepR[0] + colb * stepR[1] + k];
                tv = lrelu(tv, a.slopePost);
                if (a.mask) tv *= (mskB[row * stepM[0] + colb * stepM[1] + k] > 0.f) ? 1.f : a.slopeMask;
                outp[k] = tv;
                if (two) out2B[row * stepO2[0] + colb * stepO2[1] + k] = tv + add2B[row * stepA2[0] + colb * stepA2[1] + k];
            }
        }
    }
}

struct Wino6Plan { int ks; int nt; dim3 grid; };

// same small-grid policy as the fp32 tile (conv_wino.hip::wino_plan), in chunks of 16 channels
Wino6Plan wino6_plan(ConvKArgs& a, int split_mode, int tile_hint = 0) {
    Wino6Plan p;
    // (tile_hint 4, an experiment switch: the 32-channel form for every layer -- three instead of two workgroups per CU,
    //  twice the V transforms / splits per product)
    p.nt = (a.Cout > 32 && tile_hint != 4) ? 2 : 1;
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, TH);
    // Small grids (round 5): where the 64-channel form would launch at most two workgroups per CU's worth (<= 512) and K is
    // short (< 9 chunks = up to 128 input channels), the 32-channel form doubles the workgroups (three per CU) and mostly makes
    // the split-K finishing launch unnecessary: B=1 103.2 -> 98.4 ms, B=2 143.0 -> 139.8, neutral at B=8 (410.1 vs 409.5; without
    // the K limit its bottleneck convs -- 256 channels at 32 x 32 -- took the slower form: 413 ms).  REFID_WINO_TILE=1 keeps the
    // 64-channel form.  Decided like split-K: by the total grid under policy 2, by the per-sample geometry (as if 8 samples)
    // under policy 1.
    if (p.nt == 2 && tile_hint == 0 && split_mode) {
        const int ncot2 = cdiv(a.Cout, 64);
        const int nwg2 = (split_mode == 2) ? round_up(a.tilesX * a.tilesY * a.N, 8) * ncot2 : a.tilesX * a.tilesY * ncot2 * 8;
        static const int small_wg = []() { const char* e = getenv("REFID_WINO6_SMALL"); return e ? atoi(e) : 512; }();
        static const int small_k = []() { const char* e = getenv("REFID_WINO6_SMALL_K"); return e ? atoi(e) : 9; }();
        if (nwg2 <= small_wg && cdiv(a.Ctot, KC) < small_k) p.nt = 1;
    }
    const int bn = 32 * p.nt;
    a.nchunks = cdiv(a.Ctot, KC);
    a.ncot = cdiv(a.Cout, bn);
    p.grid = dim3(round_up(a.tilesX * a.tilesY * a.N, 8) * a.ncot);
    const int nwg = (split_mode == 2) ? (int)p.grid.x : a.tilesX * a.tilesY * a.ncot * 8;   // "sample": as if N = 8
    int ks = 1;
    if (split_mode && nwg <= 256 && a.nchunks >= 8) {
        ks = 512 / nwg;
        if (ks > a.nchunks / 4) ks = a.nchunks / 4;
        if (ks > 8) ks = 8;
        if (ks < 1) ks = 1;
    }
    p.ks = ks;
    return p;
}

template <int NT, bool F16>
int launch_wino6_t(const ConvKArgs& a, dim3 grid, hipStream_t st, const char* what) {
    static std::atomic<unsigned long long> done{0};
    if (int rc = refid_lds_attr_once(done, &conv_wino6_kernel<NT, F16>, lds_bytes(NT), "conv_wino6")) return rc;
    hipLaunchKernelGGL((conv_wino6_kernel<NT, F16>), grid, dim3(256), lds_bytes(NT), st, a);
    REFID_LAUNCH_CHECK(what);
    return 0;
}

template <int NT>
int launch_wino6(const ConvKArgs& a, bool f16, dim3 grid, hipStream_t st, const char* what) {
    return f16 ? launch_wino6_t<NT, true>(a, grid, st, what) : launch_wino6_t<NT, false>(a, grid, st, what);
}

}  // namespace

bool refid_wino6_eligible(const ConvKArgs& a) {
    const long long lim = 0x7fffffffLL;
    // (thin outputs -- pred's 3 channels -- ride the 32-channel form: its MFMAs are cheap, the direct fp32 tile that padded them
    //  to 32 columns was bound by the fp32 matrix pipe at 100 us per launch)
    return a.Cout >= 1 && a.Ctot % 4 == 0 && (a.inB == nullptr || a.Ca % KC == 0) &&
           (long long)a.N * a.H * a.W * a.ldA * 4 < lim && (!a.inB || (long long)a.N * a.H * a.W * a.ldB * 4 < lim) &&
           (long long)cdiv(a.Ctot, KC) * 16 * 3 * a.CoutPad * KC * 2 < lim;
}

size_t refid_wino6_workspace_bytes(const ConvKArgs& ka, int split_mode) {
    size_t need = 0;
    for (int hint : {0, 1, 4}) {                           // any tile form may be asked for (wino_tile): the largest need
        ConvKArgs a = ka;
        const Wino6Plan p = wino6_plan(a, split_mode, hint);
        if (p.ks > 1) need = std::max(need, (size_t)p.ks * a.N * a.Ho * a.Wo * round_up(a.Cout, 4) * sizeof(float));
    }
    return need;
}

int refid_launch_wino6(const ConvKArgs& ka, float* ws, size_t ws_bytes, int split_mode, int tile_hint, int terms, hipStream_t st) {
    ConvKArgs a = ka;
    REFID_CHECK(terms == 0 || terms == 6 || terms == 3, "conv2d: algo 5 takes mfma_terms 0 / 6 (six bf16 products) or 3 (three fp16 "
                "products, w_packed from refid_pack_conv_weights_wino3h), got %d", terms);
    const bool f16 = terms == 3;
    REFID_CHECK(!f16 || (reinterpret_cast<uintptr_t>(a.w) & 15) == 0, "conv2d: the fp16 Winograd packing must be 16-byte aligned");
    REFID_CHECK(refid_wino6_eligible(a),
                "conv2d: the Winograd six-product tile needs input-channel counts that are multiples "
                "of 4 (two sources: c_a a multiple of 16) and tensors below 2 GiB");
    // (a wide tile -- 8x32 pixels, 8 waves, weight fragments shared through an LDS ring, same bits -- measured 0-10 % slower
    //  and was removed in round 6: DESIGN.md section 7)
    const Wino6Plan pl = wino6_plan(a, ws ? split_mode : 0, tile_hint);
    dim3 grid = pl.grid;
    const int ks = pl.ks;
    if (ks == 1) return pl.nt == 2 ? launch_wino6<2>(a, f16, grid, st, "conv_wino6") : launch_wino6<1>(a, f16, grid, st, "conv_wino6/32");
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const int ldW = round_up(a.Cout, 4);
    const size_t need = (size_t)ks * npix * ldW * sizeof(float);
    if (need > ws_bytes || (reinterpret_cast<uintptr_t>(ws) & 15)) {
        refid_set_error("conv_wino6: split-K workspace too small or misaligned (%zu bytes given, %zu needed: "
                        "refid_conv_workspace_bytes)", ws_bytes, need);
        return 1;
    }
    ConvKArgs p = a;                       // partial pass: raw sums into the workspace (the finishing pass writes out / out2)
    p.ksplit = ks; p.wsStride = npix * ldW; p.out = ws; p.ldO = ldW; p.out2 = nullptr;
    grid.y = ks;
    if (int rc = pl.nt == 2 ? launch_wino6<2>(p, f16, grid, st, "conv_wino6/splitk") : launch_wino6<1>(p, f16, grid, st, "conv_wino6/32/splitk"))
        return rc;
    ConvKArgs f = a;
    f.ksplit = ks; f.wsStride = npix * ldW;
    return refid_launch_splitk_finish(f, ws, ldW, npix, st);
}
