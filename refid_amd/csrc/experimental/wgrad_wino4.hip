// Winograd F(3x3, 4x4) weight gradient of the 3x3 / stride-1 convolutions on the fp32 matrix cores
// (refid_wgrad_desc.algo = 5; autograd's conv-backward weight gradient; layers
// recurrent_sub_modules.py:659-678,719-726,755-758 under twoImage_event_recurrent_model.py:303).
//
// The weight gradient of a 3x3 conv is itself a correlation with a SMALL output (3x3) and a large "filter" (the output
// gradient), so the minimal-filtering identity is applied with the roles swapped: per 4x4 tile of the output gradient dY and
// the 6x6 input window d it covers,
//
//     dW(3x3) = A^T [ sum_{tiles} (G dY G^T)_xi (x) (B^T d B)_xi ] A               xi = 0..35  (points 0, +-1, +-2, inf)
//
// i.e. 36 transform-domain GEMMs  dU_xi[i][o] += V_xi[tile][i] * Z_xi[tile][o]  over 4x4-pixel tiles: 36 fp32 MFMA-units per
// 16 output pixels = 2.25 per pixel, where F(2x2,3x3) (wgrad_wino.hip) needs 4 and the direct form 9.  The inverse
// transform A^T dU A (36 -> 9 values) is applied once per weight by the slab reduction.  G rows are scaled to integers
// ({1,0,0,0} {1,1,1,1} {1,-1,1,-1} {1,2,4,8} {1,-2,4,-8} {0,0,0,1}); the scale lives in A^T, so every transform coefficient in
// the K loop is +-1, 2, 4 or 5 and rides in an FMA.  Measured deviation from the float64 weight gradient: 4e-6 .. 8e-6 of
// the tensor's scale (F(2x2): 1.3e-6 .. 1.7e-6, direct fp32: 2e-6) -- tests/test_hip_conv.py::test_wgrad_f4_*.
//
// Mapping: workgroup = 384 threads = 6 waves; wave w owns transform ROW w (xi = 6w .. 6w+5) of a 32(o) x 32(i) channel tile:
// 6 accumulators = 96 registers, <= 168 in all, so two workgroups = THREE waves per SIMD share a CU.  A wave needs only ITS
// row of Z and V, so both transforms are computed on the fly in registers from the raw NHWC tiles in LDS (lane = channel:
// conflict-free ds_read_b32), specialised per row (a wave-uniform switch selects one of six K loops whose row coefficients
// are compile-time constants: rows 0 / 5 read one gradient row and three input rows, the others four and four).
// K tile = 2 x 4 Winograd tiles (8 x 16 output pixels; MFMA K half = tile row): a lane walks the four tile columns, its
// 6-column input window slides by 4, so a step row-transforms 4 new columns.  Per step and wave: 6 MFMAs (384 matrix-pipe
// cycles), 30-46 VALU, 16-32 LDS reads.
// Staging: the LDS image IS the memory layout ([pixel][32 channels]), so tiles travel global -> LDS by LDS-DMA
// (buffer_load ... lds: no staging registers, which is what lets the tile fit 168, no ds_write pass, hardware zero fill
// outside the image), two buffers of 38.5 KB, ONE barrier per K tile.  A halo row is 18 pixels = two 1 KB pieces + one
// 256-byte piece (4-byte DMA: 2 pixels x 32 channels); row validity is wave-uniform, only the column test is per lane.
// Every workgroup walks a CONTIGUOUS range of K tiles (coordinates advance by scalar increments).  The workgroups that
// share a K range (all (o, i) tiles of one split) are `nsplit` apart in the grid, a multiple of 8 wherever the grid allows,
// i.e. on ONE XCD, so a tile is fetched from HBM once and re-read from that XCD's L2.
// Split-K slabs [split][36][o][i] persist over the T recurrent steps exactly like wgrad_wino.hip's (phase 1 / 2 / 3); the
// bias gradient is the transform point (1, 1) of G dY G^T (rows {1,1,1,1}: the tile sum), summed by wave 1 as it goes by.
//
// STATUS (round 5): correct (tests/test_hip_conv.py with REFID_EXPERIMENTAL_TILES=1) and NO FASTER than the 2x2-tile form it was
// meant to replace -- 53-57 TF/s issued with scalar transforms, 57-61 packed (0.38 of the fp32 pipe; 2x2 tiles: 0.70 at 64/36 the
// MFMAs), 48-52 in this file's final form (the explicit software pipeline below, which spills a few registers at 168).  With
// nothing but its transforms and MFMAs (no DMA, no LDS reads, no barrier) it reaches 0.44: at ~7 vector instructions per MFMA
// the transforms of a 6x6 window cost more than the MFMAs they save (profiles/r05_wgrad_f4x4_ablation.txt, r05_wgrad_f4x4_pmc.txt).
// The product kernel is csrc/wgrad_wino24.hip (2x4 tiles: one F(3,2) direction keeps a 64x32 channel tile and a +-1 row pass).
#include "../common.h"
#include <cstdlib>
#include <type_traits>

// Timing experiments only (tools/probes/w4_ablate.py builds the variants; results are wrong for n != 0):
//   1: no DMA requests (stale tiles: no global traffic, no LDS writes)   2: no MFMAs (operands kept alive)
//   3: no LDS reads (operands from opaque registers)   4: no barrier   5: 1 + 4   6: 1 + 3 + 4 (transforms + MFMAs alone)
#ifndef REFID_W4_ABLATE
#define REFID_W4_ABLATE 0
#endif
#define W4_NO_DMA (REFID_W4_ABLATE == 1 || REFID_W4_ABLATE == 5 || REFID_W4_ABLATE == 6)
#define W4_NO_MFMA (REFID_W4_ABLATE == 2)
#define W4_NO_LDS (REFID_W4_ABLATE == 3 || REFID_W4_ABLATE == 6)
#define W4_NO_BAR (REFID_W4_ABLATE == 4 || REFID_W4_ABLATE == 5 || REFID_W4_ABLATE == 6)

namespace {

__device__ __forceinline__ float w4_fake(float seed) { asm volatile("" : "+v"(seed)); return seed; }

constexpr int OT = 32, IT = 32;                // channel tile (o x i)
constexpr int TC = 4;                          // Winograd tile columns per K tile (tile rows: 2 = the MFMA K halves)
constexpr int GH = 8, GW = 4 * TC;             // output-gradient pixels of a K tile
constexpr int XH = GH + 2, XW = GW + 2;        // input halo
constexpr int X_BYTES = XH * XW * IT * 4;      // 23,040
constexpr int G_BYTES = GH * GW * OT * 4;      // 16,384
constexpr int BUF_BYTES = X_BYTES + G_BYTES;   // 39,424
constexpr int LDS4_BYTES = 2 * BUF_BYTES;      // 78,848: two workgroups per CU
constexpr int NXI = 36;
typedef __attribute__((address_space(3))) void* lds_ptr4;

struct W4Args {
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cf4;     // typed LDS pointers: 32-bit, ds_read instructions

// The transforms of one wave (transform row I), PACKED over two consecutive tile columns (steps s, s+1): a ds_read2st64 of
// (column c, column c + 4) lands in a register pair, every transform instruction is a v_pk_* on such pairs (half the vector
// instructions -- they do not hide under fp32 MFMAs on gfx950), and the MFMAs of step s / s+1 take the low / high
// halves.  LDS bases: xb[k] -> X[4 kh][column k][li], gb[k] -> dY[4 kh][column k][li] (k = column mod 4), xe / xo = columns
// 0 / 1 again: every read is base + a multiple of 256 bytes (a row is 9 x 256 B resp. 8 x 256 B, four columns are 512 B),
// i.e. ds_read2st64_b32 with no address arithmetic.  The bases are opaque to the compiler (it would re-base with a v_add per
// 1 KB window), and one base per column residue makes the offset-NEIGHBOURS it pairs exactly (column c, column c + 4).
template <int I>
struct W4Row {
    static constexpr int XR = XW * IT, GR = GW * OT;
    static constexpr int NX = (I == 0 || I == 5) ? 3 : 4;          // window rows this B^T row reads
    static constexpr int X0 = I == 0 ? 0 : 1, XS = (I == 0 || I == 5) ? 2 : 1;
    static constexpr int NG = (I == 0 || I == 5) ? 1 : 4;          // tile rows this G row reads
    static constexpr int G0 = I == 5 ? 3 : 0;

    // raw reads of one pair iteration: window columns 4s+2 .. 4s+5 (and their partners four columns on)
    static __device__ __forceinline__ void x_issue(lds_cf4* const (&xb)[4], int s, f32x2 (&r)[4][NX]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * s + 2 + c;
            lds_cf4* p = xb[col & 3] + (col & ~3) * IT;
#pragma unroll
            for (int k = 0; k < NX; ++k)
                r[c][k] = W4_NO_LDS ? f32x2{w4_fake(1.f + c), w4_fake(2.f + k)} : f32x2{p[(X0 + k * XS) * XR], p[(X0 + k * XS) * XR + 4 * IT]};
        }
    }
    // the tile's first two window columns (read alone)
    static __device__ __forceinline__ void x_issue01(lds_cf4* xe, lds_cf4* xo, float (&r)[2][NX]) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            r[0][k] = W4_NO_LDS ? w4_fake(1.f + k) : xe[(X0 + k * XS) * XR];
            r[1][k] = W4_NO_LDS ? w4_fake(2.f + k) : xo[(X0 + k * XS) * XR];
        }
    }
    // row I of B^T over the window rows
    template <class T>
    static __device__ __forceinline__ T x_row(const T (&r)[NX]) {
        if constexpr (I == 0) return 4.f * r[0] + (-5.f * r[1] + r[2]);
        else if constexpr (I == 1) return (r[2] + r[3]) - 4.f * (r[0] + r[1]);
        else if constexpr (I == 2) return 4.f * (r[0] - r[1]) + (r[3] - r[2]);
        else if constexpr (I == 3) return 2.f * (r[2] - r[0]) + (r[3] - r[1]);
        else if constexpr (I == 4) return (r[3] - r[1]) - 2.f * (r[2] - r[0]);
        else return 4.f * r[0] + (-5.f * r[1] + r[2]);
    }
    static __device__ __forceinline__ void g_issue(lds_cf4* const (&gb)[4], int s, f32x2 (&r)[4][NG]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * s + c;
            lds_cf4* p = gb[col & 3] + (col & ~3) * OT;
#pragma unroll
            for (int k = 0; k < NG; ++k)
                r[c][k] = W4_NO_LDS ? f32x2{w4_fake(1.f + c), w4_fake(2.f + k)} : f32x2{p[(G0 + k) * GR], p[(G0 + k) * GR + 4 * OT]};
        }
    }
    // row I of G over the tile rows
    static __device__ __forceinline__ f32x2 g_row(const f32x2 (&r)[NG]) {
        if constexpr (I == 0 || I == 5) return r[0];
        else if constexpr (I == 1) return (r[0] + r[2]) + (r[1] + r[3]);
        else if constexpr (I == 2) return (r[0] + r[2]) - (r[1] + r[3]);
        else if constexpr (I == 3) return 2.f * (4.f * r[3] + r[1]) + (4.f * r[2] + r[0]);
        else return (4.f * r[2] + r[0]) - 2.f * (4.f * r[3] + r[1]);
    }
    // V = (row) B over the six window columns t[0..5]
    static __device__ __forceinline__ void v_cols(const f32x2 (&t)[6], f32x2 (&v)[6]) {
        v[0] = 4.f * t[0] + (-5.f * t[2] + t[4]);
        v[1] = (t[3] + t[4]) - 4.f * (t[1] + t[2]);
        v[2] = 4.f * (t[1] - t[2]) + (t[4] - t[3]);
        const f32x2 d42 = t[4] - t[2], d31 = t[3] - t[1];
        v[3] = 2.f * d31 + d42;
        v[4] = d42 - 2.f * d31;
        v[5] = 4.f * t[1] + (-5.f * t[3] + t[5]);
    }
    // Z = (row) G^T over the four tile columns
    static __device__ __forceinline__ void z_cols(const f32x2 (&x)[4], f32x2 (&z)[6]) {
        const f32x2 e = x[0] + x[2], o = x[1] + x[3], e4 = 4.f * x[2] + x[0], o4 = 4.f * x[3] + x[1];
        z[0] = x[0]; z[1] = e + o; z[2] = e - o; z[3] = 2.f * o4 + e4; z[4] = e4 - 2.f * o4; z[5] = x[3];
    }
};

__global__ __launch_bounds__(384, 3) void wgrad_wino4_kernel(const W4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int co0 = blockIdx.z * OT, ci0 = blockIdx.y * IT;
    const int split = blockIdx.x;

    // the input-channel tile lies in one source (host: c_a % 32 == 0 for two sources), so the descriptor is workgroup-uniform;
    // a tile beyond the sources (first recurrent step: no second source yet) keeps a valid descriptor, all lanes out of range
    const bool xFromA = ci0 < a.Ca || ci0 >= a.Ctot;
    const int xld = xFromA ? a.ldA : a.ldB;
    const long long gpixAll = (long long)a.N * a.Ho * a.Wo, xpixAll = (long long)a.N * a.H * a.W;
    const int limG = (int)min(gpixAll * a.ldG * 4, 0x7fffffffLL), limX = (int)min(xpixAll * xld * 4, 0x7fffffffLL);
    const int ntAll = a.ntiles * a.groups;
    const int chunk = (ntAll + a.nsplit - 1) / a.nsplit;
    const int p0 = min(split * chunk, ntAll), p1 = min(p0 + chunk, ntAll);

    // ---- DMA roles.  16-byte pieces: lane -> (pixel lane>>3 of 8, channel quad lane&7); 4-byte tail piece of a halo row:
    // lane -> (pixel 16 + lane>>5, channel lane&31).  Out-of-range channels are forced out of range with an OR mask.  The
    // per-lane constants are recomputed per request from an opaque copy of the lane id: hoisted out of the K loop they would
    // hold six of the 72 registers the loop has next to its accumulators.
    // coordinates of the next tile to request (wave-uniform; advanced by increments)
    int qg, qn, qy, qx;
    {
        int t = p0 < ntAll ? p0 : 0;
        qg = t / a.ntiles; t -= qg * a.ntiles;
        qx = t % a.tilesX; t /= a.tilesX;
        qy = t % a.tilesY; qn = t / a.tilesY;
    }
    auto request = [&](int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l8 = ln >> 3, l32 = ln >> 5;
        const int xq = ci0 + (ln & 7) * 4, xt = ci0 + (ln & 31), gq = co0 + (ln & 7) * 4;
        const int xlc = (l8 * xld + (xFromA ? xq : xq - a.Ca)) * 4;                // bytes from the piece's first pixel
        const int xlt = (l32 * xld + (xFromA ? xt : xt - a.Ca)) * 4;
        const int glc = (l8 * a.ldG + gq) * 4;
        const int xbadq = xq < a.Ctot ? 0 : -1, xbadt = xt < a.Ctot ? 0 : -1, gbadq = gq < a.Co ? 0 : -1;
        const int oy0 = qy * GH, ox0 = qx * GW;
        const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[qg]), 0, limG, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xFromA ? a.inA[qg] : a.inB[qg]), 0, limX, 0x00020000);
        char* xdst = smem + buf * BUF_BYTES;
        char* gdst = xdst + X_BYTES;
        const int cx0 = (unsigned)(ix0 + l8) < (unsigned)a.W ? 0 : -1;                   // column tests (per lane)
        const int cx1 = (unsigned)(ix0 + 8 + l8) < (unsigned)a.W ? 0 : -1;
        const int cxt = (unsigned)(ix0 + 16 + l32) < (unsigned)a.W ? 0 : -1;
        const int cg0 = ox0 + l8 < a.Wo ? 0 : -1, cg1 = ox0 + 8 + l8 < a.Wo ? 0 : -1;
        // halo rows: wave w moves rows w and w + 6 (waves 0-3)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = wave + 6 * k;
            if (r < XH && !W4_NO_DMA) {
                const int iy = iy0 + r;
                const int rbad = (unsigned)iy < (unsigned)a.H ? 0 : -1;
                const int base = ((qn * a.H + iy) * a.W + ix0) * xld * 4;                 // bytes, < 2^31 for live lanes (host check)
                char* dst = xdst + r * (XW * IT * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr4)dst, 16, (base + xlc) | xbadq | rbad | cx0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr4)(dst + 1024), 16,
                                                         (base + 8 * xld * 4 + xlc) | xbadq | rbad | cx1, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr4)(dst + 2048), 4,
                                                         (base + 16 * xld * 4 + xlt) | xbadt | rbad | cxt, 0, 0, 0);
            }
        }
        // gradient rows: waves 0-3 row w, wave 4 rows 4 and 5, wave 5 rows 6 and 7
        const int gr0 = wave < 4 ? wave : 2 * wave - 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if ((k == 0 || wave >= 4) && !W4_NO_DMA) {
                const int r = gr0 + k;
                const int oy = oy0 + r;
                const int rbad = oy < a.Ho ? 0 : -1;
                const int base = ((qn * a.Ho + oy) * a.Wo + ox0) * a.ldG * 4;
                char* dst = gdst + r * (GW * OT * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr4)dst, 16, (base + glc) | gbadq | rbad | cg0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr4)(dst + 1024), 16,
                                                         (base + 8 * a.ldG * 4 + glc) | gbadq | rbad | cg1, 0, 0, 0);
            }
        }
        // advance (scalar selects)
        qx += 1;
        const int wx = qx == a.tilesX ? 1 : 0;
        qx = wx ? 0 : qx;
        qy += wx;
        const int wy = qy == a.tilesY ? 1 : 0;
        qy = wy ? 0 : qy;
        qn += wy;
        const int wn = qn == a.N ? 1 : 0;
        qn = wn ? 0 : qn;
        qg = min(qg + wn, a.groups - 1);
    };

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bs = 0.f;

    // The K loop as an explicit software pipeline (the scheduling fences pin the order; the compiler only places the waits).
    // Per pair iteration (two tile columns, 12 MFMAs): gradient reads -> V column pass (from the window rows transformed one
    // iteration earlier) -> Z -> the NEXT iteration's input reads are issued -> 12 MFMAs (the LDS latency rides under them)
    // -> row pass of what has arrived.  A K tile is two iterations; its ONE barrier sits in front of the second
    // iteration's MFMAs: behind it the next tile has landed and nobody reads this tile's buffer any more (this wave's last
    // reads of it were consumed by the Z transform just before), so the tile after the next is requested into it and the
    // next tile's first input reads are issued from the other buffer.
    auto kloop = [&](auto TI) {
        constexpr int I = decltype(TI)::value;
        using R = W4Row<I>;
        if (p0 >= p1) return;
        lds_cf4* xb[4]; lds_cf4* gb[4]; lds_cf4* xe; lds_cf4* xo;
        auto bases_x = [&](int buf) {
            xe = (lds_cf4*)(smem + buf * BUF_BYTES) + (4 * kh) * (XW * IT) + li;
            xb[0] = xe; xb[1] = xe + IT; xb[2] = xe + 2 * IT; xb[3] = xe + 3 * IT; xo = xe + IT;
            asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xe), "+v"(xo));
        };
        auto bases_g = [&](int buf) {
            lds_cf4* ge = (lds_cf4*)(smem + buf * BUF_BYTES + X_BYTES) + (4 * kh) * (GW * OT) + li;
            gb[0] = ge; gb[1] = ge + OT; gb[2] = ge + 2 * OT; gb[3] = ge + 3 * OT;
            asm volatile("" : "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3]));
        };
        f32x2 t[6], v[6], z[6];
        f32x2 xr[4][R::NX], gr[4][R::NG];
        float x01[2][R::NX];
        auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
        // window rows of a tile's FIRST iteration out of what x_issue / x_issue01 fetched
        auto rows_first = [&] {
#pragma unroll
            for (int c = 0; c < 4; ++c) t[2 + c] = R::x_row(xr[c]);
            t[0] = f32x2{R::x_row(x01[0]), t[4][0]};
            t[1] = f32x2{R::x_row(x01[1]), t[5][0]};
        };
        // ... of its second iteration: columns 0 / 1 of the low half are the first iteration's columns 4 / 5 of the high half
        auto rows_second = [&] {
            const float c0 = t[4][1], c1 = t[5][1];
#pragma unroll
            for (int c = 0; c < 4; ++c) t[2 + c] = R::x_row(xr[c]);
            t[0] = f32x2{c0, t[4][0]};
            t[1] = f32x2{c1, t[5][0]};
        };
        auto zpass = [&] {
            f32x2 x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = R::g_row(gr[c]);
            R::z_cols(x, z);
            if constexpr (I == 1) bs += z[1][0] + z[1][1];                // point (1, 1) = sum of the 4x4 gradient tile
        };
        auto mfmas = [&] {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (W4_NO_MFMA) { asm volatile("" :: "v"(v[j][h]), "v"(z[j][h])); continue; }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][h], z[j][h], acc[j], 0, 0, 0);
                }
        };

        request(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the first tile has landed (this wave's pieces)
        if (!W4_NO_BAR) __builtin_amdgcn_s_barrier();      // ... everybody's
        fence();
        if (p0 + 1 < p1) request(1);
        bases_x(0);
        R::x_issue(xb, 0, xr);
        R::x_issue01(xe, xo, x01);
        rows_first();
        int cur = 0;
        for (int pt = p0; pt < p1; ++pt, cur ^= 1) {
            // ---- first iteration (tile columns 0, 1) ----
            fence();
            bases_g(cur);
            R::g_issue(gb, 0, gr);
            fence();
            R::v_cols(t, v);
            fence();
            zpass();
            fence();
            R::x_issue(xb, 2, xr);                         // second iteration's window columns, same buffer
            fence();
            mfmas();
            fence();
            rows_second();
            // ---- second iteration (tile columns 2, 3) ----
            fence();
            bases_g(cur);
            R::g_issue(gb, 2, gr);
            fence();
            R::v_cols(t, v);
            fence();
            zpass();
            fence();
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's pieces of the next tile have landed
            if (!W4_NO_BAR) __builtin_amdgcn_s_barrier();  // ... everybody's; and everybody is done with this tile's buffer
            fence();
            if (pt + 2 < p1) request(cur);
            bases_x(cur ^ 1);
            R::x_issue(xb, 0, xr);                         // (past the last tile: stale data, never used)
            fence();
            mfmas();
            fence();
            R::x_issue01(xe, xo, x01);                     // (the tile's first two columns: 8 more registers would spill above)
            rows_first();
        }
    };
    switch (wave) {
        case 0: kloop(std::integral_constant<int, 0>{}); break;
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        case 3: kloop(std::integral_constant<int, 3>{}); break;
        case 4: kloop(std::integral_constant<int, 4>{}); break;
        default: kloop(std::integral_constant<int, 5>{}); break;
    }

    // ---- slab: [split][xi][co][ci]; D[ci][co]: lane li = output channel, register quad = 4 ci ------
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float* sl = a.slabs + ((long long)(split * NXI + wave * 6 + j) * a.CoP) * a.CiP;
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
    if (a.bslabs != nullptr && blockIdx.y == 0 && wave == 1) {
        // wave 1 holds the tile sums of its 32 output channels: the two tile rows (kh) by one shuffle -- fixed order
        const float tot = bs + __shfl_xor(bs, 32, 64);
        if (kh == 0) {
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + li;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}

struct W4rArgs {
    const float* slabs; const float* bslabs; float* dw; float* db;
    int nsplit, Co, Ci, CoP, CiP, iBase, iTotal, perGroup;
};

// slab reduction + inverse transform dg = A^T dU A, accumulated into OIHW (9 contiguous floats).  Deterministic:
// `perGroup` = LPE (power of two <= 16) adjacent lanes share one (co, ci) element, lane `sub` adds slabs sub, sub + LPE, ...
// in order, a fixed xor-shuffle tree combines them, lane 0 owns the gradient element (no atomics).
//   A^T = [ 1/4 -1/6 -1/6 1/24  1/24 0 ]
//         [ 0   -1/6  1/6 1/12 -1/12 0 ]
//         [ 0   -1/6 -1/6 1/6   1/6  1 ]
__global__ __launch_bounds__(256) void wgrad_wino4_reduce_kernel(const W4rArgs a) {
    const long long plane = (long long)a.CoP * a.CiP;
    const long long slabStride = NXI * plane;
    const int lpe = a.perGroup;
    const long long gid = blockIdx.x * 256ll + threadIdx.x;
    const long long e = gid / lpe;                                 // (co, ci), ci fastest
    const int sub = (int)(gid % lpe);
    {
        const bool live = e < (long long)a.Co * a.Ci;
        const int ci = live ? (int)(e % a.Ci) : 0, co = live ? (int)(e / a.Ci) : 0;
        const float* p = a.slabs + (long long)co * a.CiP + ci;
        float u[NXI];
#pragma unroll
        for (int x = 0; x < NXI; ++x) u[x] = 0.f;
        if (live) {
            for (int k = sub; k < a.nsplit; k += lpe) {
#pragma unroll
                for (int x = 0; x < NXI; ++x) u[x] += p[k * slabStride + x * plane];
            }
        }
        for (int o = 1; o < lpe; o <<= 1) {
#pragma unroll
            for (int x = 0; x < NXI; ++x) u[x] += __shfl_xor(u[x], o, 64);
        }
        // t[p][j] = sum_i A^T[p][i] u[i][j] ;  dg[p][q] = sum_j t[p][j] A^T[q][j]
        constexpr float c4 = 0.25f, c6 = 1.f / 6.f, c12 = 1.f / 12.f, c24 = 1.f / 24.f;
        auto at3 = [&](float u0, float u1, float u2, float u3, float u4, float u5, float (&r)[3]) {
            const float s12 = u1 + u2, d21 = u2 - u1, s34 = u3 + u4, d34 = u3 - u4;
            r[0] = c4 * u0 - c6 * s12 + c24 * s34;
            r[1] = c6 * d21 + c12 * d34;
            r[2] = c6 * (s34 - s12) + u5;
        };
        float t[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float r[3];
            at3(u[j], u[6 + j], u[12 + j], u[18 + j], u[24 + j], u[30 + j], r);
            t[0][j] = r[0]; t[1][j] = r[1]; t[2][j] = r[2];
        }
        float dg[9];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            float r[3];
            at3(t[pp][0], t[pp][1], t[pp][2], t[pp][3], t[pp][4], t[pp][5], r);
            dg[pp * 3 + 0] = r[0]; dg[pp * 3 + 1] = r[1]; dg[pp * 3 + 2] = r[2];
        }
        if (live && sub == 0) {
            float* dst = a.dw + ((long long)co * a.iTotal + a.iBase + ci) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] += dg[k];
        }
    }
    if (a.db != nullptr && blockIdx.x == 0) {
        for (int co = threadIdx.x; co < a.Co; co += 256) {
            float s = 0.f;
            for (int k = 0; k < a.nsplit; ++k) s += a.bslabs[(long long)k * a.CoP + co];
            a.db[co] += s;
        }
    }
}

struct Geo4 { int ncoT, nciT, tilesX, tilesY, ntiles, nsplit, CoP, CiP; };

Geo4 geo4_of(const refid_wgrad_desc* d) {
    Geo4 g;
    g.ncoT = cdiv(d->c_o, OT);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;     // stable across steps
    g.nciT = cdiv(ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b, IT);
    g.tilesX = cdiv(d->wo, GW);
    g.tilesY = cdiv(d->ho, GH);
    g.ntiles = g.tilesX * g.tilesY * d->n;
    // two workgroups per CU; a multiple of 8 splits keeps the workgroups of one K range on one XCD (grid x is fastest)
    static const int wgs = []() { const char* e = getenv("REFID_W4_WGS"); return e ? atoi(e) : 512; }();
    int want = cdiv(wgs, g.ncoT * g.nciT);
    if (want >= 8) want = want / 8 * 8;
    if (want < 1) want = 1;
    if (want > g.ntiles) want = g.ntiles;
    g.nsplit = want;
    g.CoP = g.ncoT * OT;
    g.CiP = g.nciT * IT;
    return g;
}

}  // namespace

size_t refid_wgrad_wino4_workspace_bytes(const refid_wgrad_desc* d) {
    const Geo4 g = geo4_of(d);
    return ((size_t)g.nsplit * NXI * g.CoP * g.CiP + (size_t)g.nsplit * g.CoP) * sizeof(float);
}

int refid_wgrad_wino4_launch(const refid_wgrad_desc* d, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_wino4_kernel, LDS4_BYTES, "wgrad_wino4")) return rc;
    const Geo4 g = geo4_of(d);
    if (getenv("REFID_W4_OCC")) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wgrad_wino4_kernel, 384, LDS4_BYTES);
        fprintf(stderr, "wgrad_wino4: occupancy %d workgroups per CU (%s), grid %d x %d x %d\n", nb, hipGetErrorString(e), g.nsplit, g.nciT, g.ncoT);
    }
    REFID_CHECK(d->c_b == 0 || d->c_a % IT == 0, "wgrad (Winograd F(3x3,4x4)): c_a must be a multiple of %d for two sources", IT);
    REFID_CHECK(d->ld_g % 4 == 0 && d->ld_a % 4 == 0 && (d->c_b == 0 || d->ld_b % 4 == 0) && d->c_o % 4 == 0 &&
                    d->c_a % 4 == 0 && d->c_b % 4 == 0,
                "wgrad (Winograd F(3x3,4x4)): pitches and channel counts must be multiples of 4 floats (16-byte LDS-DMA pieces)");
    {
        const long long lim = 0x7fffffffLL;
        REFID_CHECK((long long)d->n * d->ho * d->wo * d->ld_g * 4 < lim && (long long)d->n * d->h * d->w * d->ld_a * 4 < lim &&
                        (d->c_b == 0 || (long long)d->n * d->h * d->w * d->ld_b * 4 < lim),
                    "wgrad (Winograd F(3x3,4x4)): tensor too large for 32-bit buffer offsets (use algo 0)");
    }
    W4Args a;
    const int ngrp = d->groups > 1 ? d->groups : 1;
    REFID_CHECK(ngrp <= REFID_WGRAD_MAX_GROUPS, "wgrad: at most %d grouped time steps", REFID_WGRAD_MAX_GROUPS);
    for (int k = 0; k < REFID_WGRAD_MAX_GROUPS; ++k) {
        const bool on = k > 0 && k < ngrp;
        a.g[k] = k == 0 ? d->g : (on ? d->g_more[k - 1] : d->g);
        a.inA[k] = k == 0 ? d->in_a : (on ? d->in_a_more[k - 1] : d->in_a);
        a.inB[k] = k == 0 ? d->in_b : (on ? d->in_b_more[k - 1] : d->in_b);
        REFID_CHECK(a.g[k] && a.inA[k] && (d->c_b == 0 || a.inB[k]), "wgrad: null tensor pointer in group %d", k);
        REFID_CHECK(((uintptr_t)a.g[k] | (uintptr_t)a.inA[k] | (uintptr_t)(d->c_b ? a.inB[k] : nullptr)) % 16 == 0,
                    "wgrad (Winograd F(3x3,4x4)): tensors must be 16-byte aligned (group %d)", k);
    }
    a.groups = ngrp;
    a.ldG = d->ld_g; a.Co = d->c_o;
    a.ldA = d->ld_a; a.ldB = d->ld_b;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.slabs = d->slabs;
    a.bslabs = d->db ? d->slabs + (size_t)g.nsplit * NXI * g.CoP * g.CiP : nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pad = d->pad;
    a.tilesX = g.tilesX; a.tilesY = g.tilesY; a.ntiles = g.ntiles; a.nsplit = g.nsplit;
    a.CoP = g.CoP; a.CiP = g.CiP;
    a.accum = (d->phase == 2);
    if (d->phase != 3) {
        hipLaunchKernelGGL(wgrad_wino4_kernel, dim3(g.nsplit, g.nciT, g.ncoT), dim3(384), LDS4_BYTES, st, a);
        REFID_LAUNCH_CHECK("wgrad_wino4");
    }
    if (d->phase == 1 || d->phase == 2) return 0;          // reduction deferred (phase 3)
    W4rArgs r;
    r.slabs = a.slabs; r.bslabs = a.bslabs; r.dw = d->dw; r.db = d->db;
    r.nsplit = g.nsplit; r.Co = d->o_real;
    r.Ci = (d->phase == 0 && a.Ctot < d->i_total - d->i_base) ? a.Ctot : d->i_total - d->i_base;
    r.CoP = g.CoP; r.CiP = g.CiP; r.iBase = d->i_base; r.iTotal = d->i_total;
    const long long total = (long long)r.Co * r.Ci;
    int lpe = 1;                           // lanes per element (small weight tensors only)
    while (lpe < 16 && (long long)lpe * 2 * total <= 65536 && lpe * 2 <= g.nsplit) lpe *= 2;
    r.perGroup = lpe;
    hipLaunchKernelGGL(wgrad_wino4_reduce_kernel, dim3((int)((total * lpe + 255) / 256)), dim3(256), 0, st, r);
    REFID_LAUNCH_CHECK("wgrad_wino4_reduce");
    return 0;
}
