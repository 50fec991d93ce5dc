// Winograd weight gradient of the 3x3 / stride-1 convolutions over 2 x 4 tiles of the output gradient, fp32 matrix cores
// (refid_wgrad_desc.algo = 5; autograd's conv-backward weight gradient; layers
// recurrent_sub_modules.py:659-678,719-726,755-758 under twoImage_event_recurrent_model.py:303).
//
// The weight gradient of a 3x3 conv is itself a correlation with a SMALL output (3x3) and a large "filter" (the output
// gradient), so the minimal-filtering identity applies with the roles swapped.  Per 2 x 4 tile of the output gradient dY and
// the 4 x 6 input window d it covers,
//
//     dW(3x3) = Ay^T [ sum_{tiles} (Gy dY Gx^T)_xi (x) (By^T d Bx)_xi ] Ax            xi = 0..23
//
// F(3,2) down the rows (points 0, +-1, inf: coefficients +-1 only) and F(3,4) along the rows (points 0, +-1, +-2, inf):
// 24 transform-domain GEMMs  dU_xi[i][o] += V_xi[tile][i] * Z_xi[tile][o]  over 2x4-pixel tiles = 3 fp32 MFMA-units per
// output pixel, where F(2x2,3x3) tiles (wgrad_wino.hip, algo 1) need 4 and the direct form 9.  The inverse transform
// Ay^T dU Ax (24 -> 9 values) is applied once per weight by the slab reduction.  Gx rows are scaled to integers
// ({1,0,0,0} {1,1,1,1} {1,-1,1,-1} {1,2,4,8} {1,-2,4,-8} {0,0,0,1}); the scale lives in Ax^T, so every coefficient of the K
// loop is +-1, 2, 4 or 5 and rides in an FMA.
//
// Why 2 x 4 and not 4 x 4 (36 MFMA-units per 16 pixels, 2.25 per pixel): on gfx950 a vector instruction next to fp32 MFMAs
// costs the matrix pipe ~10 cycles (the F(2x2) tile at 2.5 per MFMA sits at 0.70 of the pipe, the 4 x 4 tile --
// csrc/experimental/wgrad_wino4.hip, algo 6: six transform rows = six waves, 32 x 32 channel tiles, three waves per SIMD --
// at 7 per MFMA reaches 0.44 WITHOUT any staging: both transforms of a 6 x 6 window cost more than the MFMAs they
// save).  One F(3,2) direction keeps the four-wave / 64 x 32-channel shape of the F(2x2) tile (a V operand feeds two
// MFMAs) and its +-1 row pass; the transforms are PACKED over two neighbouring tile columns (v_pk_* on register pairs
// straight out of ds_read2st64_b32), which is what brings this tile to ~2.3 vector instructions per MFMA.
//
// Mapping: workgroup = 256 threads = 4 waves; wave w owns row w of the F(3,2) transform (xi = 6w .. 6w+5) of a
// 64(o) x 32(i) channel tile: 12 accumulators = 192 registers, two workgroups = two waves per SIMD.  A wave needs only ITS
// row of Z and V, so both transforms are computed on the fly from the raw NHWC tiles in LDS (lane = channel: conflict-free
// reads), specialised per row (a wave-uniform switch selects one of four K loops).  K tile = 2 x 4 Winograd tiles (4 x 16
// output pixels; MFMA K half = tile row): a lane walks the four tile columns two at a time.
// Staging: the LDS image IS the memory layout ([pixel][32 channels]; the gradient tile as two 32-channel halves), so tiles
// travel global -> LDS by LDS-DMA (buffer_load ... lds: no staging registers -- the F(2x2) tile's 60 would not fit next to
// 192 accumulators --, no ds_write pass, hardware zero fill outside the image), two buffers of 29.5 KB, ONE barrier per
// K tile.  A halo row is 18 pixels = two 1 KB pieces + one 256-byte piece (4-byte DMA: 2 pixels x 32 channels); row
// validity is wave-uniform, only the column test is per lane.  Every workgroup walks a CONTIGUOUS range of K tiles
// (coordinates advance by scalar increments).
// Split-K slabs [split][24][o][i] persist over the T recurrent steps exactly like wgrad_wino.hip's (phase 1 / 2 / 3); the
// bias gradient is the transform point (1, 1) of Gy dY Gx^T (rows {1,1} x {1,1,1,1}: the tile sum), summed by wave 1.
#include "common.h"
#include "wgrad_args.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

// Timing experiments only (tools/probes/w24_ablate.py builds the variants; results are wrong for n != 0):
//   1: no DMA requests (stale tiles: no global traffic, no LDS writes)   2: no MFMAs (operands kept alive)
//   3: no LDS reads (operands from opaque registers)   4: no barrier   5: 1 + 4   6: 1 + 3 + 4 (transforms + MFMAs alone)
//   7: every request fetches the workgroup's FIRST tile again (same instructions, same LDS writes, cache hits instead of HBM)
#ifndef REFID_W24_ABLATE
#define REFID_W24_ABLATE 0
#endif
#define W24_NO_DMA (REFID_W24_ABLATE == 1 || REFID_W24_ABLATE == 5 || REFID_W24_ABLATE == 6)
#define W24_NO_MFMA (REFID_W24_ABLATE == 2)
#define W24_NO_LDS (REFID_W24_ABLATE == 3 || REFID_W24_ABLATE == 6)
#define W24_NO_BAR (REFID_W24_ABLATE == 4 || REFID_W24_ABLATE == 5 || REFID_W24_ABLATE == 6)

namespace {

__device__ __forceinline__ float w24_fake(float seed) { asm volatile("" : "+v"(seed)); return seed; }

constexpr int IT = 32;                         // input-channel tile; output-channel tile = 32 NS (NS = 2; 1 for c_o <= 32): the
                                               // gradient tile is staged as NS 32-channel halves
constexpr int TC = 4;                          // Winograd tile columns per K tile (tile rows: 2 = the MFMA K halves)
constexpr int GH = 4, GW = 4 * TC;             // output-gradient pixels of a K tile
constexpr int XH = GH + 2, XW = GW + 2;        // input halo
constexpr int X_BYTES = XH * XW * IT * 4;      // 13,824
constexpr int GS_BYTES = GH * GW * 32 * 4;     // 8,192 per 32-channel half
constexpr int buf_bytes(int ns) { return X_BYTES + ns * GS_BYTES; }     // 30,208 (NS = 2) / 22,016
constexpr int lds24_bytes(int ns) { return 2 * buf_bytes(ns); }         // 60,416: two workgroups per CU / 44,032: three
constexpr int NXI = 24;
typedef __attribute__((address_space(3))) void* lds_ptr24;
typedef __attribute__((address_space(3))) const float lds_cf24;     // typed LDS pointers: 32-bit, ds_read instructions
typedef float f32x2 __attribute__((ext_vector_type(2)));

// {a.lo or a.hi, b.lo or b.hi} in ONE instruction; the result is opaque to the compiler, which otherwise un-packs every
// transform instruction downstream of a pair it would have to assemble from two registers
template <int AH, int BH>
__device__ __forceinline__ f32x2 w24_pair(f32x2 a, f32x2 b) {
    f32x2 d;
    if constexpr (AH == 0 && BH == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (AH == 1 && BH == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (AH == 0 && BH == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

struct W24Args {
    const float* g[REFID_WGRAD_MAX_GROUPS]; const float* inA[REFID_WGRAD_MAX_GROUPS]; const float* inB[REFID_WGRAD_MAX_GROUPS];
    int groups;
    int ldG, Co;
    int ldA, ldB, Ca, Ctot;
    float* slabs; float* bslabs;
    int N, H, W, Ho, Wo, pad;
    int tilesX, tilesY, ntiles, nsplit;
    int CoP, CiP;
    int accum;
    // DOWN form (conv_down, 4x4 stride 2 pad 1): H, W above are the PHASE image's size (= Ho, Wo), pad = 1
    int Hin, Win, ncoT;
};

// The transforms of one wave (F(3,2) row I), PACKED over two consecutive tile columns (steps s, s+1): a ds_read2st64 of
// (column c, column c + 4) lands in a register pair, every transform instruction is a v_pk_* on such pairs, and the MFMAs
// of step s / s+1 take the low / high halves.  LDS bases: xb[k] -> X[2 kh][column k][li], gb[k] -> dY[half 0][2 kh][column
// k][li] (k = column mod 4), xe / xo = columns 0 / 1 again: every read is base + a multiple of 256 bytes (an input row is
// 9 x 256 B, a gradient row 8 x 256 B, a gradient half 32 x 256 B, four columns 512 B), i.e. ds_read2st64_b32 with no
// address arithmetic.  The bases are opaque to the compiler (it would re-base with a v_add per 1 KB window), and one base
// per column residue makes the offset-NEIGHBOURS it pairs exactly (column c, column c + 4).
template <int I>
struct W24Row {
    static constexpr int XR = XW * IT, GR = GW * 32, GSUB = GH * GW * 32;
    static constexpr int XA = I == 0 ? 0 : (I == 2 ? 2 : (I == 3 ? 3 : 1));     // v-row = X[XA] +- X[XB]
    static constexpr int XB = I == 0 ? 2 : (I == 2 ? 1 : (I == 3 ? 1 : 2));
    static constexpr bool XPLUS = I == 1;

    template <class T>
    static __device__ __forceinline__ T x_comb(T a, T b) { return XPLUS ? a + b : a - b; }
    // window columns 4s+2 .. 4s+5 (and their partners four columns on), row-transformed
    static __device__ __forceinline__ void x_rows(lds_cf24* const (&xb)[4], int s, f32x2 (&t)[6]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * s + 2 + c;
            lds_cf24* p = xb[col & 3] + (col & ~3) * IT;
            const f32x2 a = W24_NO_LDS ? f32x2{w24_fake(1.f + c), w24_fake(2.f)} : f32x2{p[XA * XR], p[XA * XR + 4 * IT]};
            const f32x2 b = W24_NO_LDS ? f32x2{w24_fake(3.f + c), w24_fake(4.f)} : f32x2{p[XB * XR], p[XB * XR + 4 * IT]};
            t[2 + c] = x_comb(a, b);
        }
    }
    // the tile's first two window columns as ONE pair (column 0, column 1): neighbours in memory, a plain ds_read2_b32
    static __device__ __forceinline__ f32x2 x_rows01(lds_cf24* xe) {
        if (W24_NO_LDS) return f32x2{w24_fake(5.f), w24_fake(6.f)};
        return x_comb(f32x2{xe[XA * XR], xe[XA * XR + IT]}, f32x2{xe[XB * XR], xe[XB * XR + IT]});
    }
    // gradient tile columns 4s .. 4s+3 (and partners) of one 32-channel half, row-transformed: {g0, g0+g1, g0-g1, g1}
    static __device__ __forceinline__ void g_rows(lds_cf24* const (&gb)[4], int s, int sub, f32x2 (&x)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * s + c;
            lds_cf24* p = gb[col & 3] + sub * GSUB + (col & ~3) * 32;
            if (W24_NO_LDS) { x[c] = f32x2{w24_fake(7.f + c), w24_fake(8.f + sub)}; continue; }
            if constexpr (I == 0) x[c] = f32x2{p[0], p[4 * 32]};
            else if constexpr (I == 3) x[c] = f32x2{p[GR], p[GR + 4 * 32]};
            else {
                const f32x2 g0 = f32x2{p[0], p[4 * 32]}, g1 = f32x2{p[GR], p[GR + 4 * 32]};
                x[c] = I == 1 ? g0 + g1 : g0 - g1;
            }
        }
    }
    // V = (row) Bx over the six window columns t[0..5]
    static __device__ __forceinline__ void v_cols(const f32x2 (&t)[6], f32x2 (&v)[6]) {
        v[0] = 4.f * t[0] + (-5.f * t[2] + t[4]);
        v[1] = (t[3] + t[4]) - 4.f * (t[1] + t[2]);
        v[2] = 4.f * (t[1] - t[2]) + (t[4] - t[3]);
        const f32x2 d42 = t[4] - t[2], d31 = t[3] - t[1];
        v[3] = 2.f * d31 + d42;
        v[4] = d42 - 2.f * d31;
        v[5] = 4.f * t[1] + (-5.f * t[3] + t[5]);
    }
    // Z = (row) Gx^T over the four tile columns
    static __device__ __forceinline__ void z_cols(const f32x2 (&x)[4], f32x2 (&z)[6]) {
        const f32x2 e = x[0] + x[2], o = x[1] + x[3], e4 = 4.f * x[2] + x[0], o4 = 4.f * x[3] + x[1];
        z[0] = x[0]; z[1] = e + o; z[2] = e - o; z[3] = 2.f * o4 + e4; z[4] = e4 - 2.f * o4; z[5] = x[3];
    }
};

// DOWN = true: the weight gradient of conv_down (4x4, stride 2, pad 1; recurrent_sub_modules.py:12-14).  A stride-2 conv is four
// stride-1 convs with 2x2 taps on the input's PARITY PHASES  P_pq[Y][X] = in[2Y + p][2X + q]:  tap ky reads phase p = (ky - 1) & 1
// at row offset (ky - 1 - p) / 2, i.e. offsets {0, +1} for p = 0 and {-1, 0} for p = 1.  So per phase the wanted 2x2 taps are a
// sub-block of the 3x3 correlation this kernel computes on (P_pq, dY) with pad 1 -- ky = 2u + p - 1 for u = 0..2 where that
// lies in 0..3 --: 4 phases x 3 = 12 fp32 MFMA-units per output pixel instead of the direct form's 16.  The only differences:
// the phase (grid z) selects the slab block, and the DMA gathers every other pixel of every other row (per-lane offsets, so it
// costs nothing); the reduction scatters the 2x2 sub-blocks into the 4x4 gradient.
template <int NS, bool DOWN>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 3) void wgrad_wino24_kernel(const W24Args a) {
    constexpr int OT = 32 * NS, BUF_BYTES = buf_bytes(NS), XPS = DOWN ? 2 : 1;     // XPS: input pixels per phase pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int phase = DOWN ? blockIdx.z / a.ncoT : 0, php = phase >> 1, phq = phase & 1;
    const int co0 = (DOWN ? blockIdx.z % a.ncoT : blockIdx.z) * OT, ci0 = blockIdx.y * IT;
    const int split = blockIdx.x;
    const int Hin = DOWN ? a.Hin : a.H, Win = DOWN ? a.Win : a.W;

    // the input-channel tile lies in one source (host: c_a % 32 == 0 for two sources), so the descriptor is workgroup-uniform;
    // a tile beyond the sources (first recurrent step: no second source yet) keeps a valid descriptor, all lanes out of range
    const bool xFromA = ci0 < a.Ca || ci0 >= a.Ctot;
    const int xld = xFromA ? a.ldA : a.ldB;
    const long long gpixAll = (long long)a.N * a.Ho * a.Wo, xpixAll = (long long)a.N * Hin * Win;
    const int limG = (int)min(gpixAll * a.ldG * 4, 0x7fffffffLL), limX = (int)min(xpixAll * xld * 4, 0x7fffffffLL);
    const int ntAll = a.ntiles * a.groups;
    const int chunk = (ntAll + a.nsplit - 1) / a.nsplit;
    const int p0 = min(split * chunk, ntAll), p1 = min(p0 + chunk, ntAll);

    // coordinates of the next tile to request (wave-uniform; advanced by increments)
    int qg, qn, qy, qx;
    {
        int t = p0 < ntAll ? p0 : 0;
        qg = t / a.ntiles; t -= qg * a.ntiles;
        qx = t % a.tilesX; t /= a.tilesX;
        qy = t % a.tilesY; qn = t / a.tilesY;
    }
    // ---- DMA roles.  16-byte pieces: lane -> (pixel lane>>3 of 8, channel quad lane&7); 4-byte tail piece of a halo row:
    // lane -> (pixel 16 + lane>>5, channel lane&31).  Three per-lane constants (byte offset of the lane inside a piece); a
    // piece's offset is (tile-dependent scalar) + constant.  FAST path (the tile touches neither the left nor the right image
    // border and the channel tiles are full -- every layer of the network away from the borders): no per-lane test at all,
    // an out-of-image ROW is a scalar select of an out-of-range base: one v_add per DMA instruction.  Otherwise out-of-range
    // columns / channels are forced out of range with OR masks.
    const int xq = ci0 + (lane & 7) * 4, xt = ci0 + (lane & 31), gq = co0 + (lane & 7) * 4;
    const int xlc = ((lane >> 3) * XPS * xld + (xFromA ? xq : xq - a.Ca)) * 4;      // bytes from the piece's first pixel
    const int xlt = ((lane >> 5) * XPS * xld + (xFromA ? xt : xt - a.Ca)) * 4;
    const int glc = ((lane >> 3) * a.ldG + gq) * 4;
    const bool fullch = ci0 + IT <= a.Ctot && co0 + OT <= a.Co;                     // workgroup-uniform
    constexpr int OOB = 0x7ff00000;                        // base of a dead row: + any lane constant (< 1 MB) stays out of range
    // Scalar state of the walk: the current time step's tensors (reloaded from the argument block only when the walk crosses
    // into the next step -- an s_load per request would put a full scalar-memory latency in front of every tile), the byte
    // offset of the next tile's first halo / gradient pixel (advanced by one tile width; recomputed at the end of a tile row)
    const int xRowB = XPS * Win * xld * 4, gRowB = a.Wo * a.ldG * 4;      // bytes per (phase) image row
    auto x_origin = [&](int n, int ty, int tx) {           // byte offset of phase pixel (ty GH - pad, tx GW - pad) of sample n
        return ((n * Hin + XPS * (ty * GH - a.pad) + php) * Win + XPS * (tx * GW - a.pad) + phq) * xld * 4;
    };
    const float* gPtr = a.g[qg];
    const float* xPtr = xFromA ? a.inA[qg] : a.inB[qg];
    int xTile = x_origin(qn, qy, qx);
    int gTile = ((qn * a.Ho + qy * GH) * a.Wo + qx * GW) * a.ldG * 4;
    auto request = [&](int buf) {
        const int oy0 = qy * GH, ox0 = qx * GW;
        const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gPtr), 0, limG, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xPtr), 0, limX, 0x00020000);
        char* xdst = smem + buf * BUF_BYTES;
        char* gdst = xdst + X_BYTES;
        const bool fast = fullch && ix0 >= 0 && ix0 + XW <= a.W && ox0 + GW <= a.Wo && xld * 8 * XPS * 4 < 0x100000 && a.ldG * 8 * 4 < 0x100000;
        // gradient (half, row) pairs q = 4 half + row of this wave: NS = 2: wave 0 q = 0, wave 1 q = 1, wave 2 q = 2..4, wave 3
        // q = 5..7 (with the halo rows: 8 / 8 / 9 / 9 DMA instructions); NS = 1: waves 2 / 3 two rows each (6 / 6 / 7 / 7)
        const int q0 = NS == 2 ? (wave < 2 ? wave : 3 * wave - 4) : 2 * (wave - 2), nq = NS == 2 ? (wave < 2 ? 1 : 3) : (wave < 2 ? 0 : 2);
        if (fast) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = wave + 4 * k;
                if (r < XH && !W24_NO_DMA) {
                    const int base = (unsigned)(iy0 + r) < (unsigned)a.H ? xTile + r * xRowB : OOB;
                    char* dst = xdst + r * (XW * IT * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)dst, 16, base + xlc, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)(dst + 1024), 16, base + 8 * XPS * xld * 4 + xlc, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)(dst + 2048), 4, base + 16 * XPS * xld * 4 + xlt, 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < nq && !W24_NO_DMA) {
                    const int q = q0 + k, r = q & 3, sub = q >> 2;
                    const int base = oy0 + r < a.Ho ? gTile + r * gRowB + sub * 128 : OOB;
                    char* dst = gdst + sub * GS_BYTES + r * (GW * 32 * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr24)dst, 16, base + glc, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr24)(dst + 1024), 16, base + 8 * a.ldG * 4 + glc, 0, 0, 0);
                }
            }
        } else {
            const int l8 = lane >> 3, l32 = lane >> 5;
            const int xbadq = xq < a.Ctot ? 0 : -1, xbadt = xt < a.Ctot ? 0 : -1;
            const int gbad0 = gq < a.Co ? 0 : -1, gbad1 = gq + 32 < a.Co ? 0 : -1;         // (NS = 1: no second half)
            const int cx0 = (unsigned)(ix0 + l8) < (unsigned)a.W ? 0 : -1;               // column tests (per lane)
            const int cx1 = (unsigned)(ix0 + 8 + l8) < (unsigned)a.W ? 0 : -1;
            const int cxt = (unsigned)(ix0 + 16 + l32) < (unsigned)a.W ? 0 : -1;
            const int cg0 = ox0 + l8 < a.Wo ? 0 : -1, cg1 = ox0 + 8 + l8 < a.Wo ? 0 : -1;
            // halo rows: wave w moves row w, waves 0 / 1 also rows 4 / 5
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = wave + 4 * k;
                if (r < XH && !W24_NO_DMA) {
                    const int rbad = (unsigned)(iy0 + r) < (unsigned)a.H ? 0 : -1;
                    const int base = xTile + r * xRowB;                                   // bytes, < 2^31 for live lanes (host check)
                    char* dst = xdst + r * (XW * IT * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)dst, 16, (base + xlc) | xbadq | rbad | cx0, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)(dst + 1024), 16,
                                                             (base + 8 * XPS * xld * 4 + xlc) | xbadq | rbad | cx1, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr24)(dst + 2048), 4,
                                                             (base + 16 * XPS * xld * 4 + xlt) | xbadt | rbad | cxt, 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < nq && !W24_NO_DMA) {
                    const int q = q0 + k, r = q & 3, sub = q >> 2;
                    const int rbad = oy0 + r < a.Ho ? 0 : -1;
                    const int base = gTile + r * gRowB + sub * 128;
                    const int gbad = sub ? gbad1 : gbad0;
                    char* dst = gdst + sub * GS_BYTES + r * (GW * 32 * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr24)dst, 16, (base + glc) | gbad | rbad | cg0, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr24)(dst + 1024), 16,
                                                             (base + 8 * a.ldG * 4 + glc) | gbad | rbad | cg1, 0, 0, 0);
                }
            }
        }
        // advance
        if (REFID_W24_ABLATE == 7) return;
        qx += 1;
        if (qx != a.tilesX) {
            xTile += GW * XPS * xld * 4;
            gTile += GW * a.ldG * 4;
        } else {
            qx = 0;
            qy += 1;
            if (qy == a.tilesY) {
                qy = 0;
                qn += 1;
                if (qn == a.N) {
                    qn = 0;
                    qg = min(qg + 1, a.groups - 1);
                    gPtr = a.g[qg];
                    xPtr = xFromA ? a.inA[qg] : a.inB[qg];
                }
            }
            xTile = x_origin(qn, qy, 0);
            gTile = (qn * a.Ho + qy * GH) * a.Wo * a.ldG * 4;
        }
    };

    f32x16 acc[6][NS];                                     // [j][o half]
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int sm = 0; sm < NS; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][sm][r] = 0.f;
    float bs[NS];
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) bs[sm] = 0.f;

    // The LDS bases point into buffer 0 and never change: buffer 1 is 118 x 256 bytes further, an immediate of the same
    // ds_read2st64 (the largest unit offset is 218 of 255), so the K loop is written for a PAIR of tiles (buffer 0, buffer 1)
    // and carries no address arithmetic at all.
    static_assert(BUF_BYTES % 256 == 0 && (BUF_BYTES + X_BYTES + (NS - 1) * GS_BYTES + 4 * GW * 32 * 4) / 256 < 256, "ds_read2st64 offset range");
    lds_cf24* xe = (lds_cf24*)smem + (2 * kh) * (XW * IT) + li;
    lds_cf24* ge = (lds_cf24*)(smem + X_BYTES) + (2 * kh) * (GW * 32) + li;
    lds_cf24* xb[4] = {xe, xe + IT, xe + 2 * IT, xe + 3 * IT};
    lds_cf24* gb[4] = {ge, ge + 32, ge + 2 * 32, ge + 3 * 32};
    asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xe));
    asm volatile("" : "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3]));

    auto kloop = [&](auto TI) {
        constexpr int I = decltype(TI)::value;
        using R = W24Row<I>;
        auto tile = [&](auto BUF) {
            constexpr int BO = decltype(BUF)::value * (BUF_BYTES / 4);          // floats
            lds_cf24* xbb[4] = {xb[0] + BO, xb[1] + BO, xb[2] + BO, xb[3] + BO};
            lds_cf24* gbb[4] = {gb[0] + BO, gb[1] + BO, gb[2] + BO, gb[3] + BO};
            f32x2 t[6], p4, p5;
#pragma unroll
            for (int s = 0; s < TC; s += 2) {
                // window columns 4s+k (low half) and 4s+4+k (high half); columns 0 / 1 of the low half are the previous
                // pair's columns 4 / 5 of the high half (p4 / p5 carry them; the tile's first two columns are read as a pair)
                R::x_rows(xbb, s, t);
                if (s == 0) {
                    const f32x2 a01 = R::x_rows01(xe + BO);
                    t[0] = w24_pair<0, 0>(a01, t[4]);
                    t[1] = w24_pair<1, 0>(a01, t[5]);
                } else {
                    t[0] = w24_pair<1, 0>(p4, t[4]);
                    t[1] = w24_pair<1, 0>(p5, t[5]);
                }
                p4 = t[4]; p5 = t[5];
                f32x2 v[6];
                R::v_cols(t, v);
#pragma unroll
                for (int sm = 0; sm < NS; ++sm) {
                    f32x2 x[4], z[6];
                    R::g_rows(gbb, s, sm, x);
                    R::z_cols(x, z);
                    if constexpr (I == 1) bs[sm] += z[1][0] + z[1][1];     // point (1, 1) = sum of the 2x4 gradient tile
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            if (W24_NO_MFMA) { asm volatile("" :: "v"(v[j][h]), "v"(z[j][h])); continue; }
                            acc[j][sm] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][h], z[j][h], acc[j][sm], 0, 0, 0);
                        }
                }
            }
        };
        auto sync = [&] {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's pieces of the tile have landed
            if (!W24_NO_BAR) __builtin_amdgcn_s_barrier(); // ... everybody's; and everybody is done with the other buffer
            __builtin_amdgcn_sched_barrier(0);
        };
        // (pairs in the loop, an odd last tile after it: an exit from the middle of the loop body makes the register allocator
        //  copy all 192 accumulators on one of the two paths)
        const int npairs = (p1 - p0) >> 1;
        if (p0 < p1) request(0);
        for (int i = 0; i < npairs; ++i) {
            sync();
            request(1);
            tile(std::integral_constant<int, 0>{});
            sync();
            if (p0 + 2 * i + 2 < p1) request(0);
            tile(std::integral_constant<int, 1>{});
        }
        if ((p1 - p0) & 1) {
            sync();
            tile(std::integral_constant<int, 0>{});
        }
    };
    switch (wave) {
        case 0: kloop(std::integral_constant<int, 0>{}); break;
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        default: kloop(std::integral_constant<int, 3>{}); break;
    }

    // ---- slab: [split][xi][co][ci]; D[ci][co]: lane li = output channel, register quad = 4 ci ------
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        // slab image of one split: [phase (DOWN: 4)][xi][co][ci]
        float* sl = a.slabs + ((long long)((split * (DOWN ? 4 : 1) + phase) * NXI + wave * 6 + j) * a.CoP) * a.CiP;
#pragma unroll
        for (int sm = 0; sm < NS; ++sm) {
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
    if (a.bslabs != nullptr && blockIdx.y == 0 && phase == 0 && wave == 1) {
        // wave 1 holds the tile sums of the output channels: the two tile rows (kh) by one shuffle -- fixed order
#pragma unroll
        for (int sm = 0; sm < NS; ++sm) {
            const float tot = bs[sm] + __shfl_xor(bs[sm], 32, 64);
            if (kh == 0) {
                float* dst = a.bslabs + (long long)split * a.CoP + co0 + sm * 32 + li;
                *dst = a.accum ? *dst + tot : tot;
            }
        }
    }
}

struct W24rArgs {
    const float* slabs; const float* bslabs; float* dw; float* db;
    int nsplit, Co, Ci, CoP, CiP, iBase, iTotal, perGroup;
    int nsplitW;                           // slabs behind `slabs` (= nsplit, or the folded count); bslabs always has nsplit rows
    int down;                              // (batched launch) the conv_down form of the stage
};

// First stage of the slab reduction.  Reading the slabs per (co, ci) element touches 64-256 contiguous bytes of each of 24 planes
// of every slab per wave instruction -- 0.5-0.9 TB/s, 113-360 us per weight (100 MB each).  Here a workgroup owns 1024
// consecutive floats of the slab image (4 KB contiguous per slab) and adds the slabs s, s + S, s + 2S, ... in order into partial
// slab s: plain streaming at ~6 TB/s.  The second stage (below) then reads S <= 16 slabs instead of nsplit.  Deterministic:
// fixed partition, fixed order.
__device__ __forceinline__ void w24_fold_body(const float* __restrict__ slabs, float* __restrict__ part, long long slabFloats,
                                              int nsplit, int S, int piece, int s) {
    const long long off = ((long long)piece * 256 + threadIdx.x) * 4;
    if (off >= slabFloats) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs + off);
    const long long st4 = slabFloats / 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int k = s;
    for (; k + S < nsplit; k += 2 * S) { a0 += p[k * st4]; a1 += p[(k + S) * st4]; }
    if (k < nsplit) a0 += p[k * st4];
    *reinterpret_cast<f32x4*>(part + (long long)s * slabFloats + off) = a0 + a1;
}
__global__ __launch_bounds__(256) void wgrad_wino24_fold_kernel(const float* __restrict__ slabs, float* __restrict__ part,
                                                                long long slabFloats, int nsplit, int S) {
    w24_fold_body(slabs, part, slabFloats, nsplit, S, blockIdx.x, blockIdx.y);
}

// the folds queued by phase-4 calls in ONE launch (before the batched element-wise stages): job j owns pieces x S workgroups
constexpr int FOLD_BATCH = 96;
struct FoldJob { const float* slabs; float* part; long long slabFloats; int nsplit, S; };
struct FoldBatch { FoldJob job[FOLD_BATCH]; int blk0[FOLD_BATCH + 1]; int n; };
static_assert(sizeof(FoldBatch) <= 4096, "kernel-argument block");
__global__ __launch_bounds__(256) void wgrad_fold_batch_kernel(const FoldBatch b) {
    int j = 0;
    for (int k = 1; k < b.n; ++k) j = (int)blockIdx.x >= b.blk0[k] ? k : j;       // (workgroup-uniform)
    const FoldJob& f = b.job[j];
    const int blk = (int)blockIdx.x - b.blk0[j];
    const int pieces = (int)((f.slabFloats / 4 + 255) / 256);
    w24_fold_body(f.slabs, f.part, f.slabFloats, f.nsplit, f.S, blk % pieces, blk / pieces);
}
thread_local std::vector<FoldJob> fold_queue;

// slab reduction + inverse transform dg = Ay^T dU Ax, accumulated into OIHW (9 contiguous floats).  Deterministic (no atomics).
// A workgroup owns EB = 256 / LPE consecutive (co, ci) elements (ci fastest); thread (sub = t / EB, e = t % EB) adds the slabs
// sub, sub + LPE, ... of its element in order -- a 16-lane group reads 64 contiguous bytes of one slab plane (the first form of
// this kernel put the LPE partial sums of ONE element on adjacent lanes: 16-byte pieces of 16 slabs per instruction, 0.19 of HBM)
// -- the LPE partial sums meet in LDS and are added in index order by the element's first thread, which applies the transform.
//   Ay^T = [ 1 1/2  1/2 0 ]      Ax^T = [ 1/4 -1/6 -1/6 1/24  1/24 0 ]
//          [ 0 1/2 -1/2 0 ]             [ 0   -1/6  1/6 1/12 -1/12 0 ]
//          [ 0 1/2  1/2 1 ]             [ 0   -1/6 -1/6 1/6   1/6  1 ]
__device__ __forceinline__ void w24_reduce_body(const W24rArgs& a, const int blk, float* part) {
    const long long plane = (long long)a.CoP * a.CiP;
    const long long slabStride = NXI * plane;
    const int lpe = a.perGroup, eb = 256 / lpe;
    const int el = threadIdx.x % eb, sub = threadIdx.x / eb;
    const long long e = (long long)blk * eb + el;           // (co, ci), ci fastest
    const bool live = e < (long long)a.Co * a.Ci;
    const int ci = live ? (int)(e % a.Ci) : 0, co = live ? (int)(e / a.Ci) : 0;
    {
        const float* p = a.slabs + (long long)co * a.CiP + ci;
        float u[NXI];
#pragma unroll
        for (int x = 0; x < NXI; ++x) u[x] = 0.f;
        if (live) {
            for (int k = sub; k < a.nsplitW; k += lpe) {
#pragma unroll
                for (int x = 0; x < NXI; ++x) u[x] += p[k * slabStride + x * plane];
            }
        }
        if (lpe > 1) {
#pragma unroll
            for (int x = 0; x < NXI; ++x) part[threadIdx.x * (NXI + 1) + x] = u[x];
            __syncthreads();
            if (sub == 0) {
                for (int s2 = 1; s2 < lpe; ++s2) {
#pragma unroll
                    for (int x = 0; x < NXI; ++x) u[x] += part[(s2 * eb + el) * (NXI + 1) + x];
                }
            }
        }
        if (live && sub == 0) {
            // t[p][j] = sum_i Ay^T[p][i] u[i][j] ;  dg[p][q] = sum_j t[p][j] Ax^T[q][j]
            constexpr float c4 = 0.25f, c6 = 1.f / 6.f, c12 = 1.f / 12.f, c24 = 1.f / 24.f;
            float dg[9];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                float t[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float m = 0.5f * (u[6 + j] + u[12 + j]), d = 0.5f * (u[6 + j] - u[12 + j]);
                    t[j] = pp == 0 ? u[j] + m : (pp == 1 ? d : m + u[18 + j]);
                }
                const float s12 = t[1] + t[2], d21 = t[2] - t[1], s34 = t[3] + t[4], d34 = t[3] - t[4];
                dg[pp * 3 + 0] = c4 * t[0] - c6 * s12 + c24 * s34;
                dg[pp * 3 + 1] = c6 * d21 + c12 * d34;
                dg[pp * 3 + 2] = c6 * (s34 - s12) + t[5];
            }
            float* dst = a.dw + ((long long)co * a.iTotal + a.iBase + ci) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] += dg[k];
        }
    }
    if (a.db != nullptr && blk == 0) {
        for (int c2 = threadIdx.x; c2 < a.Co; c2 += 256) {
            float s = 0.f;
            for (int k = 0; k < a.nsplit; ++k) s += a.bslabs[(long long)k * a.CoP + c2];
            a.db[c2] += s;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_wino24_reduce_kernel(const W24rArgs a) {
    __shared__ float part[256 * (NXI + 1)];
    w24_reduce_body(a, blockIdx.x, part);
}

// DOWN form: per (co, ci) the four phases' 24 planes -> four 3x3 blocks -> the 2x2 sub-block of each that exists in the 4x4
// gradient (ky = 2u + p - 1, kx = 2v + q - 1), accumulated into OIHW (16 contiguous floats).  Reads the (folded) slab image
// [split][phase][xi][co][ci]; one thread per element, slabs in order (deterministic).
__device__ __forceinline__ void w24_reduce_down_body(const W24rArgs& a, const int blk) {
    const long long plane = (long long)a.CoP * a.CiP;
    const long long slabStride = 4 * NXI * plane;
    const long long gid = (long long)blk * 256 + threadIdx.x;
    const long long e = gid >> 2;                              // (co, ci), ci fastest; four threads = the four phases: every tap
    const int ph = (int)(gid & 3);                             // of the 4x4 gradient belongs to exactly one of them
    const bool live = e < (long long)a.Co * a.Ci;
    const int ci = live ? (int)(e % a.Ci) : 0, co = live ? (int)(e / a.Ci) : 0;
    if (live) {
        const float* p = a.slabs + (long long)co * a.CiP + ci + ph * NXI * plane;
        float u[NXI];
#pragma unroll
        for (int x = 0; x < NXI; ++x) u[x] = 0.f;
        for (int k = 0; k < a.nsplitW; ++k) {
#pragma unroll
            for (int x = 0; x < NXI; ++x) u[x] += p[k * slabStride + x * plane];
        }
        constexpr float c4 = 0.25f, c6 = 1.f / 6.f, c12 = 1.f / 12.f, c24 = 1.f / 24.f;
        float* dst = a.dw + ((long long)co * a.iTotal + a.iBase + ci) * 16;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            const int ky = 2 * pp + (ph >> 1) - 1;
            float t[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float m = 0.5f * (u[6 + j] + u[12 + j]), d = 0.5f * (u[6 + j] - u[12 + j]);
                t[j] = pp == 0 ? u[j] + m : (pp == 1 ? d : m + u[18 + j]);
            }
            const float s12 = t[1] + t[2], d21 = t[2] - t[1], s34 = t[3] + t[4], d34 = t[3] - t[4];
            const float r3[3] = {c4 * t[0] - c6 * s12 + c24 * s34, c6 * d21 + c12 * d34, c6 * (s34 - s12) + t[5]};
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int kx = 2 * q + (ph & 1) - 1;
                if (kx >= 0 && kx <= 3) dst[ky * 4 + kx] += r3[q];
            }
        }
    }
    if (a.db != nullptr && blk == 0) {
        for (int c2 = threadIdx.x; c2 < a.Co; c2 += 256) {
            float sacc = 0.f;
            for (int k = 0; k < a.nsplit; ++k) sacc += a.bslabs[(long long)k * a.CoP + c2];
            a.db[c2] += sacc;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_wino24_reduce_down_kernel(const W24rArgs a) { w24_reduce_down_body(a, blockIdx.x); }

// every queued reduction of the family in ONE launch (refid_wgrad_desc.phase = 4 + refid_wgrad_finish_flush)
struct W24rBatch { W24rArgs job[REFID_FINISH_BATCH]; int blk0[REFID_FINISH_BATCH + 1]; int n; };
static_assert(sizeof(W24rBatch) <= 4096, "kernel-argument block");
__global__ __launch_bounds__(256) void wgrad_wino24_reduce_batch_kernel(const W24rBatch b) {
    __shared__ float part[256 * (NXI + 1)];
    int j = 0;
    for (int k = 1; k < b.n; ++k) j = (int)blockIdx.x >= b.blk0[k] ? k : j;       // (workgroup-uniform)
    const int blk = (int)blockIdx.x - b.blk0[j];
    if (b.job[j].down) w24_reduce_down_body(b.job[j], blk);
    else w24_reduce_body(b.job[j], blk, part);
}

struct W24rQueued { W24rArgs r; int nblocks; };
thread_local std::vector<W24rQueued> w24_queue;

struct Geo24 { int ncoT, nciT, tilesX, tilesY, ntiles, nsplit, CoP, CiP; };

int ns_of(const refid_wgrad_desc* d) { return d->c_o <= 32 ? 1 : 2; }      // 32-channel output tile for the thin layers
bool is_down(const refid_wgrad_desc* d) { return d->kh == 4; }              // algo 7: conv_down through its parity phases

Geo24 geo24_of(const refid_wgrad_desc* d) {
    Geo24 g;
    const int OT = 32 * ns_of(d);
    g.ncoT = cdiv(d->c_o, OT);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;     // stable across steps
    g.nciT = cdiv(ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b, IT);
    g.tilesX = cdiv(d->wo, GW);
    g.tilesY = cdiv(d->ho, GH);
    g.ntiles = g.tilesX * g.tilesY * d->n;
    // two (NS = 1: three) workgroups per CU; a multiple of 8 splits keeps the workgroups of one K range on one XCD (grid x
    // is fastest)
    static const int wgs = []() { const char* e = getenv("REFID_W24_WGS"); return e ? atoi(e) : 512; }();
    int want = cdiv(ns_of(d) == 1 ? wgs * 3 / 2 : wgs, g.ncoT * g.nciT * (is_down(d) ? 4 : 1));
    if (want >= 8) want = want / 8 * 8;
    if (want < 1) want = 1;
    if (want > g.ntiles) want = g.ntiles;
    g.nsplit = want;
    g.CoP = g.ncoT * OT;
    g.CiP = g.nciT * IT;
    return g;
}

}  // namespace

// Generic first stage for every split-K slab family ([split][...]: slabFloats floats per split): S partial slabs -- as few as
// keep the streaming stage at >= 2048 workgroups of 4 KB pieces (64 x 64 weights: 16 partial slabs out of 256; 512 x 256
// weights: 1 out of 8); 0 = one slab only, nothing to fold.
int refid_slab_fold_count(long long slabFloats, int nsplit) {
    if (nsplit < 2) return 0;
    const long long pieces = (slabFloats / 4 + 255) / 256;
    int S = (int)((2048 + pieces - 1) / pieces);
    if (S > 16) S = 16;
    if (S >= nsplit) S = nsplit / 2;
    return S < 1 ? 1 : S;
}
int refid_launch_slab_fold(const float* slabs, float* part, long long slabFloats, int nsplit, int S, hipStream_t st) {
    static const bool batch_folds = !(getenv("REFID_FOLD_BATCH") && getenv("REFID_FOLD_BATCH")[0] == '0');
    if (batch_folds && refid_finish_defer_now()) {         // phase 4: with the other queued folds (refid_slab_fold_flush)
        fold_queue.push_back({slabs, part, slabFloats, nsplit, S});
        return 0;
    }
    hipLaunchKernelGGL(wgrad_wino24_fold_kernel, dim3((unsigned)((slabFloats / 4 + 255) / 256), S), dim3(256), 0, st, slabs, part,
                       slabFloats, nsplit, S);
    REFID_LAUNCH_CHECK("wgrad_slab_fold");
    return 0;
}

int refid_slab_fold_flush(hipStream_t st) {
    size_t at = 0;
    while (at < fold_queue.size()) {
        FoldBatch b;
        memset(&b, 0, sizeof(b));
        int n = 0;
        long long blk = 0;
        for (; n < FOLD_BATCH && at < fold_queue.size(); ++n, ++at) {
            const FoldJob& f = fold_queue[at];
            const long long nb = ((f.slabFloats / 4 + 255) / 256) * f.S;
            if (n > 0 && blk + nb > 0x3fffffffLL) break;
            b.job[n] = f;
            b.blk0[n] = (int)blk;
            blk += nb;
        }
        b.blk0[n] = (int)blk; b.n = n;
        hipLaunchKernelGGL(wgrad_fold_batch_kernel, dim3((unsigned)blk), dim3(256), 0, st, b);
        if (hipGetLastError() != hipSuccess) { fold_queue.clear(); refid_set_error("wgrad_fold_batch: launch failed"); return 1; }
    }
    fold_queue.clear();
    return 0;
}

int refid_wino24_finish_flush(hipStream_t st) {
    size_t at = 0;
    while (at < w24_queue.size()) {
        W24rBatch b;
        memset(&b, 0, sizeof(b));
        int n = 0, blk = 0;
        for (; n < REFID_FINISH_BATCH && at < w24_queue.size(); ++n, ++at) {
            b.job[n] = w24_queue[at].r;
            b.blk0[n] = blk;
            blk += w24_queue[at].nblocks;
        }
        b.blk0[n] = blk; b.n = n;
        hipLaunchKernelGGL(wgrad_wino24_reduce_batch_kernel, dim3(blk), dim3(256), 0, st, b);
        if (hipGetLastError() != hipSuccess) { w24_queue.clear(); refid_set_error("wgrad_wino24_reduce_batch: launch failed"); return 1; }
    }
    w24_queue.clear();
    return 0;
}

namespace {
long long slab_floats(const refid_wgrad_desc* d, const Geo24& g) { return (long long)(is_down(d) ? 4 : 1) * NXI * g.CoP * g.CiP; }
}  // namespace

size_t refid_wgrad_wino24_workspace_bytes(const refid_wgrad_desc* d) {
    const Geo24 g = geo24_of(d);
    const long long slab = slab_floats(d, g);
    return ((size_t)g.nsplit * slab + (size_t)g.nsplit * g.CoP + (size_t)refid_slab_fold_count(slab, g.nsplit) * slab) * sizeof(float);
}

int refid_wgrad_wino24_launch(const refid_wgrad_desc* d, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0}, attr_done1{0}, attr_done2{0}, attr_done3{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_wino24_kernel<2, false>, lds24_bytes(2), "wgrad_wino24")) return rc;
    if (int rc = refid_lds_attr_once(attr_done1, &wgrad_wino24_kernel<1, false>, lds24_bytes(1), "wgrad_wino24<1>")) return rc;
    if (int rc = refid_lds_attr_once(attr_done2, &wgrad_wino24_kernel<2, true>, lds24_bytes(2), "wgrad_wino24<down>")) return rc;
    if (int rc = refid_lds_attr_once(attr_done3, &wgrad_wino24_kernel<1, true>, lds24_bytes(1), "wgrad_wino24<1, down>")) return rc;
    const Geo24 g = geo24_of(d);
    const bool down = is_down(d);
    const long long slab = slab_floats(d, g);
    if (down)
        REFID_CHECK(d->kw == 4 && d->stride == 2 && d->pad == 1 && d->h % 2 == 0 && d->w % 2 == 0 && d->ho == d->h / 2 && d->wo == d->w / 2,
                    "wgrad (algo 7): a 4x4 stride-2 pad-1 conv over an even-sized input");
    REFID_CHECK(d->c_b == 0 || d->c_a % IT == 0, "wgrad (Winograd 2x4 tiles): c_a must be a multiple of %d for two sources", IT);
    REFID_CHECK(d->ld_g % 4 == 0 && d->ld_a % 4 == 0 && (d->c_b == 0 || d->ld_b % 4 == 0) && d->c_o % 4 == 0 &&
                    d->c_a % 4 == 0 && d->c_b % 4 == 0,
                "wgrad (Winograd 2x4 tiles): pitches and channel counts must be multiples of 4 floats (16-byte LDS-DMA pieces)");
    {
        const long long lim = 0x7fffffffLL;
        REFID_CHECK((long long)d->n * d->ho * d->wo * d->ld_g * 4 < lim && (long long)d->n * d->h * d->w * d->ld_a * 4 < lim &&
                        (d->c_b == 0 || (long long)d->n * d->h * d->w * d->ld_b * 4 < lim),
                    "wgrad (Winograd 2x4 tiles): tensor too large for 32-bit buffer offsets (use algo 0)");
    }
    W24Args a;
    const int ngrp = d->groups > 1 ? d->groups : 1;
    REFID_CHECK(ngrp <= REFID_WGRAD_MAX_GROUPS, "wgrad: at most %d grouped time steps", REFID_WGRAD_MAX_GROUPS);
    for (int k = 0; k < REFID_WGRAD_MAX_GROUPS; ++k) {
        const bool on = k > 0 && k < ngrp;
        a.g[k] = k == 0 ? d->g : (on ? d->g_more[k - 1] : d->g);
        a.inA[k] = k == 0 ? d->in_a : (on ? d->in_a_more[k - 1] : d->in_a);
        a.inB[k] = k == 0 ? d->in_b : (on ? d->in_b_more[k - 1] : d->in_b);
        REFID_CHECK(a.g[k] && a.inA[k] && (d->c_b == 0 || a.inB[k]), "wgrad: null tensor pointer in group %d", k);
        REFID_CHECK(((uintptr_t)a.g[k] | (uintptr_t)a.inA[k] | (uintptr_t)(d->c_b ? a.inB[k] : nullptr)) % 16 == 0,
                    "wgrad (Winograd 2x4 tiles): tensors must be 16-byte aligned (group %d)", k);
    }
    a.groups = ngrp;
    a.ldG = d->ld_g; a.Co = d->c_o;
    a.ldA = d->ld_a; a.ldB = d->ld_b;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.slabs = d->slabs;
    a.bslabs = d->db ? d->slabs + (size_t)g.nsplit * slab : nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pad = d->pad;
    a.Hin = d->h; a.Win = d->w; a.ncoT = g.ncoT;
    if (down) { a.H = d->ho; a.W = d->wo; a.pad = 1; }      // the phase image
    a.tilesX = g.tilesX; a.tilesY = g.tilesY; a.ntiles = g.ntiles; a.nsplit = g.nsplit;
    a.CoP = g.CoP; a.CiP = g.CiP;
    a.accum = (d->phase == 2);
    if (d->phase != 3) {
        if (down) {
            if (ns_of(d) == 1)
                hipLaunchKernelGGL((wgrad_wino24_kernel<1, true>), dim3(g.nsplit, g.nciT, 4 * g.ncoT), dim3(256), lds24_bytes(1), st, a);
            else
                hipLaunchKernelGGL((wgrad_wino24_kernel<2, true>), dim3(g.nsplit, g.nciT, 4 * g.ncoT), dim3(256), lds24_bytes(2), st, a);
        } else if (ns_of(d) == 1)
            hipLaunchKernelGGL((wgrad_wino24_kernel<1, false>), dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), lds24_bytes(1), st, a);
        else
            hipLaunchKernelGGL((wgrad_wino24_kernel<2, false>), dim3(g.nsplit, g.nciT, g.ncoT), dim3(256), lds24_bytes(2), st, a);
        REFID_LAUNCH_CHECK("wgrad_wino24");
    }
    if (d->phase == 1 || d->phase == 2) return 0;          // reduction deferred (phase 3)
    W24rArgs r;
    r.slabs = a.slabs; r.bslabs = a.bslabs; r.dw = d->dw; r.db = d->db;
    r.nsplit = g.nsplit; r.Co = d->o_real;
    int nred = g.nsplit;                   // slabs the element-wise stage reads
    if (const int S = refid_slab_fold_count(slab, g.nsplit)) {
        const long long slabFloats = slab;
        float* part = d->slabs + (size_t)g.nsplit * slab + (size_t)g.nsplit * g.CoP;      // (behind the bias slabs)
        if (int rc = refid_launch_slab_fold(a.slabs, part, slabFloats, g.nsplit, S, st)) return rc;
        r.slabs = part;
        nred = S;
    }
    r.Ci = (d->phase == 0 && a.Ctot < d->i_total - d->i_base) ? a.Ctot : d->i_total - d->i_base;
    r.CoP = g.CoP; r.CiP = g.CiP; r.iBase = d->i_base; r.iTotal = d->i_total;
    const long long total = (long long)r.Co * r.Ci;
    int lpe = 1;                           // threads per element (small weight tensors only)
    while (lpe < 16 && (long long)lpe * 2 * total <= 65536 && lpe * 2 <= nred) lpe *= 2;
    r.perGroup = lpe;
    r.nsplitW = nred;
    r.down = down ? 1 : 0;
    if (refid_finish_defer_now()) {
        for (const W24rQueued& q : w24_queue)              // (two jobs on one gradient block would race: flush first)
            if (q.r.dw == r.dw && q.r.iBase == r.iBase) {
                if (int rc = refid_slab_fold_flush(st)) return rc;
                if (int rc = refid_wino24_finish_flush(st)) return rc;
                break;
            }
        w24_queue.push_back({r, down ? (int)((total * 4 + 255) / 256) : (int)((total + 256 / lpe - 1) / (256 / lpe))});
        return 0;
    }
    if (down) {
        hipLaunchKernelGGL(wgrad_wino24_reduce_down_kernel, dim3((int)((total * 4 + 255) / 256)), dim3(256), 0, st, r);
        REFID_LAUNCH_CHECK("wgrad_wino24_reduce_down");
        return 0;
    }
    hipLaunchKernelGGL(wgrad_wino24_reduce_kernel, dim3((int)((total + 256 / lpe - 1) / (256 / lpe))), dim3(256), 0, st, r);
    REFID_LAUNCH_CHECK("wgrad_wino24_reduce");
    return 0;
}
