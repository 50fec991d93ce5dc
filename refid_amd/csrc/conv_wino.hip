// Fused Winograd F(2x2, 3x3) convolution tile on the gfx950 fp32 matrix cores.
//
// 88 % of the REFID hot path's FLOPs are 3x3 / stride-1 convolutions (SURVEY.md section 8a,
// Appendix B).  gfx950 has no reduced-precision shortcut for fp32 (no xf32/TF32; the fp32 MFMA
// runs at the vector rate), so the lever that remains is arithmetic: Winograd's minimal
// filtering computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per
// (cin, cout) pair instead of 36 -- 2.25x fewer MFMAs at identical fp32 arithmetic class
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A            (Lavin & Gray, 2016)
// with the transforms' +-1 / 0.5 constants exact in fp32.
//
//   out = mask( post( pre(conv3x3(src) + bias) + res ) ),  src = in_a or [in_a | in_b]
// (same contract, epilogue and two-source input as conv_igemm.hip; also serves the input
// gradient on flipped/transposed weights.)
//
// Mapping (one workgroup = 256 threads = 4 waves, 2 workgroups per CU):
//   * workgroup tile = 4x32 output pixels = 32 Winograd tiles (2 tile rows x 16 tile cols)
//     x 64 output channels; K walks input channels in chunks of 8.
//   * the 16 transform-domain GEMMs  M_xi[cout][tile] += U_xi[cout][c] * V_xi[c][tile]  run on
//     v_mfma_f32_32x32x2_f32.  Wave w owns transform ROW i = w (xi = 4i..4i+3) for all 64 output channels:
//     8 accumulators = 128 registers, two waves per SIMD.
//   * on gfx950 VALU and LDS instructions do NOT hide under MFMAs (tools/probes/mfma_valu_overlap.hip: ~2.3
//     matrix-pipe cycles per VALU instruction, ~28 per ds_read_b128, from either wave of the SIMD), so the
//     loop is organised to minimise non-MFMA instructions per MFMA: a wave transforms only its own row of
//     B^T d B (8 x ds_read_b128 + 32 VALU per chunk) and feeds 32 MFMAs from it (both column tiles reuse it).
//   * both MFMA operands live in REGISTERS: the raw 6x34-pixel input halo of the next chunk is
//     prefetched global->VGPR->LDS (double buffered, 6.5 KB each) while the current chunk's MFMAs
//     run.  One barrier per chunk.
//   * transformed weights U = G g G^T are produced once per step by the pack kernel in the
//     [chunk][xi][cout][8] layout; a lane's fragment is one 16-byte buffer load per (xi, column tile) (a wave
//     reads 1 KB contiguous), prefetched one chunk ahead -- weights never pass through LDS.
//   * output transform: each lane reduces its 4 accumulators along j in registers, the four rows are combined
//     through LDS once (64 KB, after the K loop), and every wave finishes one output row parity of one column
//     tile with the fused 16-byte epilogue.
#include "common.h"
#include "conv_args.h"
#include <cstdlib>

namespace {

constexpr int TW = 32;                  // output pixels per workgroup row
constexpr int KC = 8;                   // input channels per chunk
constexpr int HWD = TW + 2;
constexpr int LDS_BYTES = 64 * 66 * 16;   // 2 raw buffers (13-22 KB) in the K loop; 66 KB row exchange afterwards
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = -1;                 // voffset 0xFFFFFFFF: buffer loads return 0 (hardware range check)

// tools/probes/wino_trace.py builds this file with -DREFID_WINO_TRACE: every workgroup records wall-clock stamps
// (100 MHz) at its phase boundaries + the CU it ran on, to see where a tile's time goes.  Never in the product build.
#ifdef REFID_WINO_TRACE
__device__ unsigned long long* g_wino_trace = nullptr;
#define WINO_STAMP(slot)                                                                                     \
    do {                                                                                                     \
        if (g_wino_trace && threadIdx.x == 0)                                                                \
            g_wino_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64();        \
    } while (0)
// fine-grained: shader-clock stamps of the K-loop phases of waves 0 and 3 of ONE workgroup (g_wino_ktrace_wg)
__device__ unsigned long long* g_wino_ktrace = nullptr;
__device__ int g_wino_ktrace_wg = -1;
#define WINO_KSTAMP(ch, slot)                                                                                  \
    do {                                                                                                       \
        if (ktrace && (threadIdx.x & 63) == 0)                                                                 \
            g_wino_ktrace[(((threadIdx.x >> 6) * 64 + ((ch) & 63)) * 8) + (slot)] = clock64();                  \
    } while (0)
#else
#define WINO_STAMP(slot) do {} while (0)
#define WINO_KSTAMP(ch, slot) do {} while (0)
#endif

// <NTN, MTN>: a wave's second dimension is either two 32-channel column tiles (NTN = 2: workgroup tile
// 4x32 pixels x 64 channels) or two 4-row pixel tiles (MTN = 2: 8x32 pixels x 32 channels, 32-channel layers).
template <int NTN, int MTN>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvKArgs a) {
    static_assert(NTN * MTN == 2, "a wave owns one transform row x 2 (column | pixel) tiles");
    constexpr int TH = 4 * MTN;             // output rows per workgroup
    constexpr int BN = 32 * NTN;            // output channels per workgroup
    constexpr int HP = (TH + 2) * HWD;      // raw input halo pixels
    constexpr int R_F4 = 2 * HP;            // one raw halo buffer: [quad][pixel] float4
    constexpr int R_ITEMS = (R_F4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sR = reinterpret_cast<f32x4*>(smem);            // two raw halo buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row i owned by this wave (xi = 4i .. 4i+3)
    WINO_STAMP(0);
#ifdef REFID_WINO_TRACE
    if (g_wino_trace && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wino_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    // XCD-aware work mapping: workgroup b runs on XCD b % 8 (observed dispatch order; used for L2
    // affinity only).  The ncot channel tiles of one pixel tile are consecutive on the SAME XCD, so
    // the input halo they all read is fetched into that XCD's L2 once instead of ncot times.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int bt = (slot / a.ncot) * 8 + xcd;                    // pixel-tile index
    if (bt >= a.tilesX * a.tilesY * a.N) return;
    const int n0 = (slot % a.ncot) * BN;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int n = bt / a.tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // ---- loaders: buffer loads (wave-uniform descriptor + per-chunk scalar offset + a per-thread
    // byte offset that never changes across chunks); out-of-image pixels / out-of-range weight rows
    // carry an out-of-range offset and come back as zeros -- no branches, no per-chunk address math.
    const int limA = (int)min((long long)a.N * a.H * a.W * a.ldA * 4, 0x7fffffffLL);
    const int limB = a.inB ? (int)min((long long)a.N * a.H * a.W * a.ldB * 4, 0x7fffffffLL) : 0;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)min((long long)a.nchunks * 16 * a.CoutPad * KC * 4, 0x7fffffffLL), 0x00020000);
    const int q = tid & 1;
    int voA[R_ITEMS], voB[R_ITEMS];
#pragma unroll
    for (int it = 0; it < R_ITEMS; ++it) {
        const int hp = (tid >> 1) + it * 128;
        const int iy = oy0 - a.pad + hp / HWD, ix = ox0 - a.pad + hp % HWD;
        const bool ok = hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int pix = (n * a.H + iy) * a.W + ix;
        voA[it] = ok ? pix * a.ldA * 4 + q * 16 : OOB;
        voB[it] = ok ? pix * a.ldB * 4 + q * 16 : OOB;
    }
    // U fragments of this lane: rows (cout) n0 + nt*32 + li, K quad kh, the wave's 4 xi -- straight from the
    // packed [chunk][xi][cout][8] weights (a wave reads 1 KB contiguous per fragment), never through LDS.
    int voU[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int urow = a.coBase + n0 + nt * 32 + li;
        voU[nt] = (urow < a.CoutPad) ? ((ti * 4 * a.CoutPad + urow) * KC + kh * 4) * 4 : OOB;
    }
    // split-K (small grids only): this workgroup reduces chunks [kc0, kc1) and writes a raw partial output
    const int kper = ((a.nchunks + a.ksplit - 1) / a.ksplit + 1) & ~1;       // even, so chunk parity == buffer parity
    const int kc0 = blockIdx.y * kper, kc1 = min(a.nchunks, kc0 + kper);
    const int uStep = a.CoutPad * KC * 4;                  // bytes between xi and xi+1
    const int uChunk = 16 * a.CoutPad * KC * 4;            // bytes per K chunk

    f32x4 rr[R_ITEMS], ufA[4 * NTN], ufB[4 * NTN];

    auto load_raw = [&](int ch, f32x4 (&rr)[R_ITEMS]) {
        const int c0 = ch * KC;                            // chunk-uniform: Ca % 8 == 0 for two sources
        const bool fromA = c0 < a.Ca;
        const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
        const bool qok = c0 + q * 4 < a.Ctot;              // Ctot % 8 == 4: upper quad of the last chunk
        // ONE descriptor, selected with scalar ops: a per-lane choice between two descriptors makes the compiler
        // emit a readfirstlane "waterfall" loop plus full vmcnt(0) drains inside the phase
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(fromA ? a.inA : a.inB), 0, fromA ? limA : limB, 0x00020000);
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            const int vo = qok ? (fromA ? voA[it] : voB[it]) : OOB;
            rr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
        }
    };
    auto store_raw = [&](int buf, const f32x4 (&rr)[R_ITEMS]) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            const int hp = (tid >> 1) + it * 128;
            if (hp < HP) sR[buf * R_F4 + q * HP + hp] = rr[it];
        }
    };
    auto load_u = [&](int ch, f32x4 (&dst)[4 * NTN]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
                dst[j * NTN + nt] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voU[nt], ch * uChunk + j * uStep, 0));
    };

    // acc[j][t]: transform column j, second-dimension tile t (column tile nt or pixel tile mt)
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    // B^T row i of the 4x4 input patch of this lane's tile: t = d[P] + sgn * d[M]
    //   i=0: d0 - d2   i=1: d1 + d2   i=2: d2 - d1   i=3: d1 - d3
    const int rowP = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rowM = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const int hp0 = (2 * (li >> 4)) * HWD + 2 * (li & 15);           // tile origin inside a 4-row pixel tile
    const int offP = hp0 + rowP * HWD, offM = hp0 + rowM * HWD;

    // one K chunk: cur = this chunk's U fragments, nxt = register set the next chunk's are prefetched into
#ifdef REFID_WINO_TRACE
    const bool ktrace = g_wino_ktrace && (int)blockIdx.x == g_wino_ktrace_wg;
#endif
    auto phase = [&](int ch, f32x4 (&cur)[4 * NTN], f32x4 (&nxt)[4 * NTN]) {
        const bool more = ch + 1 < kc1;
        WINO_KSTAMP(ch, 0);
        // Everything still in flight was issued one phase ago and is needed NOW (U(ch) by the MFMAs, raw(ch+1) by
        // store_raw).  Stating that as an explicit vmcnt(0) keeps the compiler's conservative, path-merged counters
        // from draining THIS phase's prefetches inside the MFMA section.  simm16: vmcnt 0, expcnt 7, lgkmcnt 15.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        WINO_KSTAMP(ch, 1);
        if (more) {
            store_raw((ch + 1) & 1, rr);             // raw(ch+1): loaded one phase ago
            load_u(ch + 1, nxt);
            if (ch + 2 < kc1) load_raw(ch + 2, rr);
        }
        WINO_KSTAMP(ch, 2);
        // Measured on gfx950 (tools/probes/mfma_valu_overlap.hip): every VALU instruction costs ~2.3 and every
        // ds_read_b128 ~28 cycles of matrix-pipe time, from either wave of the SIMD -- they do not hide under
        // MFMAs.  So a wave transforms only ITS row of B^T d B (8 reads, 32 VALU per pixel tile) and reuses it for
        // both column tiles.
        const f32x4* r = sR + (ch & 1) * R_F4 + kh * HP;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            const f32x4* rp = r + offP + mt * 4 * HWD;
            const f32x4* rm = r + offM + mt * 4 * HWD;
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = rp[b] + rm[b] * sgn;
            f32x4 v[4];
            v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)          // consecutive MFMAs hit different accumulators
                        acc[j][NTN == 2 ? nt : mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            cur[j * NTN + nt][kk], v[j][kk], acc[j][NTN == 2 ? nt : mt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        WINO_KSTAMP(ch, 3);
        __syncthreads();                             // raw(ch) consumed by every wave; raw(ch+1) visible
        WINO_KSTAMP(ch, 4);
    };

    // prologue: raw(0) -> LDS, U(0) -> regs; raw(1) in flight.  All three requests go out back to back (ONE memory
    // latency up front instead of two: the accumulators are not live yet, a second staging set costs nothing here)
    if (kc0 < kc1) {
        f32x4 rr0[R_ITEMS];
        load_raw(kc0, rr0);
        if (kc0 + 1 < kc1) load_raw(kc0 + 1, rr);
        load_u(kc0, ufA);
        store_raw(0, rr0);
    }
    __syncthreads();
    WINO_STAMP(1);

    for (int ch = kc0; ch < kc1; ch += 2) {
        phase(ch, ufA, ufB);
        if (ch + 1 < kc1) phase(ch + 1, ufB, ufA);
    }
    WINO_STAMP(2);

    // ---- output transform --------------------------------------------------------------------------
    // lane: tile li, channels (r&3)+8(r>>2)+4kh of a 32-channel tile;  acc[j][t] = M[i][j]
    //   R_i[b] = sum_j M[i][j] A[j][b] :  b=0: M0+M1+M2   b=1: M1-M2-M3          (in registers)
    //   Y[a][b] = sum_i A^T[a][i] R_i[b]:  a=0: R0+R1+R2   a=1: R1-R2-R3          (across the 4 waves, through LDS)
    // exchange layout: [i][b][t][register quad][kh][33] float4 (the 33 spreads the banks for the readers below)
    constexpr int XL = 66;
    f32x4* xch = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 r0, r1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * rq + k;
                r0[k] = acc[0][t][r] + acc[1][t][r] + acc[2][t][r];
                r1[k] = acc[1][t][r] - acc[2][t][r] - acc[3][t][r];
            }
            xch[(((ti * 2 + 0) * 2 + t) * 4 + rq) * XL + kh * 33 + li] = r0;
            xch[(((ti * 2 + 1) * 2 + t) * 4 + rq) * XL + kh * 33 + li] = r1;
        }
    // residual / mask of this thread's 8 output pieces are requested NOW, so their latency hides behind the exchange
    // barrier and the LDS reads below (the accumulators are dead: registers are plentiful here)
    constexpr int NIT = (TH * TW * (BN / 4)) / 256;
    const bool pre = a.vecOK && a.ksplit == 1 && (a.res != nullptr || a.mask != nullptr);
    f32x4 pres[NIT], pmask[NIT];
    if (pre) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = it * 256 + tid;
            const int c4 = f % (BN / 4), pr = f / (BN / 4);
            const int oy = oy0 + pr / TW, ox = ox0 + pr % TW, j0 = n0 + c4 * 4;
            const bool ok = oy < a.Ho && ox < a.Wo && j0 + 3 < a.Cout;
            const long long op = (long long)(n * a.Ho + oy) * a.Wo + ox;
            pres[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            pmask[it] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (ok && a.res) pres[it] = *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
            if (ok && a.mask) pmask[it] = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
        }
    }
    __syncthreads();
    WINO_STAMP(3);

    // ---- fused epilogue, COALESCED: thread -> (output pixel, channel quad) in memory order, so bias / residual /
    // mask loads and the store are contiguous 1 KB per wave instruction (the MFMA D layout would scatter 16-byte
    // pieces over 64 rows; at 256^2 the store phase was 15 % of the kernel).  The row combination happens here,
    // on the way out of LDS.
    constexpr int C4 = BN / 4;                              // float4 per pixel
#pragma unroll
    for (int it = 0; it < (TH * TW * C4) / 256; ++it) {
        const int f = it * 256 + tid;
        const int c4 = f % C4, pr = f / C4;
        const int row = pr / TW, col = pr % TW;
        const int c = c4 * 4;
        const int nt = c >> 5, rq = (c & 31) >> 3, ckh = (c & 7) >> 2;
        const int mt = row >> 2, oa = row & 1, tile = ((row & 3) >> 1) * 16 + (col >> 1), ob = col & 1;
        const int t = (NTN == 2) ? nt : mt;
        const int oy = oy0 + row, ox = ox0 + col;
        const int j0 = n0 + c;
        if (oy >= a.Ho || ox >= a.Wo || j0 >= a.Cout) continue;
        const f32x4* xp = xch + (((oa * 2 + ob) * 2 + t) * 4 + rq) * XL + ckh * 33 + tile;   // row i0 = oa
        const float sg = oa ? -1.f : 1.f;                   // a=0: R0+R1+R2 ; a=1: R1-R2-R3
        f32x4 v = xp[0] + (xp[16 * XL] + xp[32 * XL]) * sg;  // rows i0, i0+1, i0+2 (16*XL float4 per row)
        const long long op = (long long)(n * a.Ho + oy) * a.Wo + ox;
        if (a.ksplit > 1) {                                 // raw partial sums; refid_wino_splitk_finish applies the epilogue
            *reinterpret_cast<f32x4*>(a.out + blockIdx.y * a.wsStride + op * a.ldO + j0) = v;
            continue;
        }
        const bool vec = a.vecOK && (j0 + 3 < a.Cout);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const float* bp = a.bias + a.coBase + j0;
            if (vec) bv = *reinterpret_cast<const f32x4*>(bp);
            else
#pragma unroll
                for (int k = 0; k < 4; ++k) if (j0 + k < a.Cout) bv[k] = bp[k];
        }
        v += bv;
        lrelu4(v, a.slopePre, a.slopePre != 1.f);
        if (vec) {
            if (a.res) v += pre ? pres[it] : *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
            lrelu4(v, a.slopePost, a.slopePost != 1.f);
            if (a.mask) {
                const f32x4 mv = pre ? pmask[it] : *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
            }
            *reinterpret_cast<f32x4*>(a.out + op * a.ldO + j0) = v;
            if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + op * a.ldO2 + j0) = v + *reinterpret_cast<const f32x4*>(a.add2 + op * a.ldA2 + j0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= a.Cout) break;
                float tv = v[k];
                if (a.res) tv += a.res[op * a.ldR + j0 + k];
                tv = lrelu(tv, a.slopePost);
                if (a.mask) tv *= (a.mask[op * a.ldM + j0 + k] > 0.f) ? 1.f : a.slopeMask;
                a.out[op * a.ldO + j0 + k] = tv;
                if (a.out2) a.out2[op * a.ldO2 + j0 + k] = tv + a.add2[op * a.ldA2 + j0 + k];
            }
        }
    }
    WINO_STAMP(4);
}

// out = mask( post( pre( sum_s ws[s] + bias ) + res ) ) for the split-K partial outputs (channels padded to 4 in ws)
__global__ __launch_bounds__(256) void wino_splitk_finish_kernel(const ConvKArgs a, const float* __restrict__ ws, int ldW,
                                                                long long npix, int C4) {
    const long long total = npix * C4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long op = e / C4;
        const int j0 = (int)(e % C4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + op * ldW + j0);
        for (int s = 1; s < a.ksplit; ++s) v += *reinterpret_cast<const f32x4*>(ws + s * a.wsStride + op * ldW + j0);
        if (a.vecOK && j0 + 3 < a.Cout) {
            if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + a.coBase + j0);
            lrelu4(v, a.slopePre, a.slopePre != 1.f);
            if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
            lrelu4(v, a.slopePost, a.slopePost != 1.f);
            if (a.mask) {
                const f32x4 mv = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
            }
            *reinterpret_cast<f32x4*>(a.out + op * a.ldO + j0) = v;
            if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + op * a.ldO2 + j0) = v + *reinterpret_cast<const f32x4*>(a.add2 + op * a.ldA2 + j0);
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (j0 + k >= a.Cout) break;
            float t = v[k] + (a.bias ? a.bias[a.coBase + j0 + k] : 0.f);
            t = lrelu(t, a.slopePre);
            if (a.res) t += a.res[op * a.ldR + j0 + k];
            t = lrelu(t, a.slopePost);
            if (a.mask) t *= (a.mask[op * a.ldM + j0 + k] > 0.f) ? 1.f : a.slopeMask;
            a.out[op * a.ldO + j0 + k] = t;
            if (a.out2) a.out2[op * a.ldO2 + j0 + k] = t + a.add2[op * a.ldA2 + j0 + k];
        }
    }
}

}  // namespace

namespace {
// Small problems (deep levels, small batches) leave most of the 256 CUs idle with one long K loop per workgroup:
// split K over grid.y into a partial-sum workspace and finish with a streaming epilogue kernel.
//   split_mode 1 ("sample"): the split depends on the per-sample geometry only, so a sample's result is bit-identical
//       whatever the batch size (tests rely on that property);
//   split_mode 2 ("auto"): by total grid size -- best throughput at 1-2 samples per GPU;
//   split_mode 0: never.
struct WinoPlan { int th, bn, ks; dim3 grid; };

WinoPlan wino_plan(ConvKArgs& a, int split_mode) {
    WinoPlan p;
    const bool narrow = a.Cout <= 32;           // 32-channel layers: split pixels instead of channels
    p.th = narrow ? 8 : 4; p.bn = narrow ? 32 : 64;
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, p.th);
    a.nchunks = cdiv(a.Ctot, KC);
    a.ncot = cdiv(a.Cout, p.bn);
    p.grid = dim3(round_up(a.tilesX * a.tilesY * a.N, 8) * a.ncot);
    const int nwg = (split_mode == 2) ? (int)p.grid.x : a.tilesX * a.tilesY * a.ncot * 8;   // "sample": as if N = 8
    int ks = 1;
    if (split_mode && nwg <= 256 && a.nchunks >= 16) {
        ks = 512 / nwg;
        if (ks > a.nchunks / 8) ks = a.nchunks / 8;
        if (ks > 8) ks = 8;
        if (ks < 1) ks = 1;
    }
    p.ks = ks;
    return p;
}
}  // namespace

#ifdef REFID_WINO_TRACE
extern "C" int refid_wino_trace_set(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int refid_wino_ktrace_set(unsigned long long* buf, int wg) {
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_wino_ktrace), &buf, sizeof(buf)) == hipSuccess &&
            hipMemcpyToSymbol(HIP_SYMBOL(g_wino_ktrace_wg), &wg, sizeof(wg)) == hipSuccess) ? 0 : 1;
}
#endif

size_t refid_wino3x3_workspace_bytes(const ConvKArgs& ka, int split_mode) {
    ConvKArgs a = ka;
    const WinoPlan p = wino_plan(a, split_mode);
    if (p.ks == 1) return 0;
    return (size_t)p.ks * a.N * a.Ho * a.Wo * round_up(a.Cout, 4) * sizeof(float);
}

// Generic finishing pass of a split-K launch (also used by the direct tile): out = epilogue(sum_s ws[s]) over an
// output of npix pixels x a.Cout channels (a.out / ldO / res / mask / bias describe the final tensor).
int refid_launch_splitk_finish(const ConvKArgs& f, const float* ws, int ldW, long long npix, hipStream_t st) {
    const long long tot4 = npix * (ldW / 4);
    int nb = (int)((tot4 + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(wino_splitk_finish_kernel, dim3(nb), dim3(256), 0, st, f, ws, ldW, npix, ldW / 4);
    REFID_LAUNCH_CHECK("splitk_finish");
    return 0;
}

namespace {
// CU count per device (persistent-tile grid size); 0 = query failed -> the persistent tile is not used
int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 0;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = -1;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c > 0 ? c : 0;
}
}  // namespace

int refid_launch_wino3x3(const ConvKArgs& ka, float* ws, size_t ws_bytes, int split_mode, int tile_hint, hipStream_t st) {
    ConvKArgs a = ka;
    // (tile_hint 2 once selected a persistent one-wave-per-SIMD tile: measured 10-30 % slower at every config-2 shape -- a
    //  lone wave per SIMD pays ~130 cycles per buffer-load issue and every LDS / barrier stall itself; two co-resident
    //  workgroups hide exactly that, DESIGN.md section 7, profiles/r02_wino_* -- and was removed in round 6)
    // no workspace from the caller = no split (still correct, one K loop per workgroup)
    const WinoPlan pl = wino_plan(a, ws ? split_mode : 0);
    const bool narrow = a.Cout <= 32;
    dim3 grid = pl.grid;
    static std::atomic<unsigned long long> done12{0}, done21{0};
    if (int rc = refid_lds_attr_once(done12, &conv_wino_kernel<1, 2>, LDS_BYTES, "conv_wino")) return rc;
    if (int rc = refid_lds_attr_once(done21, &conv_wino_kernel<2, 1>, LDS_BYTES, "conv_wino")) return rc;
    const int ks = pl.ks;
    if (ks == 1) {
        if (narrow) hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(256), LDS_BYTES, st, a);
        else hipLaunchKernelGGL((conv_wino_kernel<2, 1>), grid, dim3(256), LDS_BYTES, st, a);
        REFID_LAUNCH_CHECK("conv_wino");
        return 0;
    }
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const int ldW = round_up(a.Cout, 4);
    const size_t need = (size_t)ks * npix * ldW * sizeof(float);
    if (need > ws_bytes || (reinterpret_cast<uintptr_t>(ws) & 15)) {
        refid_set_error("conv_wino: split-K workspace too small or misaligned (%zu bytes given, %zu needed: "
                        "refid_conv_workspace_bytes)", ws_bytes, need);
        return 1;
    }
    ConvKArgs p = a;                       // partial pass: raw sums into the workspace
    p.ksplit = ks; p.wsStride = npix * ldW; p.out = ws; p.ldO = ldW;
    grid.y = ks;
    if (narrow) hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(256), LDS_BYTES, st, p);
    else hipLaunchKernelGGL((conv_wino_kernel<2, 1>), grid, dim3(256), LDS_BYTES, st, p);
    REFID_LAUNCH_CHECK("conv_wino/splitk");
    ConvKArgs f = a;
    f.ksplit = ks; f.wsStride = npix * ldW;
    return refid_launch_splitk_finish(f, ws, ldW, npix, st);
}
