// Direct 3x3 / stride-1 convolution tile with SPLIT fp32 operands on the gfx950 bf16 matrix cores.
//
//   out = mask( post( pre(conv3x3(src) + bias) + res ) ),   src = in_a or [in_a | in_b]
// (same contract, epilogue and two-source input as conv_igemm.hip / conv_wino.hip; serves the forward conv and,
// on flipped/transposed weights, the input gradient.)
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 32 MAC/clk/SIMD, the bf16 one (v_mfma_f32_32x32x16_bf16) at
// 512 -- sixteen times the rate with the same fp32 accumulator.  An fp32 number is EXACTLY the sum of three bf16
// numbers (8 + 8 + 8 significand bits: h = rne(v), m = rne(v - h), l = v - h - m), so
//     a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)  + O(2^-24 |ab|)
// -- six bf16 MFMAs reproduce the fp32 product to one fp32 rounding (the three dropped cross terms are below
// 2^-24 relative), at 6/16 of the fp32 MFMA's pipe time.  Measured (tools/bench_split.py):
//   * against the fp32-MFMA DIRECT tiles (convolutions with no Winograd form: conv_down 4x4 stride 2 and its input
//     gradient) six products are 1.6-2x faster at the same distance from the float64 result -- the default there;
//   * against the fp32 Winograd F(2x2,3x3) tile (2.25x fewer multiplies) they only tie: 2.25 * 6/16 = 0.84 of its
//     matrix-pipe time on paper, but the bf16 pipe with real operands is power limited (0.68 of nominal), and the
//     accumulated rounding is 2-3x Winograd's (8e-6 vs 3e-6 on O(1) sums: the fp32 class either way,
//     tests/test_hip_conv.py::test_split_tile_accuracy_classes) -- an experiment switch for 3x3;
//   * TERMS = 3 keeps the first three products (two planes per operand, 2^-16 relative: finer than TF32, which is
//     what the reference's own GPU path multiplies with by default): 1.4-1.55x faster than the Winograd tile, an
//     explicit opt-in (compute_dtype bf16x3);  TERMS = 1 is plain bf16 operands (compute_dtype bf16);
//   * F16 (round 6, mfma_terms 19; conv_down forward / input gradient): TWO fp16 planes h = rne16(v), l = rne16(v - h) carry 22
//     bits, three v_mfma_f32_32x32x16_f16 (hh + hl + lh) give the product to ~2^-22 -- half the MFMAs of the six-bf16-product
//     form at the fp32 class's accuracy.  fp16's range (2^-14 .. 2^16) is bridged by exact power-of-two scales: the weights by
//     2^eW per packing (refid_pack_conv_weights_split_f16: exponent in the packing's header), the activations per WORKGROUP
//     and online along K -- a staged halo pixel feeds several output pixels and all four waves, so the scale is the
//     workgroup's: every stage's largest |x| (thread max -> wave max -> four LDS slots read after the stage's barrier) is
//     compared with the reference exponent; a stage more than 2^6 above it multiplies the accumulators by 2^(E - E') first
//     (exact), the output epilogue undoes 2^(8 - E + eW) with v_ldexp.  (conv_wino6.hip's fp16 form scales per tile: there a
//     lane's operand column is its own.)
//
// Mapping (one workgroup = 256 threads = 4 waves, one per SIMD; two workgroups per CU cover each other's staging,
// barriers, prologue and epilogue -- the regime the trace of the Winograd tile showed to work on this chip):
//   * workgroup tile = (4*MT) x 32 output pixels x (32*NT) output channels; wave w owns rows w*MT .. of the tile
//     for all column tiles: MT*NT accumulators of 16 registers.
//   * K walks input channels in chunks of 8.  The bf16 MFMA's K = 16 is filled with TWO TAPS x 8 channels: lanes
//     0-31 supply tap 2j, lanes 32-63 tap 2j+1 of the same 8 channels (five steps per chunk, the tenth tap is zero
//     weights: 10 % of the issued MFMAs, the price of a chunk that is small enough -- 47 KB of LDS -- for two
//     workgroups per CU).  Per chunk the raw fp32 halo ((4*MT+2) x 34 pixels x 8 channels) is fetched once with
//     buffer loads (hardware zero fill outside the image), split into its bf16 planes on the way to LDS
//     ([plane][pixel] 16-byte slots) and re-used by all taps and all output channels; the weights come pre-split
//     from the pack kernel ([chunk][plane][tap 0..9][cout][8]) and are staged next to it.  A lane's MFMA fragment is
//     ONE ds_read_b128 (conflict free: consecutive lanes, consecutive slots).
//   * per step: PL*(MT+NT) LDS reads feed TERMS*MT*NT MFMAs (12 reads : 24 MFMAs at MT=NT=2); measured with
//     tools/probes/mfma_lds_feed.hip: fragment reads at this density hide completely under bf16 MFMAs with two waves
//     per SIMD (0.88 of the nominal peak = the same as with no reads at all).  The next chunk's global loads are in
//     flight during the MFMAs; the split + LDS store sits between two barriers (register double buffering).
//   * epilogue: each wave turns its accumulator rows around in its own LDS strip (no barrier) so that every
//     bias / residual / mask load and the store is a contiguous 1 KB per wave instruction.
#include "common.h"
#include "conv_args.h"

// tools/probes/split_ablate.py builds this file with -DREFID_SPLIT_ABLATE=n (one piece of the kernel removed, results wrong) to
// price the pieces: 1 = no MFMAs, 2 = no operand split (raw registers stored as planes), 3 = global loads of chunk 0 in every
// stage (cache resident), 4 = no global loads in the K loop, 5 = no K loop, 6 = no epilogue traffic (no residual / mask loads,
// no stores), 7 = 5 + 6, 8 = no LDS fragment reads in the K loop (operands from registers).  Never in the product.
#ifndef REFID_SPLIT_ABLATE
#define REFID_SPLIT_ABLATE 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int F16_TGT = 8;              // a freshly scaled stage's largest |x| lands in [2^8, 2^9)
constexpr int F16_SLACK = 6;            // binades a later stage may exceed the reference before the accumulators are rescaled
constexpr int F16_E0 = F16_TGT + 1;     // initial (biased) reference exponent: any real data is larger
constexpr int F16_HEADER = 64;          // bytes in front of the fp16 planes: int eW (runtime.hip)
constexpr int TW = 32;                  // output pixels per tile row
constexpr int KC = 8;                   // input channels per chunk (x 2 taps = K of one bf16 MFMA)
constexpr int NTH = 256;
constexpr int OOB = -1;                 // voffset 0xFFFFFFFF: buffer loads return 0 (hardware range check)
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// tools/probes/split_trace.py builds this file with -DREFID_SPLIT_TRACE: every workgroup stamps the 100 MHz wall clock
// at its phase boundaries (+ the CU it ran on), one chosen workgroup stamps every K-loop phase.  Never in the product build.
#ifdef REFID_SPLIT_TRACE
__device__ unsigned long long* g_split_trace = nullptr;
__device__ unsigned long long* g_split_ktrace = nullptr;
__device__ int g_split_ktrace_wg = -1;
#define SPLIT_STAMP(slot)                                                                    \
    do {                                                                                     \
        if (g_split_trace && threadIdx.x == 0) g_split_trace[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#define SPLIT_KSTAMP(ch, slot)                                                               \
    do {                                                                                     \
        if (ktrace && (threadIdx.x & 63) == 0)                                               \
            g_split_ktrace[(((threadIdx.x >> 6) * 64 + ((ch) & 63)) * 8) + (slot)] = wall_clock64(); \
    } while (0)
#else
#define SPLIT_STAMP(slot) do {} while (0)
#define SPLIT_KSTAMP(ch, slot) do {} while (0)
#endif

// MODE 0: 3x3 / stride 1 (9 taps + a tenth of zero weights = 5 tap pairs per 8-channel sub-chunk).
// MODE 1: 4x4 / stride 2 / pad 1 (`conv_down` forward, recurrent_sub_modules.py:12-14) as a 2x2 / stride-1 conv over 2x2
//         input BLOCKS that start at odd coordinates (block b = rows 2b-1, 2b): output row oy reads rows 2oy-1 .. 2oy+2 =
//         blocks oy, oy+1, so K = 4 block taps x (4 positions inside a block x C channels) with no padding taps.  A stage
//         holds the two column positions (KS = 2) of one row position of 8 channels; the block gather is address
//         arithmetic in the loader.
// MODE 2: input gradient of conv_down: four output-parity classes (blockIdx.y), each a 2x2-tap stride-1 conv over the
//         output gradient with its own weights; output pixels (2y+py, 2x+px).
template <int MT_, int NT_, int PL_, int KS_, int MODE_, bool F16_ = false>
struct SCfg {
    static constexpr int MT = MT_, NT = NT_, PL = PL_, KS = KS_;   // KS: 8-channel sub-chunks per LDS stage (barrier pair)
    static constexpr int MODE = MODE_;
    static constexpr bool F16 = F16_;                       // two fp16 planes + scales instead of PL bf16 planes
    static constexpr int TH = 4 * MT, BN = 32 * NT;
    static constexpr int HWD = (MODE == 1) ? TW + 1 : TW + 2;
    static constexpr int NTAP = (MODE == 0) ? 10 : 4;        // taps per sub-chunk in LDS / packed weights
    static constexpr int NSTEP = NTAP / 2;                   // MFMA steps (tap pairs) per sub-chunk
    static constexpr int HP = ((MODE == 1) ? TH + 1 : TH + 2) * HWD;   // halo pixels (MODE 1: input blocks)
    static constexpr int A_SLOTS = HP;                      // per plane: [pixel]
    static constexpr int B_SLOTS = NTAP * BN;               // per plane: [tap][cout]
    static constexpr int A_ITEMS = (A_SLOTS + NTH - 1) / NTH;
    static constexpr int B_ITEMS = (PL * B_SLOTS + NTH - 1) / NTH;
    static constexpr int C4 = BN / 4;                       // float4 per output pixel
    static constexpr int XS = C4 + 1;                       // padded pixel pitch of the epilogue strip
    static constexpr int LDS_LOOP = KS * PL * (A_SLOTS + B_SLOTS) * 16 + (F16 ? 16 : 0);   // (+ the four wave maxima)
    static constexpr int LDS_EPI = 4 * 32 * XS * 16;
    static constexpr int LDS_BYTES = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
};

// v (8 fp32 channels) -> PL bf16 planes; the planes sum to v exactly for PL = 3 and to 2^-17 |v| for PL = 2
template <int PL>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[PL]) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = k < 4 ? v0[k] : v1[k - 4];
        const __bf16 h = (__bf16)v;
        p0[k] = h;
        if (PL >= 2) {
            const float r = v - (float)h;
            const __bf16 m = (__bf16)r;
            p1[k] = m;
            if (PL == 3) p2[k] = (__bf16)(r - (float)m);
        }
    }
    pl[0] = __builtin_bit_cast(f32x4, p0);
    if constexpr (PL >= 2) pl[1] = __builtin_bit_cast(f32x4, p1);
    if constexpr (PL == 3) pl[2] = __builtin_bit_cast(f32x4, p2);
}

// v (8 fp32 channels, scaled into fp16's range) -> two fp16 planes, h + l = v to 22 bits
__device__ __forceinline__ void split8h(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[2]) {
    f16x8 h, l;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 ab = {k < 4 ? v0[k] : v1[k - 4], k < 4 ? v0[k + 1] : v1[k - 3]};
        const f16x2 hh = __builtin_convertvector(ab, f16x2);
        const f32x2 r = ab - __builtin_convertvector(hh, f32x2);
        const f16x2 ll = __builtin_convertvector(r, f16x2);
        h[k] = hh[0]; h[k + 1] = hh[1];
        l[k] = ll[0]; l[k + 1] = ll[1];
    }
    pl[0] = __builtin_bit_cast(f32x4, h);
    pl[1] = __builtin_bit_cast(f32x4, l);
}

template <int MT, int NT, int PL, int KS, int MODE, bool F16 = false>
__global__ __launch_bounds__(NTH, 2) void conv_split_kernel(const ConvKArgs a) {
    using C = SCfg<MT, NT, PL, KS, MODE, F16>;
    static_assert(!F16 || PL == 2, "the fp16 form has two planes");
    constexpr int HP = C::HP, BN = C::BN, HWD = C::HWD, NTAP = C::NTAP, NSTEP = C::NSTEP;
    static_assert(MODE != 1 || KS == 2, "MODE 1: a stage = the two column positions of a block");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem);              // [KS][PL][HP]
    f32x4* sB = sA + KS * PL * C::A_SLOTS;                   // [KS][PL][10][BN]
    float* sM = reinterpret_cast<float*>(sB + KS * PL * C::B_SLOTS);   // F16: the waves' largest |x| of the stage being staged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;

    // XCD-aware work mapping (as conv_wino.hip): the channel tiles of one pixel tile run back to back on one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int bt = (slot / a.ncot) * 8 + xcd;
    if (bt >= a.tilesX * a.tilesY * a.N) return;
    const int n0 = (slot % a.ncot) * BN;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int n = bt / a.tilesY;
    const int oy0 = ty * C::TH, ox0 = tx * TW;
    SPLIT_STAMP(0);
#ifdef REFID_SPLIT_TRACE
    const bool ktrace = g_split_ktrace && (int)blockIdx.x == g_split_ktrace_wg;
    if (g_split_trace && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_split_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    // ---- loaders: per-thread byte offsets that never change across chunks ----------------------------------
    const int limA = (int)min((long long)a.N * a.H * a.W * a.ldA * 4, 0x7fffffffLL);
    const int limB = a.inB ? (int)min((long long)a.N * a.H * a.W * a.ldB * 4, 0x7fffffffLL) : 0;
    const int wChunk = PL * NTAP * a.CoutPad * 16;           // bytes of packed weights per 8-channel sub-chunk
    const int nsub = cdiv_dev(a.Ctot, KC) * (MODE == 1 ? 4 : 1);      // sub-chunks of the whole K
    const int cls = (MODE == 2) ? blockIdx.y : 0;            // MODE 2: output parity class (py, px)
    const int py = cls >> 1, px = cls & 1;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<float*>(a.w)) + (F16 ? F16_HEADER : 0) + (long long)cls * nsub * wChunk, 0,
        (int)min((long long)nsub * wChunk, 0x7fffffffLL), 0x00020000);
    const int eW = F16 ? *reinterpret_cast<const int*>(a.w) : 0;       // the weights travel as w 2^eW
    int eRef = F16_E0;                                       // F16: the workgroup's reference exponent (biased)
    // MODE 0 / 2: voA / voB[it] = the halo pixel of item `it` in source A / B.  MODE 1: voA[it] / voB[it] = the pixel at
    // ROW position sy = 0 / 1 of halo block `it`, column position 0; vo1x[sy][it] = column position 1 (validity differs)
    int voA[C::A_ITEMS], voB[C::A_ITEMS], vo1x[2][MODE == 1 ? C::A_ITEMS : 1];
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int hp = tid + it * NTH;
        if constexpr (MODE == 1) {
            const int r0 = 2 * (oy0 + hp / HWD) - 1, c0 = 2 * (ox0 + hp % HWD) - 1;
#pragma unroll
            for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const int iy = r0 + sy, ix = c0 + sx;
                    const bool ok = hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    const int vo = ok ? ((n * a.H + iy) * a.W + ix) * a.ldA * 4 : OOB;
                    if (sx == 0) (sy ? voB[it] : voA[it]) = vo;
                    else vo1x[sy][it] = vo;
                }
        } else {
            const int iy = oy0 - 1 + hp / HWD, ix = ox0 - 1 + hp % HWD;
            const bool ok = hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int pix = (n * a.H + iy) * a.W + ix;
            voA[it] = ok ? pix * a.ldA * 4 : OOB;
            voB[it] = ok ? pix * a.ldB * 4 : OOB;
        }
    }
    int voW[C::B_ITEMS];
#pragma unroll
    for (int it = 0; it < C::B_ITEMS; ++it) {
        const int s = tid + it * NTH;                        // LDS slot [plane][tap][cout]
        const int co = s % BN, r = s / BN;                   // r = plane*10 + tap
        const int row = a.coBase + n0 + co;
        voW[it] = (s < PL * C::B_SLOTS && row < a.CoutPad) ? (r * a.CoutPad + row) * 16 : OOB;
    }

    f32x4 ra[KS][C::A_ITEMS][2], rb[KS][C::B_ITEMS];
    // the KS * (2*A_ITEMS + B_ITEMS) loads of a stage are issued in NSTEP*KS parts, one per MFMA step of the previous stage
    // (a wave that issues them back to back sits in the vector-memory issue queue: tools/probes/split_trace.py)
    constexpr int NLD = 2 * C::A_ITEMS + C::B_ITEMS;
    auto load_part = [&](int chReal, int part) {
        const int ch = REFID_SPLIT_ABLATE == 3 ? 0 : chReal;
#pragma unroll
        for (int sub = 0; sub < KS; ++sub) {
            // MODE 1: stage ch = (8-channel group ch >> 1, row position ch & 1), sub = column position
            const int c0 = (MODE == 1 ? (ch >> 1) : (ch * KS + sub)) * KC;   // sub-chunk-uniform source (Ca % 8 == 0)
            const bool fromA = MODE != 0 || c0 < a.Ca;
            const bool cok = c0 < a.Ctot;                    // past the last channel: zeros (weights: buffer range check)
            const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
            const bool sy1 = MODE == 1 && (ch & 1);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(fromA ? a.inA : a.inB), 0, fromA ? limA : limB, 0x00020000);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if ((sub * NLD + i) * NSTEP / NLD != part && part >= 0) continue;
                if (i < 2 * C::A_ITEMS) {
                    const int it = i >> 1;
                    int vo;
                    if constexpr (MODE == 1) vo = sub == 0 ? (sy1 ? voB[it] : voA[it]) : (sy1 ? vo1x[1][it] : vo1x[0][it]);
                    else vo = fromA ? voA[it] : voB[it];
                    if (!cok) vo = OOB;
                    ra[sub][it][i & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff + 16 * (i & 1), 0));
                } else {
                    const int it = i - 2 * C::A_ITEMS;
                    // (the hardware range check covers the vector offset only: a sub-chunk past K -- Ctot % 16 == 8 with
                    //  two sub-chunks per stage -- must not travel as a scalar offset, it would read past the class's weights)
                    //  forced out of range with an OR: a select here becomes a branch around the load + a full wait)
                    //  Only MODE 2 can get there (MODE 0 stages one sub-chunk, MODE 1's four sub-chunks always exist).
                    int vo = voW[it];
                    if constexpr (MODE == 2) vo |= (ch * KS + sub < nsub) ? 0 : OOB;
                    rb[sub][it] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, (ch * KS + sub) * wChunk, 0));
                }
            }
        }
    };
    auto load_chunk = [&](int ch) { load_part(ch, -1); };
    // F16: this thread's / wave's largest |x| of the stage in `ra` -> the wave's LDS slot (read by everybody after the barrier)
    auto post_max = [&]() {
        float tm = 0.f;
#pragma unroll
        for (int sub = 0; sub < KS; ++sub)
#pragma unroll
            for (int it = 0; it < C::A_ITEMS; ++it)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = ra[sub][it][h];
                    tm = fmaxf(fmaxf(tm, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o));
        if (lane == 0) sM[wave] = tm;
    };
    f32x16 acc[MT][NT];
    // after the barrier: the stage's scale 2^(TGT - (eRef - 127)); the accumulators follow when the data outgrow the reference
    auto stage_scale = [&]() -> float {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(sM);
        const int eb = (int)(__float_as_uint(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]))) >> 23);
        if (eb > eRef + F16_SLACK) {                         // workgroup-uniform, rare
            const int fe = 127 + eRef - eb;
            const float f = fe >= 1 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] *= f;
            eRef = eb;
        }
        return __uint_as_float((unsigned)(F16_TGT + 254 - eRef) << 23);
    };
    auto store_chunk = [&](float sc = 1.f) {
#pragma unroll
        for (int sub = 0; sub < KS; ++sub) {
#pragma unroll
            for (int it = 0; it < C::A_ITEMS; ++it) {
                const int hp = tid + it * NTH;
                f32x4 pl[PL];
                if (REFID_SPLIT_ABLATE == 2) { pl[0] = ra[sub][it][0]; if constexpr (PL >= 2) pl[1] = ra[sub][it][1]; if constexpr (PL == 3) pl[2] = ra[sub][it][0]; }
                else if constexpr (F16) split8h(ra[sub][it][0] * sc, ra[sub][it][1] * sc, pl);
                else split8<PL>(ra[sub][it][0], ra[sub][it][1], pl);
                if (hp < HP) {
#pragma unroll
                    for (int p = 0; p < PL; ++p) sA[(sub * PL + p) * HP + hp] = pl[p];
                }
            }
#pragma unroll
            for (int it = 0; it < C::B_ITEMS; ++it) {
                const int s = tid + it * NTH;
                if (s < PL * C::B_SLOTS) sB[sub * PL * C::B_SLOTS + s] = rb[sub][it];
            }
        }
    };

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // products kept: (activation plane, weight plane), largest first
    constexpr int TERMS = (PL == 3) ? 6 : (PL == 2 ? 3 : 1);
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};

    // MFMA step j of a sub-chunk: K = {tap 2j (lanes 0-31), tap 2j+1 (lanes 32-63)} x 8 channels
    int aoff[NSTEP];
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
        if (MODE == 0) {
            const int t = min(2 * j + kh, 8);                // the tenth tap has zero weights: any valid address
            aoff[j] = (wave * MT + t / 3) * HWD + (t % 3) + li;
        } else if (MODE == 1) {                              // block taps (ty = j, tx = kh)
            aoff[j] = (wave * MT + j) * HWD + kh + li;
        } else {                                             // tap (ta = j, tb = kh) of class (py, px): conv_igemm.hip mode 2
            const int dy = (j == 0) ? 1 : (py ? 2 : 0);
            const int dx = (kh == 0) ? 1 : (px ? 2 : 0);
            aoff[j] = (wave * MT + dy) * HWD + dx + li;
        }
    }
    const f32x4* pB = sB + kh * BN + li;

    load_chunk(0);
    if constexpr (F16) {
        post_max();
        __syncthreads();
        store_chunk(stage_scale());
    } else {
        store_chunk();
    }
    __syncthreads();
    SPLIT_STAMP(1);

    for (int ch = 0; ch < ((REFID_SPLIT_ABLATE == 5 || REFID_SPLIT_ABLATE == 7) ? 0 : a.nchunks); ++ch) {
        const bool more = ch + 1 < a.nchunks;
        SPLIT_KSTAMP(ch, 0);
        SPLIT_KSTAMP(ch, 1);
#pragma unroll
        for (int sj = 0; sj < NSTEP * KS; ++sj) {
            const int sub = sj / NSTEP, j = sj % NSTEP;
            if (more && REFID_SPLIT_ABLATE != 4) load_part(ch + 1, sj);
            f32x4 af[PL][MT], bf[PL][NT];
#pragma unroll
            for (int p = 0; p < PL; ++p) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[p][m] = REFID_SPLIT_ABLATE == 8 ? ra[0][0][m & 1] : sA[(sub * PL + p) * C::A_SLOTS + m * HWD + aoff[j]];
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    bf[p][nn] = REFID_SPLIT_ABLATE == 8 ? rb[0][0] : pB[((sub * PL + p) * NTAP + 2 * j) * BN + nn * 32];
            }
#pragma unroll
            for (int e = 0; e < (REFID_SPLIT_ABLATE == 1 ? 0 : TERMS); ++e)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) {    // consecutive MFMAs hit different accumulators
                        if constexpr (F16)
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(f16x8, bf[TB[e]][nn]), __builtin_bit_cast(f16x8, af[TA[e]][m]),
                                acc[m][nn], 0, 0, 0);
                        else
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, bf[TB[e]][nn]), __builtin_bit_cast(bf16x8, af[TA[e]][m]),
                                acc[m][nn], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);           // keep the step's loads with the step
        }
        SPLIT_KSTAMP(ch, 2);
        if constexpr (F16) {
            if (more) post_max();                    // (the next stage's loads have been in flight since this stage's first step)
        }
        __syncthreads();
        SPLIT_KSTAMP(ch, 3);
#ifdef REFID_SPLIT_TRACE
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): separate the load wait from the store phase in the trace
#endif
        SPLIT_KSTAMP(ch, 4);
        if (more) {
            if constexpr (F16) store_chunk(stage_scale());
            else store_chunk();
            SPLIT_KSTAMP(ch, 5);
            __syncthreads();
        }
        SPLIT_KSTAMP(ch, 6);
    }
    SPLIT_STAMP(2);

    // ---- fused epilogue, coalesced ------------------------------------------------------------------------------
    // D[cout][pixel]: lane (li = pixel column, kh) holds, per register quad g, output channels 8g + 4kh + {0..3} of
    // pixel li.  Each wave writes one of its pixel rows into its private LDS strip ([pixel][channel quad], pitch
    // C4 + 1 slots: conflict free both ways) and reads it back as thread -> (pixel, channel quad) in memory order.
    constexpr int C4 = C::C4, XS = C::XS;
    constexpr int EIT = (32 * C4) / 64;
    auto opix = [&](int oy, int ox) -> long long {           // pixel index in the output tensor
        if (MODE == 2) return (long long)(n * 2 * a.Ho + 2 * oy + py) * (2 * a.Wo) + 2 * ox + px;
        return (long long)(n * a.Ho + oy) * a.Wo + ox;
    };
    f32x4* ex = reinterpret_cast<f32x4*>(smem) + wave * (32 * XS);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + wave * MT + m;
        // residual / mask of this row are requested before the turn-around so their latency hides behind it
        f32x4 pres[EIT], pmask[EIT];
        const bool pre = a.vecOK && (a.res != nullptr || a.mask != nullptr) && REFID_SPLIT_ABLATE != 6 && REFID_SPLIT_ABLATE != 7;
        if (pre) {
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int f = it * 64 + lane;
                const int c4 = f % C4, px = f / C4;
                const int ox = ox0 + px, j0 = n0 + c4 * 4;
                const bool ok = oy < a.Ho && ox < a.Wo && j0 + 3 < a.Cout;
                const long long op = opix(oy, ox);
                pres[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                pmask[it] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (ok && a.res) pres[it] = *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
                if (ok && a.mask) pmask[it] = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
            }
        }
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[m][nn][4 * g + k];
                    if constexpr (F16) v[k] = ldexpf(v[k], eRef - 127 - F16_TGT - eW);   // undo both scales: exact
                }
                ex[li * XS + nn * 8 + 2 * g + kh] = v;
            }
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int f = it * 64 + lane;
            const int c4 = f % C4, px = f / C4;
            const int ox = ox0 + px, j0 = n0 + c4 * 4;
            f32x4 v = ex[px * XS + c4];
            if (oy >= a.Ho || ox >= a.Wo || j0 >= a.Cout) continue;
            const long long op = opix(oy, ox);
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
            if (REFID_SPLIT_ABLATE == 6 || REFID_SPLIT_ABLATE == 7) {
                if (v[0] == 12345.678f) a.out[op * a.ldO + j0] = v[1];
                continue;
            }
            if (vec) {
                if (a.res) v += pres[it];
                lrelu4(v, a.slopePost, a.slopePost != 1.f);
                if (a.mask) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] *= (pmask[it][k] > 0.f) ? 1.f : a.slopeMask;
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
    }
    SPLIT_STAMP(3);
}

template <int MT, int NT, int PL, int KS, int MODE, bool F16 = false>
int launch_split(const ConvKArgs& ka, hipStream_t st) {
    using C = SCfg<MT, NT, PL, KS, MODE, F16>;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = refid_lds_attr_once(attr_done, &conv_split_kernel<MT, NT, PL, KS, MODE, F16>, C::LDS_BYTES, "conv_split")) return rc;
    ConvKArgs a = ka;
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, C::TH);
    a.nchunks = (MODE == 1) ? 2 * cdiv(a.Ctot, KC) : cdiv(a.Ctot, KC * KS);     // stages
    a.ncot = cdiv(a.Cout, C::BN);
    const int tiles = a.tilesX * a.tilesY * a.N;
    dim3 grid(cdiv(tiles, 8) * 8 * a.ncot, MODE == 2 ? 4 : 1);
    hipLaunchKernelGGL((conv_split_kernel<MT, NT, PL, KS, MODE, F16>), grid, dim3(NTH), C::LDS_BYTES, st, a);
    REFID_LAUNCH_CHECK("conv_split");
    return 0;
}

// stage depth of the 3x3 tile KS = 1: deeper stages (2 sub-chunks for two planes, 4 for one: fewer barriers, same LDS as
// three planes) measured 4-8 % SLOWER on every shape -- the barrier pair is not what the short stages wait for.  The
// 2x2-tap modes have two MFMA steps per sub-chunk and always stage two.
template <int PL, int MODE, bool F16 = false>
int launch_split_pl(const ConvKArgs& a, bool wide, bool tall, hipStream_t st) {
    constexpr int KS = (MODE == 0) ? 1 : 2;
    if (wide) return tall ? launch_split<2, 2, PL, KS, MODE, F16>(a, st) : launch_split<1, 2, PL, KS, MODE, F16>(a, st);
    return tall ? launch_split<2, 1, PL, KS, MODE, F16>(a, st) : launch_split<1, 1, PL, KS, MODE, F16>(a, st);
}

}  // namespace

#ifdef REFID_SPLIT_TRACE
extern "C" int refid_split_trace_set(unsigned long long* wg_stamps, unsigned long long* k_stamps, int wg) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_trace), &wg_stamps, sizeof(wg_stamps)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_ktrace), &k_stamps, sizeof(k_stamps)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_split_ktrace_wg), &wg, sizeof(wg)) != hipSuccess;
}
#endif

bool refid_split3x3_eligible(const ConvKArgs& a) {
    const long long lim = 0x7fffffffLL;
    return a.Ctot % 8 == 0 && (a.inB == nullptr || a.Ca % KC == 0) &&
           (long long)a.N * a.H * a.W * a.ldA * 4 < lim && (!a.inB || (long long)a.N * a.H * a.W * a.ldB * 4 < lim) &&
           a.ldA % 4 == 0 && (!a.inB || a.ldB % 4 == 0);
}

// terms: 6 (three planes per operand: fp32-class products), 3 (two planes: 2^-16 relative), 1 (plain bf16 operands) or 19 (two
// fp16 planes, three products on scaled operands: the fp32 class at half of 6's MFMAs; the stride-2 modes only)
// mode: 0 = 3x3 stride 1; 1 = 4x4 stride 2 pad 1 forward; 2 = its input gradient (a.Ho / a.Wo = the gradient's grid)
int refid_launch_split3x3(const ConvKArgs& a, int terms, int mode, int cus, int split_mode, hipStream_t st) {
    REFID_CHECK(terms == 1 || terms == 3 || terms == 6 || (terms == 19 && mode != 0),
                "conv2d: split tile takes 1, 3 or 6 product terms, or 19 (three fp16 products; 4x4 stride 2 and its input "
                "gradient only) (got %d)", terms);
    REFID_CHECK(terms != 19 || (reinterpret_cast<uintptr_t>(a.w) & 15) == 0, "conv2d: the fp16 split packing must be 16-byte aligned");
    REFID_CHECK(refid_split3x3_eligible(a) && (mode == 0 || a.inB == nullptr),
                "conv2d: split tile needs channel counts that are multiples of 8, tensors below 2 GiB and, for the stride-2 "
                "modes, a single source");
    {   // the weight fragments are fetched with 32-bit buffer offsets as well
        const long long planes = terms == 6 ? 3 : ((terms == 3 || terms == 19) ? 2 : 1);
        const long long wbytes = (long long)cdiv(a.Ctot, KC) * (mode == 1 ? 4 : 1) * planes * (mode == 0 ? 10 : 4) * a.CoutPad * 16;
        REFID_CHECK(wbytes < 0x7fffffffLL, "conv2d: packed weights too large for the split tile's 32-bit offsets");
    }
    const bool wide = a.Cout > 32;
    // 8-row tiles when they still give every CU its two workgroups, 4-row tiles otherwise
    // (split policy 1, "sample": decided as if the batch held 8 samples, so that a sample's bits do not depend on the batch it
    //  is in -- with the fp16 form the workgroup's tile decides its scale, and the scale the rounding of values whose low plane
    //  is subnormal; the exact-split bf16 forms give the same bits on either tile)
    const int wg8 = cdiv(a.Wo, TW) * cdiv(a.Ho, 8) * (split_mode == 1 ? 8 : a.N) * cdiv(a.Cout, wide ? 64 : 32) * (mode == 2 ? 4 : 1);
    const bool tall = wg8 >= 2 * cus;
#define SPLIT_DISPATCH(PLN)                                                          \
    (mode == 0 ? launch_split_pl<PLN, 0>(a, wide, tall, st)                          \
               : (mode == 1 ? launch_split_pl<PLN, 1>(a, wide, tall, st) : launch_split_pl<PLN, 2>(a, wide, tall, st)))
    if (terms == 19) return mode == 1 ? launch_split_pl<2, 1, true>(a, wide, tall, st) : launch_split_pl<2, 2, true>(a, wide, tall, st);
    if (terms == 6) return SPLIT_DISPATCH(3);
    if (terms == 3) return SPLIT_DISPATCH(2);
    return SPLIT_DISPATCH(1);
#undef SPLIT_DISPATCH
}
