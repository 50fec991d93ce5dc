// Fused implicit-GEMM convolution tile on the gfx950 fp32 matrix cores.
//
//   out = mask( post( pre(conv(src) + bias) + res ) ),   src = in_a or [in_a | in_b]
//
// One kernel template serves every dense convolution on the REFID hot path, forward and
// input-gradient:
//   mode 0  KxK conv, stride 1/2          (3x3, 5x5, 1x1, 4x4s2 `conv_down`, 2x2s2 = convT dgrad)
//   mode 1  ConvTranspose2d(2,2) forward  (1x1 GEMM, 4*Co columns, pixel-shuffle store)
//   mode 2  conv_down input gradient      (4 output-parity classes of 2x2 taps)
//
// Mapping to the hardware (MI355X, CDNA4):
//   * GEMM view: M = output pixels, N = output channels, K = taps x input channels.
//     D[pixel][cout] is accumulated with v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//   * A workgroup (256 threads = 4 waves, one per SIMD) owns TH x 32 output pixels x BN
//     output channels; a wave owns MT x NT tiles of 32 pixels x 32 channels (MT*NT*16
//     accumulator VGPRs).
//   * K is walked in chunks of KC = 8*NSUB input channels.  Per chunk the input HALO tile
//     ((TH-1)*S+KH) x (31*S+KW) pixels x KC channels is staged once in LDS and re-used by
//     all KH*KW taps (the 9x / 25x / 16x re-use never touches L2/HBM); the weight tile
//     [tap][BN][KC] is staged next to it.
//   * LDS image is [channel-quad][pixel][4 floats]: lane (i = l&31, kh = l>>5) fetches its
//     four K values of pixel i with ONE ds_read_b128 from plane (2*sub+kh); the four
//     MFMAs that follow consume k-pairs (c, c+4).  Consecutive pixels are consecutive
//     16-byte slots, so the b128 lane groups are conflict-free.
//   * The next chunk is prefetched global->VGPR while the current chunk's MFMAs run and is
//     written to LDS after them (register double buffering, 2 barriers per chunk); several
//     workgroups per CU cover each other's barrier bubbles.
//   * NHWC activations: a halo pixel's KC channels are one contiguous 32..128-byte piece.
//   * Epilogue is fused: bias, LeakyReLU, residual add, second LeakyReLU, activation-
//     derivative mask.  The accumulator is D[cout][pixel] (weights are the MFMA "A" operand), so
//     a lane owns 4 consecutive channels of one pixel per register quad: every epilogue tensor is
//     touched with 16-byte accesses.
#include "common.h"
#include "conv_args.h"

namespace {

template <int KH_, int KW_, int S_, int WM_, int WN_, int MT_, int NT_, int NSUB_, int MODE_>
struct Cfg {
    static constexpr int KH = KH_, KW = KW_, S = S_, WM = WM_, WN = WN_, MT = MT_, NT = NT_;
    static constexpr int NSUB = NSUB_, MODE = MODE_;
    static constexpr int TW = 32;
    static constexpr int TH = WM * MT;
    static constexpr int BN = WN * NT * 32;
    static constexpr int NTAPS = (MODE == 2) ? 4 : KH * KW;
    static constexpr int HH = (MODE == 2) ? TH + 2 : (TH - 1) * S + KH;
    static constexpr int HWD = (MODE == 2) ? TW + 2 : (TW - 1) * S + KW;
    static constexpr int HP = HH * HWD;
    static constexpr int NQ = NSUB * 2;          // channel quads per chunk
    static constexpr int KC = NSUB * 8;
    static constexpr int A_TOTAL = HP * NQ;
    static constexpr int B_TOTAL = NTAPS * BN * NQ;
    static constexpr int A_ITEMS = (A_TOTAL + 255) / 256;
    static constexpr int B_ITEMS = (B_TOTAL + 255) / 256;
    static constexpr int LDS_BYTES = (A_TOTAL + B_TOTAL) * 16;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(256 % NQ == 0, "thread -> channel-quad mapping must be static");
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// BF = true: bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulate / epilogue / tensors.
// A 16-byte LDS slot then holds 8 bf16 channels instead of 4 fp32 ones, so a K chunk is 16*NSUB
// channels; activations are converted while they are staged (v_cvt_pk_bf16_f32, RNE), the packed
// weights are already bf16.  Everything else (halo reuse, two-source input, fused epilogue) is shared.
template <class C, bool BF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKArgs a) {
    constexpr int CPQ = BF ? 8 : 4;                 // channels per 16-byte slot
    constexpr int KCC = C::NQ * CPQ;                // channels per K chunk
    constexpr int AV = BF ? 2 : 1;                  // float4 global loads per activation slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem);
    f32x4* sB = sA + C::A_TOTAL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int li = lane & 31, kh = lane >> 5;

    // split-K (small grids): blockIdx.x = tile * ksplit + part; part reduces K chunks [kc0, kc1) into a raw partial output
    const int kpart = blockIdx.x % a.ksplit;
    const int kper = (a.nchunks + a.ksplit - 1) / a.ksplit;
    const int kc0 = kpart * kper, kc1 = min(a.nchunks, kc0 + kper);
    int bt = blockIdx.x / a.ksplit;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int n = bt / a.tilesY;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int n0 = blockIdx.y * C::BN;
    const int cls = (C::MODE == 2) ? blockIdx.z : 0;
    const int py = cls >> 1, px = cls & 1;
    const int iy0 = (C::MODE == 2) ? oy0 - 1 : oy0 * C::S - a.pad;
    const int ix0 = (C::MODE == 2) ? ox0 - 1 : ox0 * C::S - a.pad;

    // ---- loader set-up (pixel offsets do not change across K chunks) --------------------
    const int q = tid % C::NQ;                  // this thread's channel quad within a chunk
    long long apix[C::A_ITEMS];                 // pixel index or -1
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int hp = tid / C::NQ + it * (256 / C::NQ);
        const int hy = hp / C::HWD, hx = hp % C::HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = (hp < C::HP) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        apix[it] = ok ? ((long long)(n * a.H + iy) * a.W + ix) : -1;
    }
    const float* wbase = reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(a.w) + (long long)cls * a.wClsStride * (BF ? 2 : 4));

    f32x4 ra[C::A_ITEMS * AV], rb[C::B_ITEMS];

    auto load_chunk = [&](int ch) {
        const int c = ch * KCC + q * CPQ;
        const bool fromA = c < a.Ca;
        const float* src = fromA ? a.inA : a.inB;
        const int ld = fromA ? a.ldA : a.ldB;
        const int cc = fromA ? c : c - a.Ca;
#pragma unroll
        for (int it = 0; it < C::A_ITEMS; ++it) {
#pragma unroll
            for (int v4 = 0; v4 < AV; ++v4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c + 4 * v4 < a.Ctot && apix[it] >= 0)
                    v = *reinterpret_cast<const f32x4*>(src + apix[it] * ld + cc + 4 * v4);
                ra[it * AV + v4] = v;
            }
        }
        // packed weights: [chunk][tap][CoutPad][KCC] in fp32 (BF: bf16) -> a 16-byte slot per (row, q)
        const char* wc = reinterpret_cast<const char*>(wbase) +
                         (long long)ch * C::NTAPS * a.CoutPad * C::NQ * 16;
#pragma unroll
        for (int it = 0; it < C::B_ITEMS; ++it) {
            const int r = tid / C::NQ + it * (256 / C::NQ);
            const int co = r % C::BN, tap = r / C::BN;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < C::NTAPS * C::BN && a.coBase + n0 + co < a.CoutPad)
                v = *reinterpret_cast<const f32x4*>(
                    wc + (((long long)(tap * a.CoutPad + a.coBase + n0 + co)) * C::NQ + q) * 16);
            rb[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < C::A_ITEMS; ++it) {
            const int hp = tid / C::NQ + it * (256 / C::NQ);
            if (hp < C::HP) {
                if constexpr (BF) {
                    bf16x8 hv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { hv[k] = (__bf16)ra[it * 2][k]; hv[4 + k] = (__bf16)ra[it * 2 + 1][k]; }
                    sA[q * C::HP + hp] = __builtin_bit_cast(f32x4, hv);
                } else {
                    sA[q * C::HP + hp] = ra[it];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < C::B_ITEMS; ++it) {
            const int r = tid / C::NQ + it * (256 / C::NQ);
            const int co = r % C::BN, tap = r / C::BN;
            if (r < C::NTAPS * C::BN) sB[(q * C::NTAPS + tap) * C::BN + co] = rb[it];
        }
    };

    // ---- accumulators ---------------------------------------------------------------------
    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int nn = 0; nn < C::NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    constexpr int HS = (C::MODE == 2) ? 1 : C::S;       // halo stride per output pixel
    int hpb[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m) hpb[m] = ((wm * C::MT + m) * HS) * C::HWD + li * HS;

    int tapoff2[4];
    if (C::MODE == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ta = t >> 1, tb = t & 1;
            const int dy = (ta == 0) ? 1 : (py ? 2 : 0);
            const int dx = (tb == 0) ? 1 : (px ? 2 : 0);
            tapoff2[t] = dy * C::HWD + dx;
        }
    }
    const int bcol = wn * C::NT * 32 + li;

    if (kc0 < kc1) {
        load_chunk(kc0);
        store_chunk();
    }
    __syncthreads();

    for (int ch = kc0; ch < kc1; ++ch) {
        const bool more = ch + 1 < kc1;
        if (more) load_chunk(ch + 1);

#pragma unroll
        for (int sub = 0; sub < C::NSUB; ++sub) {
            const f32x4* pA = sA + (sub * 2 + kh) * C::HP;
            const f32x4* pB = sB + (sub * 2 + kh) * C::NTAPS * C::BN + bcol;
#pragma unroll
            for (int t = 0; t < C::NTAPS; ++t) {
                const int toff = (C::MODE == 2) ? tapoff2[t] : (t / C::KW) * C::HWD + (t % C::KW);
                f32x4 af[C::MT], bf[C::NT];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) af[m] = pA[hpb[m] + toff];
#pragma unroll
                for (int nn = 0; nn < C::NT; ++nn) bf[nn] = pB[t * C::BN + nn * 32];
                if constexpr (BF) {
#pragma unroll
                    for (int m = 0; m < C::MT; ++m)
#pragma unroll
                        for (int nn = 0; nn < C::NT; ++nn)
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, bf[nn]), __builtin_bit_cast(bf16x8, af[m]), acc[m][nn], 0, 0, 0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int m = 0; m < C::MT; ++m)
#pragma unroll
                            for (int nn = 0; nn < C::NT; ++nn)
                                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nn][kk], af[m][kk],
                                                                                 acc[m][nn], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- fused epilogue -------------------------------------------------------------------
    // D[cout][pixel]: lane (li = pixel column, kh) holds, per register quad g, the four
    // consecutive output channels 8g + 4kh + {0..3} of pixel li -> one 16-byte access per
    // tensor (bias / residual / mask / out) instead of four 4-byte ones.
    const int Co1 = (C::MODE == 1) ? (a.Cout >> 2) : a.Cout;   // real channels (mode 1)
    const int OW = (C::MODE == 0) ? a.Wo : 2 * a.Wo;
    const int OH = (C::MODE == 0) ? a.Ho : 2 * a.Ho;
    const int ox = ox0 + li;
#pragma unroll
    for (int nn = 0; nn < C::NT; ++nn) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = n0 + wn * C::NT * 32 + nn * 32 + 8 * g + 4 * kh;   // first GEMM column of the quad
            if (j0 >= a.Cout) continue;
            int ch = j0, sy = 0, sx = 0;
            if (C::MODE == 1) { const int qd = j0 / Co1; ch = j0 - qd * Co1; sy = qd >> 1; sx = qd & 1; }
            if (C::MODE == 2) { sy = py; sx = px; }
            const bool vec = a.vecOK && (j0 + 3 < a.Cout);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                const float* bp = a.bias + ((C::MODE == 1) ? ch : a.coBase + j0);
                if (vec) bv = *reinterpret_cast<const f32x4*>(bp);
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (j0 + k < a.Cout) bv[k] = bp[k];
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
                const int oy = oy0 + wm * C::MT + m;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                long long op;
                if (C::MODE == 0) op = (long long)(n * OH + oy) * OW + ox;
                else op = (long long)(n * OH + 2 * oy + sy) * OW + 2 * ox + sx;
                f32x4 v;
                if (a.ksplit > 1) {          // raw partial sums (workspace in the OUTPUT's pixel / channel coordinates)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[m][nn][4 * g + k];
                    *reinterpret_cast<f32x4*>(a.out + kpart * a.wsStride + op * a.ldO + ch) = v;
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lrelu(acc[m][nn][4 * g + k] + bv[k], a.slopePre);
                if (vec) {
                    if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + ch);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], a.slopePost);
                    if (a.mask) {
                        const f32x4 mv = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + ch);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
                    }
                    *reinterpret_cast<f32x4*>(a.out + op * a.ldO + ch) = v;
                    if (a.out2)
                        *reinterpret_cast<f32x4*>(a.out2 + op * a.ldO2 + ch) = v + *reinterpret_cast<const f32x4*>(a.add2 + op * a.ldA2 + ch);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (j0 + k >= a.Cout) break;
                        float t = v[k];
                        if (a.res) t += a.res[op * a.ldR + ch + k];
                        t = lrelu(t, a.slopePost);
                        if (a.mask) t *= (a.mask[op * a.ldM + ch + k] > 0.f) ? 1.f : a.slopeMask;
                        a.out[op * a.ldO + ch + k] = t;
                        if (a.out2) a.out2[op * a.ldO2 + ch + k] = t + a.add2[op * a.ldA2 + ch + k];
                    }
                }
            }
        }
    }
}

// split-K plan of the direct tile: only the long-K / few-tile 4x4 stride-2 family (conv_down forward and its input
// gradient) uses it; same policy values as the Winograd tile (refid_conv_desc.split_k)
struct SplitArgs { float* ws; size_t ws_bytes; int mode; };

template <class C>
int igemm_ksplit(const ConvKArgs& a, int ncls, int split_mode, bool bf) {
    if (!split_mode || !(C::MODE == 0 || C::MODE == 2) || !a.vecOK || a.Cout % 4) return 1;
    const int tiles1 = cdiv(a.Wo, C::TW) * cdiv(a.Ho, C::TH);                  // per sample
    const int nwg = tiles1 * (split_mode == 2 ? a.N : 8) * cdiv(a.Cout, C::BN) * ncls;
    const int nchunks = cdiv(a.Ctot, C::KC * (bf ? 2 : 1));
    int ks = 1;
    if (nwg <= 128 && nchunks >= 8) {          // (one workgroup per CU already: splitting then only adds the finishing pass)
        ks = 512 / nwg;
        if (ks > nchunks / 4) ks = nchunks / 4;
        if (ks > 8) ks = 8;
        if (ks < 1) ks = 1;
    }
    return ks;
}

template <class C>
size_t igemm_ws_bytes(const ConvKArgs& a, int ncls, int split_mode, bool bf) {
    const int ks = igemm_ksplit<C>(a, ncls, split_mode, bf);
    if (ks == 1) return 0;
    const long long opix = (long long)a.N * a.Ho * a.Wo * (C::MODE == 2 ? 4 : 1);
    return (size_t)ks * opix * round_up(a.Cout, 4) * sizeof(float);
}

template <class C, bool BF>
int launch_t(const ConvKArgs& ka, int ncls, const SplitArgs& sp, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = refid_lds_attr_once(attr_done, &conv_igemm_kernel<C, BF>, C::LDS_BYTES, "conv")) return rc;
    ConvKArgs a = ka;
    a.tilesX = cdiv(a.Wo, C::TW);
    a.tilesY = cdiv(a.Ho, C::TH);
    a.nchunks = cdiv(a.Ctot, C::KC * (BF ? 2 : 1));
    const int ks = sp.ws ? igemm_ksplit<C>(a, ncls, sp.mode, BF) : 1;
    dim3 grid(a.tilesX * a.tilesY * a.N * ks, cdiv(a.Cout, C::BN), ncls);
    if (ks == 1) {
        hipLaunchKernelGGL((conv_igemm_kernel<C, BF>), grid, dim3(256), C::LDS_BYTES, st, a);
        REFID_LAUNCH_CHECK("conv_igemm");
        return 0;
    }
    const long long opix = (long long)a.N * a.Ho * a.Wo * (C::MODE == 2 ? 4 : 1);
    const int ldW = round_up(a.Cout, 4);
    const size_t need = (size_t)ks * opix * ldW * sizeof(float);
    if (need > sp.ws_bytes || (reinterpret_cast<uintptr_t>(sp.ws) & 15)) {
        refid_set_error("conv_igemm: split-K workspace too small or misaligned (%zu bytes given, %zu needed: "
                        "refid_conv_workspace_bytes)", sp.ws_bytes, need);
        return 1;
    }
    ConvKArgs p = a;                       // partial pass: raw sums into the workspace (output coordinates)
    p.ksplit = ks; p.wsStride = opix * ldW; p.out = sp.ws; p.ldO = ldW;
    hipLaunchKernelGGL((conv_igemm_kernel<C, BF>), grid, dim3(256), C::LDS_BYTES, st, p);
    REFID_LAUNCH_CHECK("conv_igemm/splitk");
    ConvKArgs f = a;
    f.ksplit = ks; f.wsStride = opix * ldW;
    return refid_launch_splitk_finish(f, sp.ws, ldW, opix, st);
}

template <class C>
int launch(const ConvKArgs& ka, int ncls, hipStream_t st, const SplitArgs& sp = SplitArgs{nullptr, 0, 0}) {
    return ka.bf16 ? launch_t<C, true>(ka, ncls, sp, st) : launch_t<C, false>(ka, ncls, sp, st);
}

// tile families: <KH,KW,S, WM,WN,MT,NT, NSUB, MODE>
using C3_64 = Cfg<3, 3, 1, 4, 1, 2, 2, 1, 0>;     // 8x32 px x 64 ch
using C3_128 = Cfg<3, 3, 1, 2, 2, 2, 2, 1, 0>;    // 4x32 px x 128 ch
using C3_32 = Cfg<3, 3, 1, 4, 1, 2, 1, 1, 0>;     // 8x32 px x 32 ch
using C5_32 = Cfg<5, 5, 1, 4, 1, 2, 1, 1, 0>;
using C1_64 = Cfg<1, 1, 1, 4, 1, 2, 2, 4, 0>;
using C1_128 = Cfg<1, 1, 1, 2, 2, 2, 2, 4, 0>;
using C1_32 = Cfg<1, 1, 1, 4, 1, 2, 1, 4, 0>;
using C4S2_64 = Cfg<4, 4, 2, 4, 1, 2, 2, 1, 0>;
using C2S2_64 = Cfg<2, 2, 2, 4, 1, 2, 2, 1, 0>;
using C2S2_128 = Cfg<2, 2, 2, 2, 2, 2, 2, 1, 0>;
using CT_128 = Cfg<1, 1, 1, 2, 2, 2, 2, 4, 1>;    // convT 2x2 forward
using CT_64 = Cfg<1, 1, 1, 4, 1, 2, 2, 4, 1>;
using CD_64 = Cfg<3, 3, 1, 4, 1, 2, 2, 2, 2>;     // conv_down dgrad
using CD_128 = Cfg<3, 3, 1, 2, 2, 2, 2, 2, 2>;

enum Family { F_3x3, F_5x5, F_1x1, F_4x4s2, F_2x2s2, F_convT, F_downDgrad, F_none };

Family family_of(int kh, int kw, int stride, int mode) {
    if (mode == 1) return F_convT;
    if (mode == 2) return F_downDgrad;
    if (kh == 3 && kw == 3 && stride == 1) return F_3x3;
    if (kh == 5 && kw == 5 && stride == 1) return F_5x5;
    if (kh == 1 && kw == 1 && stride == 1) return F_1x1;
    if (kh == 4 && kw == 4 && stride == 2) return F_4x4s2;
    if (kh == 2 && kw == 2 && stride == 2) return F_2x2s2;
    return F_none;
}

}  // namespace

extern "C" int refid_conv_kc(int kh, int kw, int stride, int mode) {
    switch (family_of(kh, kw, stride, mode)) {
        case F_3x3: case F_5x5: case F_4x4s2: case F_2x2s2: return 8;
        case F_1x1: case F_convT: return 32;
        case F_downDgrad: return 16;
        default: return -1;
    }
}

extern "C" int refid_conv_bn(int kh, int kw, int stride, int mode, int cout) {
    switch (family_of(kh, kw, stride, mode)) {
        case F_3x3: case F_1x1: return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128);
        case F_5x5: return 32;
        case F_4x4s2: return 64;
        case F_2x2s2: case F_downDgrad: case F_convT: return cout <= 64 ? 64 : 128;
        default: return -1;
    }
}

extern "C" const char* refid_conv_tile_name(int kh, int kw, int stride, int mode, int cout) {
    // template signature of the instantiation refid_conv2d launches (= the rocprof kernel name's Cfg<...>)
    const int bn = refid_conv_bn(kh, kw, stride, mode, cout);
    switch (family_of(kh, kw, stride, mode)) {
        case F_3x3: return bn == 32 ? "Cfg<3, 3, 1, 4, 1, 2, 1, 1, 0>" : bn == 64 ? "Cfg<3, 3, 1, 4, 1, 2, 2, 1, 0>"
                                                                                  : "Cfg<3, 3, 1, 2, 2, 2, 2, 1, 0>";
        case F_5x5: return "Cfg<5, 5, 1, 4, 1, 2, 1, 1, 0>";
        case F_1x1: return bn == 32 ? "Cfg<1, 1, 1, 4, 1, 2, 1, 4, 0>" : bn == 64 ? "Cfg<1, 1, 1, 4, 1, 2, 2, 4, 0>"
                                                                                  : "Cfg<1, 1, 1, 2, 2, 2, 2, 4, 0>";
        case F_4x4s2: return "Cfg<4, 4, 2, 4, 1, 2, 2, 1, 0>";
        case F_2x2s2: return bn == 64 ? "Cfg<2, 2, 2, 4, 1, 2, 2, 1, 0>" : "Cfg<2, 2, 2, 2, 2, 2, 2, 1, 0>";
        case F_convT: return bn == 64 ? "Cfg<1, 1, 1, 4, 1, 2, 2, 4, 1>" : "Cfg<1, 1, 1, 2, 2, 2, 2, 4, 1>";
        case F_downDgrad: return bn == 64 ? "Cfg<3, 3, 1, 4, 1, 2, 2, 2, 2>" : "Cfg<3, 3, 1, 2, 2, 2, 2, 2, 2>";
        default: return "none";
    }
}

extern "C" size_t refid_conv_workspace_bytes(const refid_conv_desc* d) {
    if (!d || d->split_k == 0) return 0;
    ConvKArgs a;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo;
    a.Cout = d->cout; a.CoutPad = d->cout_pad; a.coBase = d->co_base;
    a.vecOK = 1;
    if (d->algo == 1) return refid_wino3x3_workspace_bytes(a, d->split_k);
    if (d->algo == 5) return refid_wino6_workspace_bytes(a, d->split_k);
    if (d->algo != 0 && d->algo != 2) return 0;
    const bool bf = d->algo == 2;
    const int bn = refid_conv_bn(d->kh, d->kw, d->stride, d->mode, d->cout);
    switch (family_of(d->kh, d->kw, d->stride, d->mode)) {
        case F_4x4s2: return igemm_ws_bytes<C4S2_64>(a, 1, d->split_k, bf);
        case F_downDgrad: return bn == 64 ? igemm_ws_bytes<CD_64>(a, 4, d->split_k, bf) : igemm_ws_bytes<CD_128>(a, 4, d->split_k, bf);
        default: return 0;
    }
}

extern "C" int refid_conv2d(const refid_conv_desc* d, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REFID_CHECK(d != nullptr, "conv2d: null descriptor");
    const Family f = family_of(d->kh, d->kw, d->stride, d->mode);
    REFID_CHECK(f != F_none, "conv2d: unsupported geometry k=%dx%d stride=%d mode=%d", d->kh, d->kw,
                d->stride, d->mode);
    REFID_CHECK(d->in_a && d->w_packed && d->out, "conv2d: null tensor pointer");
    REFID_CHECK(d->c_a > 0 && d->c_a % 4 == 0 && d->c_b % 4 == 0 && d->c_b >= 0,
                "conv2d: channel counts must be multiples of 4 (c_a=%d c_b=%d)", d->c_a, d->c_b);
    REFID_CHECK(d->ld_a % 4 == 0 && (d->c_b == 0 || d->ld_b % 4 == 0),
                "conv2d: pixel pitches must be multiples of 4 floats");
    REFID_CHECK(d->c_b == 0 || d->in_b != nullptr, "conv2d: c_b > 0 but in_b is null");
    REFID_CHECK(d->n > 0 && d->h > 0 && d->w > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0,
                "conv2d: empty problem (n=%d h=%d w=%d ho=%d wo=%d cout=%d)", d->n, d->h, d->w, d->ho,
                d->wo, d->cout);
    REFID_CHECK(d->pw == nullptr || d->algo == 3, "conv2d: pw fusions belong to the pointwise tile (algo 3)");
    REFID_CHECK(d->algo == 0 || d->algo == 2 || ((d->algo == 1 || d->algo == 5) && f == F_3x3) || (d->algo == 3 && (f == F_1x1 || f == F_convT || f == F_2x2s2)) ||
                    (d->algo == 4 && (f == F_3x3 || f == F_4x4s2 || f == F_downDgrad)),
                "conv2d: algo %d does not fit this geometry (1 / 5 = 3x3 stride 1, 3 = 1x1 and ConvTranspose2d(2,2), 4 = 3x3 stride 1 / 4x4 stride 2 and "
                "its input gradient)", d->algo);
    REFID_CHECK(d->algo != 2 || d->c_b == 0 || d->c_a % 8 == 0, "conv2d: bf16 tile needs c_a %% 8 == 0 for two sources");
    const int bn = refid_conv_bn(d->kh, d->kw, d->stride, d->mode, d->cout);
    REFID_CHECK(d->co_base >= 0 && d->co_base + d->cout <= d->cout_pad,
                "conv2d: rows [%d, %d) exceed the packed weight's %d rows", d->co_base, d->co_base + d->cout,
                d->cout_pad);
    if (d->mode == 0) {
        const int eh = (d->h + 2 * d->pad - d->kh) / d->stride + 1;
        const int ew = (d->w + 2 * d->pad - d->kw) / d->stride + 1;
        REFID_CHECK(eh == d->ho && ew == d->wo, "conv2d: output size %dx%d does not match geometry (%dx%d)",
                    d->ho, d->wo, eh, ew);
    } else {
        REFID_CHECK(d->ho == d->h && d->wo == d->w, "conv2d: mode %d needs ho,wo == h,w", d->mode);
        REFID_CHECK(d->mode != 1 || d->cout % 4 == 0, "conv2d: mode 1 needs cout = 4*Co");
    }
    ConvKArgs a;
    a.inA = d->in_a; a.inB = d->in_b; a.ldA = d->ld_a; a.ldB = d->ld_b;
    a.Ca = d->c_a; a.Ctot = d->c_a + d->c_b;
    a.w = d->w_packed; a.bias = d->bias;
    a.out = d->out; a.ldO = d->ld_out;
    a.res = d->res; a.ldR = d->ld_res;
    a.mask = d->mask; a.ldM = d->ld_mask;
    a.add2 = d->add2; a.ldA2 = d->ld_add2; a.out2 = d->out2; a.ldO2 = d->ld_out2;
    a.maskMode = d->mask_mode;
    REFID_CHECK(d->mask_mode == 0 || (d->mask_mode == 1 && d->algo == 3 && d->mask != nullptr),
                "conv2d: mask_mode 1 (GELU derivative) belongs to the pointwise tile (algo 3) and needs a mask tensor");
    REFID_CHECK(d->out2 == nullptr || (d->add2 != nullptr && (d->algo != 3 || d->pw == nullptr)),
                "conv2d: out2 needs add2 (and, on the pointwise tile, no fusions)");
    a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo;
    a.Cout = d->cout; a.CoutPad = d->cout_pad; a.coBase = d->co_base;
    a.pad = d->pad; a.nchunks = 0; a.tilesX = a.tilesY = 0;
    a.slopePre = d->slope_pre; a.slopePost = d->slope_post; a.slopeMask = d->slope_mask;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    a.vecOK = al16(d->out) && d->ld_out % 4 == 0 && (!d->res || (al16(d->res) && d->ld_res % 4 == 0)) &&
              (!d->mask || (al16(d->mask) && d->ld_mask % 4 == 0)) && (!d->bias || al16(d->bias)) &&
              (!d->out2 || (al16(d->out2) && d->ld_out2 % 4 == 0 && al16(d->add2) && d->ld_add2 % 4 == 0)) &&
              d->co_base % 4 == 0 && (d->mode != 1 || (d->cout / 4) % 4 == 0);
    a.bf16 = (d->algo == 2);
    a.ncot = 0;
    const int kc = refid_conv_kc(d->kh, d->kw, d->stride, d->mode) * (a.bf16 ? 2 : 1);
    a.wClsStride = (long long)cdiv(a.Ctot, kc) * 4 * d->cout_pad * kc;   // mode 2 only
    if (d->algo == 3) {
        a.shuffle = (d->mode == 1) ? 1 : 0;                 // ConvTranspose2d(2,2) as a 1x1 GEMM over 4 Co columns + pixel shuffle
        REFID_CHECK(!a.shuffle || (d->pw == nullptr && d->mask == nullptr && d->out2 == nullptr && d->co_base == 0),
                    "conv2d: ConvTranspose2d on the pointwise tile takes no mask / second output / fusions / row range");
        if (f == F_2x2s2) {
            // non-overlapping 2x2 patches: output pixel (n, y, x) reads rows 2y and 2y+1, pixels 2x and 2x+1 -- two contiguous
            // runs of 2 c_a floats when pixels are dense: a two-source 1x1 GEMM with K = 4 c_a over row-pitched sources
            REFID_CHECK(d->c_b == 0 && d->ld_a == d->c_a && d->pad == 0 && d->h == 2 * d->ho && d->w == 2 * d->wo && d->pw == nullptr &&
                            (2 * d->c_a) % 8 == 0,
                        "conv2d: the 2x2 stride-2 patch GEMM (algo 3) needs one source with dense pixels (ld_a == c_a), pad 0, "
                        "even input sizes and c_a %% 4 == 0");
            a.inB = d->in_a + (long long)d->w * d->ld_a;       // the odd rows
            a.ldA = a.ldB = 2 * d->ld_a;
            a.Ca = 2 * d->c_a; a.Ctot = 4 * d->c_a;
            a.patchW = d->wo; a.patchRow = 2 * d->w * d->ld_a;
            a.H = d->ho; a.W = d->wo;                          // the GEMM's pixels
            REFID_CHECK((long long)d->n * d->h * d->w * d->ld_a * 4 < 0x7fffffffLL, "conv2d: tensor too large for the pointwise tile's 32-bit offsets");
            return refid_launch_pointwise(a, st, nullptr, 0);
        }
        REFID_CHECK(d->c_b == 0 || d->c_a % 8 == 0, "conv2d: pointwise tile needs c_a %% 8 == 0 for two sources");
        const long long lim = 0x7fffffffLL;
        REFID_CHECK((long long)d->n * d->h * d->w * d->ld_a * 4 < lim &&
                        (d->c_b == 0 || (long long)d->n * d->h * d->w * d->ld_b * 4 < lim),
                    "conv2d: tensor too large for the pointwise tile's 32-bit offsets (use algo 0)");
        if (d->pw != nullptr) {
            const refid_pw_extras* x = d->pw;
            PwExtra e;
            e.lnG = x->ln_gamma; e.lnB = x->ln_beta; e.lnEps = x->ln_eps; e.lnOut = x->ln_out; e.ldLn = x->ld_ln_out;
            e.pool = x->pool; e.poolParts = x->pool_parts; e.invHW = x->inv_hw; e.hw = x->hw; e.seC = x->se_c;
            e.seW1 = x->se_w1; e.seB1 = x->se_b1; e.seW2 = x->se_w2; e.seB2 = x->se_b2;
            e.seM = x->se_m; e.seZ1 = x->se_z1; e.seS = x->se_s;
            e.xsOut = x->xs_out; e.ldXs = x->ld_xs_out;
            e.res2 = x->res2; e.ldR2 = x->ld_res2;
            e.out2 = x->out2; e.ldO2 = x->ld_out2;
            auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0; };
            REFID_CHECK((!e.lnOut || al(e.lnOut, e.ldLn)) && (!e.xsOut || al(e.xsOut, e.ldXs)) && (!e.res2 || al(e.res2, e.ldR2)) &&
                            (!e.out2 || al(e.out2, e.ldO2)) && (!e.lnG || (al(e.lnG, 4) && al(e.lnB, 4))),
                        "conv2d: pointwise fusion tensors must be 16-byte aligned with pitches that are multiples of 4");
            return refid_launch_pointwise(a, st, &e, d->mfma_terms);
        }
        return refid_launch_pointwise(a, st, nullptr, d->mfma_terms);
    }
    if (d->algo == 4) {
        static int cus = 0;
        if (cus == 0) cus = refid_device_cu_count();
        REFID_CHECK(d->pad == 1, "conv2d: the split tile's geometries all have pad 1");
        return refid_launch_split3x3(a, d->mfma_terms ? d->mfma_terms : 6, f == F_3x3 ? 0 : (f == F_4x4s2 ? 1 : 2),
                                     cus > 0 ? cus : 256, d->split_k, st);
    }
    if (d->algo == 5) return refid_launch_wino6(a, d->ws, d->ws_bytes, d->split_k, d->wino_tile, d->mfma_terms, st);
    if (d->algo == 1) {
        REFID_CHECK(d->c_b == 0 || d->c_a % 8 == 0, "conv2d: Winograd tile needs c_a %% 8 == 0 for two sources");
        const long long lim = 0x7fffffffLL;      // buffer-load byte offsets are 32-bit
        REFID_CHECK((long long)d->n * d->h * d->w * d->ld_a * 4 < lim &&
                        (d->c_b == 0 || (long long)d->n * d->h * d->w * d->ld_b * 4 < lim),
                    "conv2d: tensor too large for the Winograd tile's 32-bit offsets (use algo 0)");
        return refid_launch_wino3x3(a, d->ws, d->ws_bytes, d->split_k, d->wino_tile, st);
    }
    switch (f) {
        case F_3x3:
            if (bn == 32) return launch<C3_32>(a, 1, st);
            if (bn == 64) return launch<C3_64>(a, 1, st);
            return launch<C3_128>(a, 1, st);
        case F_5x5: return launch<C5_32>(a, 1, st);
        case F_1x1:
            if (bn == 32) return launch<C1_32>(a, 1, st);
            if (bn == 64) return launch<C1_64>(a, 1, st);
            return launch<C1_128>(a, 1, st);
        case F_4x4s2: return launch<C4S2_64>(a, 1, st, SplitArgs{d->ws, d->ws_bytes, d->split_k});
        case F_2x2s2: return bn == 64 ? launch<C2S2_64>(a, 1, st) : launch<C2S2_128>(a, 1, st);
        case F_convT: return bn == 64 ? launch<CT_64>(a, 1, st) : launch<CT_128>(a, 1, st);
        case F_downDgrad: {
            const SplitArgs sp{d->ws, d->ws_bytes, d->split_k};
            return bn == 64 ? launch<CD_64>(a, 4, st, sp) : launch<CD_128>(a, 4, st, sp);
        }
        default: break;
    }
    refid_set_error("conv2d: no tile for this geometry");
    return 1;
}
