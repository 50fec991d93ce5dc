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
//     v_mfma_f32_32x32x2_f32.  A wave owns 8 of the 16 xi (transform rows i = 2h, 2h+1) for 32
//     output channels: 8 accumulators = 128 AGPRs, so two waves fit per SIMD.  4 waves =
//     {xi half h} x {channel half nt}.
//   * both MFMA operands live in REGISTERS: the raw 6x34-pixel input halo of the next chunk is
//     prefetched global->VGPR->LDS (double buffered, 6.5 KB each) while the current chunk's MFMAs
//     run; each lane reads the 3x4 patch rows of ITS tile / channel quad (12 x ds_read_b128) and
//     computes its own 8 fragments of B^T d B (16 float4 adds).  One barrier per chunk.
//   * transformed weights U = G g G^T are produced once per step by the pack kernel in the
//     [chunk][xi][cout][8] layout; a lane's fragment is one 16-byte buffer load per xi (a wave reads
//     1 KB contiguous), prefetched one chunk ahead -- weights never pass through LDS.
//   * output transform: each lane reduces its 8 accumulators along j in registers, the two xi
//     halves swap one half of the row-partials through LDS (32 KB, after the K loop), and every
//     wave finishes one output row parity with the fused 16-byte epilogue.
#include "common.h"
#include "conv_args.h"

namespace {

constexpr int TW = 32;                  // output pixels per workgroup row
constexpr int KC = 8;                   // input channels per chunk
constexpr int HWD = TW + 2;
constexpr int LDS_BYTES = 32768;        // 2 raw buffers (13-22 KB) in the K loop; 32 KB exchange afterwards
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = -1;                 // voffset 0xFFFFFFFF: buffer loads return 0 (hardware range check)

// <NTN, MTN>: the two waves that are not xi-halves split either the 64 output channels (NTN = 2:
// tile 4x32 pixels x 64 channels) or the pixels (MTN = 2: tile 8x32 pixels x 32 channels, for the
// 32-channel layers).
template <int NTN, int MTN>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvKArgs a) {
    static_assert(NTN * MTN == 2, "4 waves = 2 xi halves x 2");
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
    const int h = wave & 1;
    const int nt = (NTN == 2) ? (wave >> 1) : 0, mt = (MTN == 2) ? (wave >> 1) : 0;

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
    // U fragment of this lane: row (cout) nt*32+li, K quad kh, for the wave's 8 xi -- read straight
    // from the packed [chunk][xi][cout][8] weights (a wave reads 1 KB contiguous per xi): weights
    // never pass through LDS.
    const int urow = a.coBase + n0 + nt * 32 + li;
    const int voU = (urow < a.CoutPad) ? ((h * 8 * a.CoutPad + urow) * KC + kh * 4) * 4 : OOB;
    const int uStep = a.CoutPad * KC * 4;                  // bytes between xi and xi+1
    const int uChunk = 16 * a.CoutPad * KC * 4;            // bytes per K chunk

    f32x4 rr[R_ITEMS], ufA[8], ufB[8];

    auto load_raw = [&](int ch) {
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
    auto store_raw = [&](int buf) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            const int hp = (tid >> 1) + it * 128;
            if (hp < HP) sR[buf * R_F4 + q * HP + hp] = rr[it];
        }
    };
    auto load_u = [&](int ch, f32x4 (&dst)[8]) {
#pragma unroll
        for (int x = 0; x < 8; ++x)
            dst[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voU, ch * uChunk + x * uStep, 0));
    };

    f32x16 acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // input patch of this lane's tile: halo rows r0..r0+2 (h=0: patch rows 0,1,2; h=1: rows 1,2,3), 4 cols
    const int hp0 = (4 * mt + 2 * (li >> 4) + h) * HWD + 2 * (li & 15);
    // transform row 2h   = p - m with (p,m) = halo rows (0,2) for h=0, (1,0) for h=1   [d0-d2 | d2-d1]
    // transform row 2h+1 =            rows (1)+(2) for h=0, (0)-(2) for h=1            [d1+d2 | d1-d3]
    const int rAp = h ? HWD : 0, rAm = h ? 0 : 2 * HWD;
    const int rBp = h ? 0 : HWD, rBm = 2 * HWD;
    const float sB = h ? -1.f : 1.f;

    // one K chunk: cur = this chunk's U fragments, nxt = register set the next chunk's are prefetched into
    auto phase = [&](int ch, f32x4 (&cur)[8], f32x4 (&nxt)[8]) {
        const bool more = ch + 1 < a.nchunks;
        // Everything still in flight was issued one phase ago and is needed NOW (U(ch) by the MFMAs, raw(ch+1) by
        // store_raw).  Stating that as an explicit vmcnt(0) keeps the compiler's conservative, path-merged counters
        // from draining THIS phase's prefetches inside the MFMA section (which made every other phase last a full
        // memory latency).  simm16: vmcnt = 0, expcnt = 7, lgkmcnt = 15 (gfx9 encoding).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (more) {
            store_raw((ch + 1) & 1);                 // raw(ch+1): loaded one phase ago
            load_u(ch + 1, nxt);
            if (ch + 2 < a.nchunks) load_raw(ch + 2);
        }
        // B^T d B for this lane's tile / channel quad, one transform row at a time, in registers
        const f32x4* r = sR + (ch & 1) * R_F4 + kh * HP + hp0;
        __builtin_amdgcn_s_setprio(1);
        {
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = r[rAp + b] - r[rAm + b];
            const f32x4 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][kk], v0[kk], acc[0], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[1][kk], v1[kk], acc[1], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[2][kk], v2[kk], acc[2], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[3][kk], v3[kk], acc[3], 0, 0, 0);
        }
        {
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = r[rBp + b] + r[rBm + b] * sB;
            const f32x4 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[4][kk], v0[kk], acc[4], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[5][kk], v1[kk], acc[5], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[6][kk], v2[kk], acc[6], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[7][kk], v3[kk], acc[7], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();                             // raw(ch) consumed by every wave; raw(ch+1) visible
    };

    // prologue: raw(0) -> LDS, U(0) -> regs; raw(1) in flight
    load_raw(0);
    load_u(0, ufA);
    store_raw(0);
    if (a.nchunks > 1) load_raw(1);
    __syncthreads();

    for (int ch = 0; ch < a.nchunks; ch += 2) {
        phase(ch, ufA, ufB);
        if (ch + 1 < a.nchunks) phase(ch + 1, ufB, ufA);
    }

    // ---- output transform --------------------------------------------------------------------------
    // lane: tile li, channels (r&3)+8(r>>2)+4kh of the wave's 32;  acc[i'*4+j] = M[2h+i'][j]
    // R[i'][b] = sum_j M[i'][j] A^T[b][j] :  b=0: M0+M1+M2   b=1: M1-M2-M3
    f32x16 keep[2];
    float* xch = reinterpret_cast<float*>(smem);            // [wave][b][reg][lane], 8 KB per wave
    const float ks = h ? -1.f : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float r00 = acc[0][r] + acc[1][r] + acc[2][r], r01 = acc[1][r] - acc[2][r] - acc[3][r];
        const float r10 = acc[4][r] + acc[5][r] + acc[6][r], r11 = acc[5][r] - acc[6][r] - acc[7][r];
        // Y[a][b] = sum_i A^T[a][i] R[i][b]:  a=0: R0+R1+R2   a=1: R1-R2-R3
        //   h=0 (rows 0,1): own parity a=0 gets R0+R1, partner's a=1 gets R1
        //   h=1 (rows 2,3): own parity a=1 gets -R2-R3, partner's a=0 gets R2
        keep[0][r] = ks * (r00 + r10);
        keep[1][r] = ks * (r01 + r11);
        xch[((wave * 2 + 0) * 16 + r) * 64 + lane] = h ? r00 : r10;
        xch[((wave * 2 + 1) * 16 + r) * 64 + lane] = h ? r01 : r11;
    }
    __syncthreads();
    const int partner = wave ^ 1;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[b][r] += xch[((partner * 2 + b) * 16 + r) * 64 + lane];

    // ---- fused epilogue: this wave owns output row parity a = h, columns b = 0,1 ------------------
    const int oy = oy0 + 4 * mt + 2 * (li >> 4) + h;
    if (oy >= a.Ho) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int j0 = n0 + nt * 32 + 8 * g + 4 * kh;
        if (j0 >= a.Cout) continue;
        const bool vec = a.vecOK && (j0 + 3 < a.Cout);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const float* bp = a.bias + a.coBase + j0;
            if (vec) bv = *reinterpret_cast<const f32x4*>(bp);
            else
#pragma unroll
                for (int k = 0; k < 4; ++k) if (j0 + k < a.Cout) bv[k] = bp[k];
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = ox0 + 2 * (li & 15) + b;
            if (ox >= a.Wo) continue;
            const long long op = (long long)(n * a.Ho + oy) * a.Wo + ox;
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lrelu(keep[b][4 * g + k] + bv[k], a.slopePre);
            if (vec) {
                if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], a.slopePost);
                if (a.mask) {
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
                }
                *reinterpret_cast<f32x4*>(a.out + op * a.ldO + j0) = v;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 + k >= a.Cout) break;
                    float t = v[k];
                    if (a.res) t += a.res[op * a.ldR + j0 + k];
                    t = lrelu(t, a.slopePost);
                    if (a.mask) t *= (a.mask[op * a.ldM + j0 + k] > 0.f) ? 1.f : a.slopeMask;
                    a.out[op * a.ldO + j0 + k] = t;
                }
            }
        }
    }
}

}  // namespace

int refid_launch_wino3x3(const ConvKArgs& ka, hipStream_t st) {
    ConvKArgs a = ka;
    const bool narrow = a.Cout <= 32;           // 32-channel layers: split pixels instead of channels
    const int th = narrow ? 8 : 4, bn = narrow ? 32 : 64;
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, th);
    a.nchunks = cdiv(a.Ctot, KC);
    a.ncot = cdiv(a.Cout, bn);
    dim3 grid(round_up(a.tilesX * a.tilesY * a.N, 8) * a.ncot);
    if (narrow) hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(256), LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_wino_kernel<2, 1>), grid, dim3(256), LDS_BYTES, st, a);
    REFID_LAUNCH_CHECK("conv_wino");
    return 0;
}
