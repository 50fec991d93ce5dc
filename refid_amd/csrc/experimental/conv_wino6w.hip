// Winograd F(2x2,3x3) x six bf16 products, WIDE tile: the arithmetic of conv_wino6.hip (same chunk order, same
// products, same bits) on a workgroup of 8 waves that covers 8x32 output pixels x 64 output channels, with the
// pre-split weight fragments U shared by its two 4x32-pixel halves through an LDS ring.
//
// Why: in the 4-wave tile every wave fetches its own 24 KB of U fragments per 16-channel chunk into registers, and the
// two workgroups of a CU fetch the same bytes twice: 192 KB of U per chunk-pair through the CU's 64 B/clk vector-memory
// return path, 37 % of its capacity in bursts right in front of the MFMAs.  tools/probes/wino6_ablate.py priced it: no U
// loads = -14 ... -21 % kernel time, the same number of load INSTRUCTIONS with every lane reading the same 16 bytes = the
// same gain (bytes, not instructions), half the loads = two thirds of it.  Here the fragments of one transform column j
// (24.5 KB: 4 rows x 3 planes x 64 channels x 16 input channels) arrive ONCE per CU by LDS-DMA (buffer_load ... lds, no
// VGPRs) into a 4-slot ring three stages ahead of their use, and the eight waves read them with conflict-free
// ds_read_b128 (LDS: 256 B/clk, a path of its own).
//
// Mapping: wave w = (transform row i = w & 3, pixel half mt = w >> 2); per chunk two stages (column pairs j = 0,1 | 2,3),
// one barrier per stage:
//   stage top: [first stage: store the raw halo of chunk c+1] -> wait for this wave's DMA / stores -> s_barrier ->
//              DMA of the OTHER pair's two ring slots (first stage: U(c, 2..3); second: U(c+1, 0..1)), i.e. one stage
//              (24 MFMAs per wave) ahead of their use -> [first stage: request raw(c+2); read the raw halo, row
//              transform] -> per column: 6 U fragment reads, column transform + three-plane split of V_j, 12 MFMAs.
// MEASURED (tools/bench_wino6.py, B=8): 0-10 % SLOWER than the 4-wave tile on every config-2 shape, in this form (one
// barrier per column pair, DMA one stage ahead: 177 vs 163 us at 64->64 @256^2, 254 vs 229 us at 512->256 @64^2) and with
// one barrier per column and the DMA three stages ahead (173 / 232 us): what the shared fragments save, the eight-wave
// barriers and the lone workgroup per CU (nobody covers its prologue / output transform) give back.  Kept as an
// experiment (REFID_EXPERIMENTAL_TILES=1, refid_conv_desc.wino_tile = 3); the product library does not contain it.
// LDS: 2 raw halo buffers (10x34 pixels x 16 channels, de-interleaved columns) 51 KB + ring 96 KB; the 132 KB row
// exchange of the output transform re-uses it after the K loop (all DMA drained first).  One workgroup per CU.
#include "../common.h"
#include "../conv_args.h"

namespace {

constexpr int TW = 32, KC = 16, HWD = TW + 2;
constexpr int TH = 8, BN = 64, NTH = 512;
constexpr int ROWP = 40;                          // LDS slots per halo row: even columns at 0..16, odd columns at 20..36
constexpr int PLANE = (TH + 2) * ROWP + 5;        // 405 slots per channel-quad plane
constexpr int R_F4 = 4 * PLANE;                   // one raw halo buffer
constexpr int HP = (TH + 2) * HWD;                // 340 halo pixels
constexpr int R_ITEMS = (4 * HP + NTH - 1) / NTH; // 3
constexpr int USLOT = 24 * 64;                    // float4 slots of one ring slot: 24 pieces [i][plane][nt] of 1 KB
constexpr int XL = 66;
constexpr int LDS_LOOP = (2 * R_F4 + 4 * USLOT) * 16;
constexpr int LDS_EXCH = 2 * 4 * 2 * 2 * 4 * XL * 16;
constexpr int LDS_BYTES = LDS_LOOP > LDS_EXCH ? LDS_LOOP : LDS_EXCH;
constexpr int OOB = -1;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

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

// s_waitcnt vmcnt(N) only (expcnt 7, lgkmcnt 15: no wait): simm16 = N[3:0] | 7 << 4 | 15 << 8 | N[5:4] << 14
#define WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x0F70 | (((N) >> 4) << 14))
#define WAIT_VM0_LGKM0() __builtin_amdgcn_s_waitcnt(0x0070)      // vmcnt(0) and lgkmcnt(0): DMA landed, own LDS stores done

__global__ __launch_bounds__(NTH, 2) void conv_wino6w_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sR = reinterpret_cast<f32x4*>(smem);            // two raw halo buffers
    f32x4* sU = sR + 2 * R_F4;                             // ring of 4 column slots

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave & 3, mt = wave >> 2;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int bt = (slot / a.ncot) * 8 + xcd;
    if (bt >= a.tilesX * a.tilesY * a.N) return;
    const int n0 = (slot % a.ncot) * BN;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int n = bt / a.tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const int limA = (int)min((long long)a.N * a.H * a.W * a.ldA * 4, 0x7fffffffLL);
    const int limB = a.inB ? (int)min((long long)a.N * a.H * a.W * a.ldB * 4, 0x7fffffffLL) : 0;
    const int uPlane = a.CoutPad * KC * 2;
    const int uXi = 3 * uPlane;
    const int uChunk = 16 * uXi;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)min((long long)a.nchunks * uChunk, 0x7fffffffLL), 0x00020000);
    const int q = tid & 3;
    int pixo[R_ITEMS], sdst[R_ITEMS];
#pragma unroll
    for (int it = 0; it < R_ITEMS; ++it) {
        const int hp = (tid >> 2) + it * (NTH / 4);
        const int row = hp / HWD, col = hp % HWD;
        const int iy = oy0 - a.pad + row, ix = ox0 - a.pad + col;
        const bool ok = hp < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pixo[it] = ok ? (n * a.H + iy) * a.W + ix : OOB;
        sdst[it] = hp < HP ? q * PLANE + row * ROWP + (col >> 1) + (col & 1) * 20 : q * PLANE + (TH + 2) * ROWP + (tid & 3);
    }
    // this wave's three pieces of a ring slot: piece = wave*3 + k = (row pi, plane pp, column tile pnt)
    int voUp[3], soUp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int piece = wave * 3 + k;
        const int pi = piece / 6, pp = (piece >> 1) % 3, pnt = piece & 1;
        const int urow = a.coBase + n0 + pnt * 32 + li;
        voUp[k] = (urow < a.CoutPad) ? (urow * KC + kh * 8) * 2 : OOB;
        soUp[k] = pi * 4 * uXi + pp * uPlane;                  // + chunk * uChunk + j * uXi
    }
    const int kper = (a.nchunks + a.ksplit - 1) / a.ksplit;
    const int kc0 = blockIdx.y * kper, kc1 = min(a.nchunks, kc0 + kper);

    f32x4 rr[R_ITEMS];
    auto load_raw = [&](int ch, f32x4 (&dst)[R_ITEMS]) {
        const int c0 = ch * KC;
        const bool fromA = c0 < a.Ca;
        const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
        const int qmask = (ch < kc1 && c0 + q * 4 < a.Ctot) ? 0 : OOB;
        const int ld4 = (fromA ? a.ldA : a.ldB) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(fromA ? a.inA : a.inB), 0, fromA ? limA : limB, 0x00020000);
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            const int vo = (pixo[it] * ld4 + q * 16) | qmask;
            dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
        }
    };
    auto store_raw = [&](int buf, const f32x4 (&src)[R_ITEMS]) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) sR[buf * R_F4 + sdst[it]] = src[it];
    };
    // LDS-DMA of this wave's three 1 KB pieces of column j of chunk ch into ring slot j (a chunk past the range: zeros)
    auto dma_u = [&](int ch, int j) {
        const bool in = ch < kc1;
        char* dstb = smem + (2 * R_F4 + j * USLOT + wave * 3 * 64) * 16;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dstb + k * 1024), 16, in ? voUp[k] : OOB,
                                                     in ? ch * uChunk + j * uXi + soUp[k] : 0, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    const int rowP = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rowM = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const int tbase = 2 * kh * PLANE + (4 * mt + 2 * (li >> 4)) * ROWP + (li & 15);
    const int offP = tbase + rowP * ROWP, offM = tbase + rowM * ROWP;
    constexpr int BOFF[4] = {0, 20, 1, 21};
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};
    const f32x4* pU = sU + ti * (6 * 64) + lane;               // + j * USLOT + (p * 2 + nt) * 64

    // prologue: columns 0, 1 of the first chunk by DMA; raw(kc0) -> LDS, raw(kc0+1) in registers
    dma_u(kc0, 0); dma_u(kc0, 1);
    {
        f32x4 rr0[R_ITEMS];
        load_raw(kc0, rr0);
        load_raw(kc0 + 1, rr);
        store_raw(0, rr0);
    }

    for (int ch = kc0; ch < kc1; ++ch) {
        const int lc = ch - kc0;
        f32x4 t[2][4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // ---- stage top: everything this wave requested has landed; one barrier per column PAIR -------------------
            if (half == 0) store_raw((lc + 1) & 1, rr);    // raw(ch+1): requested one chunk ago
            WAIT_VM0_LGKM0();
            __builtin_amdgcn_s_barrier();                  // DMA / halo of all waves visible; the other pair's slots are free
            __builtin_amdgcn_sched_barrier(0);
            if (half == 0) {
                dma_u(ch, 2); dma_u(ch, 3);                // read one stage from now
                load_raw(ch + 2, rr);
                const f32x4* r = sR + (lc & 1) * R_F4;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int b = 0; b < 4; ++b) t[qq][b] = r[qq * PLANE + offP + BOFF[b]] + r[qq * PLANE + offM + BOFF[b]] * sgn;
            } else {
                dma_u(ch + 1, 0); dma_u(ch + 1, 1);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = half * 2 + jj;
                f32x4 uf[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) uf[p][nt] = pU[j * USLOT + (p * 2 + nt) * 64];
                f32x4 v[2], pl[3];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    v[qq] = (j == 0) ? t[qq][0] - t[qq][2] : (j == 1) ? t[qq][1] + t[qq][2]
                          : (j == 2) ? t[qq][2] - t[qq][1] : t[qq][1] - t[qq][3];
                split8(v[0], v[1], pl);
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, uf[TB[e]][nt]), __builtin_bit_cast(bf16x8, pl[TA[e]]), acc[j][nt], 0, 0, 0);
            }
        }
    }
    // every DMA (incl. the zero-filling refills past the last chunk) must have landed before the LDS is re-used
    WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- output transform (as conv_wino6.hip), one 4-row half per wave quad --------------------------------------
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + mt * (4 * 2 * 2 * 4 * XL);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 r0, r1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * rq + k;
                r0[k] = acc[0][t2][r] + acc[1][t2][r] + acc[2][t2][r];
                r1[k] = acc[1][t2][r] - acc[2][t2][r] - acc[3][t2][r];
            }
            xch[(((ti * 2 + 0) * 2 + t2) * 4 + rq) * XL + kh * 33 + li] = r0;
            xch[(((ti * 2 + 1) * 2 + t2) * 4 + rq) * XL + kh * 33 + li] = r1;
        }
    constexpr int NIT = (TH * TW * (BN / 4)) / NTH;         // 8
    const bool pre = a.vecOK && a.ksplit == 1 && (a.res != nullptr || a.mask != nullptr);
    f32x4 pres[NIT], pmask[NIT];
    if (pre) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = it * NTH + tid;
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

    constexpr int C4 = BN / 4;
    const f32x4* xall = reinterpret_cast<const f32x4*>(smem);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int f = it * NTH + tid;
        const int c4 = f % C4, pr = f / C4;
        const int row = pr / TW, col = pr % TW;
        const int c = c4 * 4;
        const int nt = c >> 5, rq = (c & 31) >> 3, ckh = (c & 7) >> 2;
        const int hm = row >> 2, oa = row & 1, tile = ((row & 3) >> 1) * 16 + (col >> 1), ob = col & 1;
        const int oy = oy0 + row, ox = ox0 + col;
        const int j0 = n0 + c;
        if (oy >= a.Ho || ox >= a.Wo || j0 >= a.Cout) continue;
        const f32x4* xp = xall + hm * (4 * 2 * 2 * 4 * XL) + (((oa * 2 + ob) * 2 + nt) * 4 + rq) * XL + ckh * 33 + tile;
        const float sg = oa ? -1.f : 1.f;
        f32x4 v = xp[0] + (xp[16 * XL] + xp[32 * XL]) * sg;
        const long long op = (long long)(n * a.Ho + oy) * a.Wo + ox;
        if (a.ksplit > 1) {
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
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k] + bv[k], a.slopePre);
        if (vec) {
            if (a.res) v += pre ? pres[it] : *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + j0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], a.slopePost);
            if (a.mask) {
                const f32x4 mv = pre ? pmask[it] : *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + j0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= (mv[k] > 0.f) ? 1.f : a.slopeMask;
            }
            *reinterpret_cast<f32x4*>(a.out + op * a.ldO + j0) = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= a.Cout) break;
                float tv = v[k];
                if (a.res) tv += a.res[op * a.ldR + j0 + k];
                tv = lrelu(tv, a.slopePost);
                if (a.mask) tv *= (a.mask[op * a.ldM + j0 + k] > 0.f) ? 1.f : a.slopeMask;
                a.out[op * a.ldO + j0 + k] = tv;
            }
        }
    }
}

}  // namespace

// the wide tile's grid for this problem (ksplit = 1): pixel tiles of 8x32, channel tiles of 64
int refid_wino6w_workgroups(const ConvKArgs& a) {
    return cdiv(a.Wo, TW) * cdiv(a.Ho, TH) * a.N * cdiv(a.Cout, BN);
}

int refid_launch_wino6w(const ConvKArgs& ka, int ks, hipStream_t st) {
    ConvKArgs a = ka;                      // out / ldO / ksplit / wsStride already set by the caller for ks > 1
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, TH);
    a.nchunks = cdiv(a.Ctot, KC);
    a.ncot = cdiv(a.Cout, BN);
    static std::atomic<unsigned long long> done{0};
    if (int rc = refid_lds_attr_once(done, &conv_wino6w_kernel, LDS_BYTES, "conv_wino6w")) return rc;
    dim3 grid(round_up(a.tilesX * a.tilesY * a.N, 8) * a.ncot, ks);
    hipLaunchKernelGGL(conv_wino6w_kernel, grid, dim3(NTH), LDS_BYTES, st, a);
    REFID_LAUNCH_CHECK("conv_wino6w");
    return 0;
}
