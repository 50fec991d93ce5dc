// Persistent Winograd F(2x2,3x3) tile: ONE wave per SIMD, the whole 512-register file per wave.
//
// Same contract as conv_wino.hip (out = mask(post(pre(conv3x3([in_a|in_b]) + bias) + res)), forward and input
// gradient), used when the problem has enough tiles to keep every CU busy for several tiles (refid_launch_wino3x3
// decides).  What the in-kernel trace of the 2-waves-per-SIMD tile showed (tools/probes/wino_trace.py, 64->64 @256^2,
// B=8): per workgroup 3.4 us prologue + 18.0 us K loop + 3.1 us row exchange + 5.0 us epilogue, and the CU's matrix
// pipe without ANY workgroup in its K loop for 26 % of the time -- a tile's fixed costs (two dependent load latencies
// up front, the exchange barrier, the store drain at the end) are as long as the K loop of the short-K layers, and two
// co-resident workgroups do not interleave them away.  This kernel removes them instead:
//   * 256 threads = 4 waves, __launch_bounds__(256, 1): 256 accumulator registers (AGPRs) + 256 VGPRs per lane.
//     A wave still owns transform row i (xi = 4i..4i+3) but for FOUR 32x32 tiles: 2 channel tiles x 2 pixel tiles
//     (workgroup tile 8x32 pixels x 64 channels) or 4 pixel tiles (16x32 x 32 channels): 64 MFMAs per K chunk from
//     8 weight-fragment loads (each now feeds two pixel tiles) and 8 LDS reads per pixel tile (each feeds two channel
//     tiles) -- on gfx950 every 16-byte-per-lane register write costs matrix-pipe time, so operand
//     re-use is what buys MFMA utilisation;
//   * persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ... (XCD-preserving).  The next tile's first two
//     raw chunks and first weight fragments are requested during the current tile's LAST K chunk and stay in
//     registers across the epilogue; output stores are fire-and-forget (nothing waits for them before the next
//     tile's MFMAs start); residual / mask are requested before the exchange barrier.  Per tile only the row
//     exchange through LDS (one barrier) and the epilogue's LDS reads are serial with the matrix pipe;
//   * LDS halo image stored with even / odd pixel columns split and a row pitch of 40 float4, so the four patch
//     columns of the 32 tiles of a wave are conflict-free ds_read_b128 (the 2-waves tile had 2-way conflicts).
#include "../common.h"
#include "../conv_args.h"

namespace {

constexpr int TW = 32;                  // output pixels per tile row
constexpr int KC = 8;                   // input channels per chunk
constexpr int HWD = TW + 2;             // halo width
constexpr int RP = 40;                  // LDS row pitch of the halo image (float4)
constexpr int XL = 66;                  // exchange row length (float4): [kh][33]
constexpr int OOB = -1;                 // voffset 0xFFFFFFFF: buffer loads return 0 (hardware range check)
constexpr int NT = 4;                   // 32x32 accumulator tiles per transform column j
constexpr int XCH_F4 = 4 * 2 * NT * 4 * XL;             // [i][b][t][rq][XL]
constexpr int LDS_BYTES_P = XCH_F4 * 16;                // 135168 B; the raw halo buffers alias its start

// tools/probes/wino_trace.py (-DREFID_WINO_TRACE): shader-clock stamps of the K-loop phases of waves 0 and 3 of ONE
// workgroup.  With one wave per SIMD the accounting is exact: nothing else runs on the SIMD.
#ifdef REFID_WINO_TRACE
__device__ unsigned long long* g_wino2_ktrace = nullptr;
__device__ int g_wino2_ktrace_wg = -1;
#define P_KSTAMP(ch, slot)                                                                                     \
    do {                                                                                                       \
        if (ktrace && (threadIdx.x & 63) == 0)                                                                 \
            g_wino2_ktrace[(((threadIdx.x >> 6) * 64 + ((ch) & 63)) * 8) + (slot)] = clock64();                 \
    } while (0)
#else
#define P_KSTAMP(ch, slot) do {} while (0)
#endif

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global loads / stores
// (__syncthreads() also drains the vector-memory counter once stores are in flight -- here the previous tile's
// output stores and the next chunks' prefetches must stay in flight across barriers).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NTN, int MTN>
__global__ __launch_bounds__(256, 1) void conv_wino_p_kernel(const ConvKArgs a) {
    static_assert(NTN * MTN == NT, "a wave owns one transform row x 4 (channel | pixel) tiles");
    constexpr int TH = 4 * MTN;             // output rows per workgroup tile
    constexpr int BN = 32 * NTN;            // output channels per workgroup tile
    constexpr int HPIX = (TH + 2) * HWD;    // raw halo pixels (loader items per channel quad)
    constexpr int HPS = (TH + 2) * RP + 8;  // LDS plane stride (float4) of one channel quad
    constexpr int R_F4 = 2 * HPS;           // one raw buffer
    constexpr int R_ITEMS = (2 * HPIX + 255) / 256;
    static_assert(2 * R_F4 * 16 <= LDS_BYTES_P, "raw buffers alias the exchange area");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sR = reinterpret_cast<f32x4*>(smem);
    f32x4* xch = reinterpret_cast<f32x4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = wave;                                   // transform row i owned by this wave

    const int limA = (int)min((long long)a.N * a.H * a.W * a.ldA * 4, 0x7fffffffLL);
    const int limB = a.inB ? (int)min((long long)a.N * a.H * a.W * a.ldB * 4, 0x7fffffffLL) : 0;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)min((long long)a.nchunks * 16 * a.CoutPad * KC * 4, 0x7fffffffLL), 0x00020000);
    const int q = tid & 1;
    const int uStep = a.CoutPad * KC * 4;                  // bytes between xi and xi+1
    const int uChunk = 16 * a.CoutPad * KC * 4;            // bytes per K chunk
    const int nch = a.nchunks;                             // even, >= 4 (host guarantees)
    const int ntiles = a.tilesX * a.tilesY * a.N;

    // loader thread -> halo pixel (it) : LDS slot (even / odd column split, row pitch RP)
    int lslot[R_ITEMS];
#pragma unroll
    for (int it = 0; it < R_ITEMS; ++it) {
        const int hp = (tid >> 1) + it * 128;
        const int hy = hp / HWD, hx = hp % HWD;
        lslot[it] = (hp < HPIX) ? q * HPS + hy * RP + (hx & 1) * 17 + (hx >> 1) : -1;
    }

    // B^T row i of the 4x4 input patch: t = d[P] + sgn * d[M]:  i=0: d0-d2  i=1: d1+d2  i=2: d2-d1  i=3: d1-d3
    const int rowP = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rowM = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const int tr = li >> 4, tc = li & 15;                   // tile row / column inside a 4-row pixel tile
    const int offP = kh * HPS + (2 * tr + rowP) * RP + tc, offM = kh * HPS + (2 * tr + rowM) * RP + tc;

    struct Tile { int n, oy0, ox0, n0; };
    auto tile_of = [&](int vb) {
        // XCD-aware: virtual block vb runs on XCD vb % 8 (gridDim.x is a multiple of 8); the ncot channel tiles of a
        // pixel tile are consecutive on the same XCD so their shared halo is fetched into that L2 once
        const int xcd = vb & 7, slot = vb >> 3;
        int bt = (slot / a.ncot) * 8 + xcd;
        Tile t;
        t.n0 = (slot % a.ncot) * BN;
        const int tx = bt % a.tilesX; bt /= a.tilesX;
        t.oy0 = (bt % a.tilesY) * TH; t.ox0 = tx * TW;
        t.n = bt / a.tilesY;
        return t;
    };
    auto tile_valid = [&](int vb) { return vb < a.gridTiles && ((vb >> 3) / a.ncot) * 8 + (vb & 7) < ntiles; };

    int voA[R_ITEMS], voB[R_ITEMS], voU[NTN];
    auto set_offsets = [&](const Tile& t) {
        int tid_s = tid;                    // opaque copy: keeps the halo coordinates from being hoisted out of the tile
        asm volatile("" : "+v"(tid_s));     // loop and held in registers across the K loop
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) {
            const int hp = (tid_s >> 1) + it * 128;
            const int iy = t.oy0 - a.pad + hp / HWD, ix = t.ox0 - a.pad + hp % HWD;
            const bool ok = hp < HPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int pix = (t.n * a.H + iy) * a.W + ix;
            voA[it] = ok ? pix * a.ldA * 4 + q * 16 : OOB;
            voB[it] = ok ? pix * a.ldB * 4 + q * 16 : OOB;
        }
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int urow = a.coBase + t.n0 + nt * 32 + li;
            voU[nt] = (urow < a.CoutPad) ? ((ti * 4 * a.CoutPad + urow) * KC + kh * 4) * 4 : OOB;
        }
    };

    f32x4 rr[R_ITEMS], rrS[R_ITEMS], ufA[4 * NTN], ufB[4 * NTN];

    // (Ctot % 8 == 0 and Ca % 8 == 0 are guaranteed by the host: a chunk never straddles the two sources and has no
    // ragged upper quad, so the source choice is a scalar select -- no per-lane branch around a load)
    auto load_raw_one = [&](int ch, f32x4 (&dst)[R_ITEMS], int it) {
        const int c0 = ch * KC;
        const bool fromA = c0 < a.Ca;
        const int soff = (fromA ? c0 : c0 - a.Ca) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(fromA ? a.inA : a.inB), 0, fromA ? limA : limB, 0x00020000);
        dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, fromA ? voA[it] : voB[it], soff, 0));
    };
    auto load_raw = [&](int ch, f32x4 (&dst)[R_ITEMS]) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it) load_raw_one(ch, dst, it);
    };
    auto store_raw = [&](int buf, const f32x4 (&src)[R_ITEMS]) {
#pragma unroll
        for (int it = 0; it < R_ITEMS; ++it)
            if (lslot[it] >= 0) sR[buf * R_F4 + lslot[it]] = src[it];
    };
    auto load_u_one = [&](int ch, f32x4 (&dst)[4 * NTN], int idx) {      // idx = j * NTN + nt
        dst[idx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voU[idx % NTN],
                                                                                ch * uChunk + (idx / NTN) * uStep, 0));
    };
    auto load_u = [&](int ch, f32x4 (&dst)[4 * NTN]) {
#pragma unroll
        for (int idx = 0; idx < 4 * NTN; ++idx) load_u_one(ch, dst, idx);
    };

    // acc[j][t]: transform column j, tile t = nt * MTN + mt
    f32x16 acc[4][NT];

    int vb = blockIdx.x;
    if (!tile_valid(vb)) return;
    Tile cur = tile_of(vb);
    set_offsets(cur);
    // very first tile: the two dependent latencies are paid once per workgroup, not once per tile
    load_raw(0, rrS);
    load_raw(1, rr);
    load_u(0, ufA);

    while (true) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

        const int nvb = vb + gridDim.x;
        const bool have_next = tile_valid(nvb);
        Tile nxt_tile = cur;
        if (have_next) nxt_tile = tile_of(nvb);

        // raw(0) was requested long ago (previous tile's last chunk / kernel start)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        store_raw(0, rrS);
        lds_barrier();

        // The K loop is an explicit software pipeline (one wave per SIMD: nothing else covers a stall).  Block m of a
        // chunk = the 4*NTN*4 MFMAs of pixel tile m; UNDER block m the wave reads the raw patch rows of the NEXT block
        // (pixel tile m+1 of this chunk, or pixel tile 0 of the next chunk) and transforms them, so every LDS read has a
        // whole MFMA block to land.  The chunk's barrier therefore sits in front of its LAST block: all reads of
        // raw(ch) have completed by then, and raw(ch+1) -- written at the top of the phase -- becomes readable.
        // __builtin_amdgcn_sched_barrier(0) pins the order (the scheduler otherwise clusters reads, waits, then MFMAs).
        f32x4 ld[8], vv[2][4];
        auto issue_reads = [&](int buf, int mt) {
            const f32x4* rp = sR + buf * R_F4 + offP + mt * 4 * RP;
            const f32x4* rm = sR + buf * R_F4 + offM + mt * 4 * RP;
            // patch columns 0..3 of tile column tc: even / odd split rows -> +0, +17, +1, +18
            ld[0] = rp[0]; ld[1] = rp[17]; ld[2] = rp[1]; ld[3] = rp[18];
            ld[4] = rm[0]; ld[5] = rm[17]; ld[6] = rm[1]; ld[7] = rm[18];
        };
        auto make_v = [&](f32x4 (&v)[4]) {
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = ld[b] + ld[4 + b] * sgn;
            v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        };
        auto mfma_kk = [&](const f32x4 (&cu)[4 * NTN], const f32x4 (&v)[4], int mt, int kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)              // consecutive MFMAs hit different accumulators
                    acc[j][nt * MTN + mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        cu[j * NTN + nt][kk], v[j][kk], acc[j][nt * MTN + mt], 0, 0, 0);
        };

        // pipeline fill: V of (chunk 0, pixel tile 0) -- the one exposed LDS round trip per tile
        issue_reads(0, 0);
        make_v(vv[0]);

        // one K chunk.  `seam`: 0 = ordinary; 1 = second-to-last chunk (request the NEXT tile's raw(0) into rrS);
        // 2 = last chunk (request the next tile's raw(1) and U(0); no next chunk to read ahead)
        // (`par` = ch & 1 as a literal: nchunks is even and a tile starts at chunk 0, so the LDS buffer of every
        // access is a compile-time constant folded into the ds instructions' immediate offsets)
#ifdef REFID_WINO_TRACE
        const bool ktrace = g_wino2_ktrace && (int)blockIdx.x == g_wino2_ktrace_wg && vb == (int)blockIdx.x + (int)gridDim.x;
#endif
        auto phase = [&](int ch, int par, f32x4 (&cu)[4 * NTN], f32x4 (&nx)[4 * NTN], int seam) {
            P_KSTAMP(ch, 0);
            // everything in flight was issued one phase ago and is needed now (U(ch) by the MFMAs, raw(ch+1) by
            // store_raw); the explicit wait keeps the compiler from draining THIS phase's prefetches mid-phase
            __builtin_amdgcn_s_waitcnt(0x0F70);
            P_KSTAMP(ch, 1);
            if (seam != 2) store_raw(1 - par, rr);       // raw(ch+1): loaded one phase ago
            P_KSTAMP(ch, 2);
            // The phase's prefetches -- U of the next chunk, then the raw halo two chunks ahead (at the tile seam: the
            // next tile's first chunks) -- are NOT issued in a burst: a lone wave spends ~130 cycles per buffer load
            // when the four waves of the CU hit the texture-address unit at once (measured: a third of the phase).
            // They ride between the MFMA groups instead, `PER` per group.
            constexpr int NL = 4 * NTN + R_ITEMS, NSLOTS = 3 * MTN, PER = (NL + NSLOTS - 1) / NSLOTS;
            auto issue_item = [&](int k) {
                if (k >= NL) return;
                if (k < 4 * NTN) {                                   // weight fragments
                    if (seam != 2) load_u_one(ch + 1, nx, k);
                    else if (have_next) load_u_one(0, nx, k);
                } else {                                             // raw halo pieces
                    const int it = k - 4 * NTN;
                    if (seam == 0) load_raw_one(ch + 2, rr, it);
                    else if (have_next) {
                        if (seam == 1) {
                            if (it == 0) set_offsets(nxt_tile);      // (after this phase's own U loads: k < 4*NTN)
                            load_raw_one(0, rrS, it);
                        } else {
                            load_raw_one(1, rr, it);
                        }
                    }
                }
            };
#pragma unroll
            for (int m = 0; m < MTN; ++m) {
                const bool last = (m == MTN - 1);
                if (last) {
                    P_KSTAMP(ch, 3);
                    lds_barrier();                       // raw(ch) consumed by every wave; raw(ch+1) visible
                    P_KSTAMP(ch, 4);
                }
                const bool ahead = !(last && seam == 2);
                __builtin_amdgcn_sched_barrier(0);
                if (ahead) issue_reads(last ? 1 - par : par, last ? 0 : m + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_kk(cu, vv[m & 1], m, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ahead) make_v(vv[(m + 1) & 1]);
#pragma unroll
                for (int kk = 1; kk < 4; ++kk) {
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_kk(cu, vv[m & 1], m, kk);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < PER; ++e) issue_item((m * 3 + kk - 1) * PER + e);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            P_KSTAMP(ch, 5);
        };

        for (int ch = 0; ch < nch - 2; ch += 2) {
            phase(ch, 0, ufA, ufB, 0);
            phase(ch + 1, 1, ufB, ufA, 0);
        }
        phase(nch - 2, 0, ufA, ufB, 1);
        phase(nch - 1, 1, ufB, ufA, 2);
        lds_barrier();                                   // every wave is done with the raw buffers: the exchange aliases them

        // ---- output transform -----------------------------------------------------------------------------
        // lane: tile li, channels (r&3)+8(r>>2)+4kh of a 32-channel tile;  acc[j][t] = M[i][j]
        //   R_i[b] = sum_j M[i][j] A[j][b] :  b=0: M0+M1+M2   b=1: M1-M2-M3          (in registers)
        //   Y[a][b] = sum_i A^T[a][i] R_i[b]:  a=0: R0+R1+R2   a=1: R1-R2-R3          (across the 4 waves, through LDS)
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
                }
                xch[(((ti * 2 + 0) * NT + t) * 4 + rq) * XL + kh * 33 + li] = r0;
                xch[(((ti * 2 + 1) * NT + t) * 4 + rq) * XL + kh * 33 + li] = r1;
            }

        // ---- fused epilogue, coalesced: thread -> (output pixel, channel quad) in memory order; residual / mask of
        // one half are requested before the LDS reads of the previous half so their latency stays off the critical
        // path
        constexpr int C4 = BN / 4;                              // float4 per pixel
        constexpr int NIT = (TH * TW * C4) / 256;               // 16
        constexpr int NG = 4, HALF = NIT / NG;                 // the epilogue runs in NG groups of HALF items
        // (host side guarantees 16-byte channel-quad accesses and Cout % 4 == 0 for this kernel)
        // The per-item indices below are cheap functions of the thread id; left alone, the compiler hoists all of them
        // out of the persistent tile loop and keeps them in ~60 registers across the K loop (spills).  An opaque copy of
        // the thread id per tile keeps them local to the epilogue.
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int c4 = tid_e % C4;                              // 256 % C4 == 0: a thread keeps its channel quad
        const int j0 = cur.n0 + c4 * 4;
        const bool cok = j0 < a.Cout;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias && cok) bv = *reinterpret_cast<const f32x4*>(a.bias + a.coBase + j0);
        f32x4 pres[HALF], pmask[HALF];
        auto req = [&](int h) {
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int pr = ((h * HALF + u) * 256 + tid_e) / C4;
                const int oy = cur.oy0 + pr / TW, ox = cur.ox0 + pr % TW;
                const bool ok = cok && oy < a.Ho && ox < a.Wo;
                // loads are unconditional per lane (clamped address, value selected afterwards): a per-lane branch
                // around a load makes the compiler wait for it on the spot
                const long long op = ok ? (long long)(cur.n * a.Ho + oy) * a.Wo + ox : 0;
                const int jj = ok ? j0 : 0;
                pres[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                pmask[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (a.res) pres[u] = *reinterpret_cast<const f32x4*>(a.res + op * a.ldR + jj);
                if (a.mask) pmask[u] = *reinterpret_cast<const f32x4*>(a.mask + op * a.ldM + jj);
            }
        };
        req(0);
        lds_barrier();

        const int nt_e = (c4 * 4) >> 5, rq_e = ((c4 * 4) & 31) >> 3, ckh_e = ((c4 * 4) & 7) >> 2;
#pragma unroll
        for (int h = 0; h < NG; ++h) {
            f32x4 val[HALF];
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int pr = ((h * HALF + u) * 256 + tid_e) / C4;
                const int row = pr / TW, col = pr % TW;
                const int mt = row >> 2, oa = row & 1, tile = ((row & 3) >> 1) * 16 + (col >> 1), ob = col & 1;
                const int t = nt_e * MTN + mt;
                const f32x4* xp = xch + (((oa * 2 + ob) * NT + t) * 4 + rq_e) * XL + ckh_e * 33 + tile;   // row i0 = oa
                const float sg = oa ? -1.f : 1.f;               // a=0: R0+R1+R2 ; a=1: R1-R2-R3
                val[u] = xp[0] + (xp[2 * NT * 4 * XL] + xp[4 * NT * 4 * XL]) * sg;
            }
            f32x4 cres[HALF], cmask[HALF];
#pragma unroll
            for (int u = 0; u < HALF; ++u) { cres[u] = pres[u]; cmask[u] = pmask[u]; }
            if (h + 1 < NG) req(h + 1);
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int pr = ((h * HALF + u) * 256 + tid_e) / C4;
                const int oy = cur.oy0 + pr / TW, ox = cur.ox0 + pr % TW;
                if (!cok || oy >= a.Ho || ox >= a.Wo) continue;
                f32x4 v = val[u];
                const long long op = (long long)(cur.n * a.Ho + oy) * a.Wo + ox;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k] + bv[k], a.slopePre);
                v += cres[u];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], a.slopePost);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= (cmask[u][k] > 0.f) ? 1.f : a.slopeMask;
                *reinterpret_cast<f32x4*>(a.out + op * a.ldO + j0) = v;
            }
        }
        if (!have_next) break;
        lds_barrier();                                 // exchange area fully read: the raw buffers may be rewritten
        vb = nvb;
        cur = nxt_tile;
    }
}

}  // namespace

#ifdef REFID_WINO_TRACE
extern "C" int refid_wino2_ktrace_set(unsigned long long* buf, int wg) {
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_wino2_ktrace), &buf, sizeof(buf)) == hipSuccess &&
            hipMemcpyToSymbol(HIP_SYMBOL(g_wino2_ktrace_wg), &wg, sizeof(wg)) == hipSuccess) ? 0 : 1;
}
#endif

bool refid_wino3x3_p_eligible(const ConvKArgs& a, int cus) {
    // even number of K chunks >= 4 (register sets alternate per chunk across the tile seam), and enough tiles that every
    // CU walks at least two of them; smaller problems keep the 2-waves-per-SIMD tile (more, smaller workgroups)
    const int nchunks = cdiv(a.Ctot, KC);
    if (nchunks < 4 || (nchunks & 1) || a.Ctot % KC || a.Ca % KC) return false;
    const bool narrow = a.Cout <= 32;
    const int th = narrow ? 16 : 8, bn = narrow ? 32 : 64;
    const long long tiles = (long long)cdiv(a.Wo, TW) * cdiv(a.Ho, th) * a.N * cdiv(a.Cout, bn);
    return tiles >= 2LL * cus;          // cus == 0: forced (tests), any tile count
}

int refid_launch_wino3x3_p(const ConvKArgs& ka, int cus, hipStream_t st) {
    ConvKArgs a = ka;
    const bool narrow = a.Cout <= 32;
    const int th = narrow ? 16 : 8, bn = narrow ? 32 : 64;
    a.tilesX = cdiv(a.Wo, TW);
    a.tilesY = cdiv(a.Ho, th);
    a.nchunks = cdiv(a.Ctot, KC);
    a.ncot = cdiv(a.Cout, bn);
    a.gridTiles = round_up(a.tilesX * a.tilesY * a.N, 8) * a.ncot;      // virtual blocks (XCD-aware enumeration)
    int grid = round_up(cus, 8);
    if (grid > a.gridTiles) grid = a.gridTiles;
    static std::atomic<unsigned long long> d22{0}, d14{0};
    if (int rc = refid_lds_attr_once(d22, &conv_wino_p_kernel<2, 2>, LDS_BYTES_P, "conv_wino_p")) return rc;
    if (int rc = refid_lds_attr_once(d14, &conv_wino_p_kernel<1, 4>, LDS_BYTES_P, "conv_wino_p")) return rc;
    if (narrow) hipLaunchKernelGGL((conv_wino_p_kernel<1, 4>), dim3(grid), dim3(256), LDS_BYTES_P, st, a);
    else hipLaunchKernelGGL((conv_wino_p_kernel<2, 2>), dim3(grid), dim3(256), LDS_BYTES_P, st, a);
    REFID_LAUNCH_CHECK("conv_wino_p");
    return 0;
}
