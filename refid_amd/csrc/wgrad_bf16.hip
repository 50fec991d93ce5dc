// Weight gradient of the 3x3 / stride-1 convolutions with bf16 matrix-core operands (BASELINE config 3:
// `compute_dtype: bf16`; fp32 accumulation, fp32 slabs, fp32 parameter gradients).
//
//   dW[o][i][ky][kx] += sum_{n,y,x} g[n,y,x,o] * src[n, y+ky-1, x+kx-1, i]        db[o] += sum g
//
// GEMM view as in conv_wgrad.hip: M = input channels, N = output channels, K = output pixels, one accumulator per
// tap.  v_mfma_f32_32x32x16_bf16 consumes 16 pixels per instruction and wants, per lane, EIGHT CONSECUTIVE K values
// (pixels) of ONE channel in a 16-byte register pair -- the transpose of the NHWC tensors.  So the tiles are
// transposed while they are staged: a thread loads 4 (gradient) / 10 (input halo) consecutive pixels x 4 channels
// as float4, converts to bf16 (RNE) and writes one 16-byte [channel][8 pixels] piece per channel.  A tap's kx shift
// would make the input reads 2-byte misaligned, so the halo tile is kept in three copies shifted by 0 / 1 / 2
// columns (the thread that holds 10 consecutive pixels emits all three); ky shifts are whole rows.  Every MFMA
// operand is then one aligned, conflict-free ds_read_b128 (row pitches of 272 B / 144 B spread the 16 lanes of a
// read over all banks).
//
// Same tile geometry, split-K, slab layout ([split][tap][CoP][CiP], D[ci][co]) and phases as the fp32 plan W3 of
// conv_wgrad.hip (64 x 64 channels, 2 x 32 pixels per K tile), so refid_conv2d_wgrad re-uses its workspace sizing and
// its reduction kernel unchanged.  This kernel is bandwidth / staging bound (the bf16 matrix rate is 16x the fp32
// one): its cost is the two fp32 tensors it streams, ~2.2x less than the fp32 Winograd weight gradient it replaces.
#include "common.h"
#include "wgrad_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 2, TW = 32;                 // output pixels per K tile
constexpr int COT = 64, CIT = 64;
constexpr int HR = TH + 2;                     // halo rows
constexpr int GP = TH * TW * 2 + 16;           // sG row pitch (bytes): 64 pixels bf16 + pad  = 144
constexpr int XP = HR * TW * 2 + 16;           // sX row pitch (bytes): 4 x 32 pixels bf16 + pad = 272
constexpr int SG_BYTES = COT * GP;             // 9216
constexpr int SX_BYTES = 3 * CIT * XP;         // 52224
constexpr int LDS_BYTES = SG_BYTES + SX_BYTES + COT * 4;

__device__ __forceinline__ bf16x8 pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
    bf16x8 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
    r[4] = (__bf16)e; r[5] = (__bf16)f; r[6] = (__bf16)g; r[7] = (__bf16)h;
    return r;
}

__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;
    char* sX = smem + SG_BYTES;
    float* sBias = reinterpret_cast<float*>(smem + SG_BYTES + SX_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;               // 32-channel sub-tiles: output (wr) x input (wc)
    const int co0 = blockIdx.z * COT, ci0 = blockIdx.y * CIT;
    const int split = blockIdx.x;

    // ---- staging roles ---------------------------------------------------------------------------------
    // gradient: all 256 threads -> (channel quad gq of 16, pixel group gpg of 16: row gpg / 8, columns (gpg % 8) * 4 ..+3)
    const int gq = tid & 15, gpg = tid >> 4;
    const int gco = co0 + gq * 4;
    const bool gcok = gco < a.Co;
    // input halo: all 256 threads -> (channel quad xq of 16, halo row xr of 4, column group xg of 4: halo columns
    // xg * 8 .. xg * 8 + 9)
    const int xq = tid & 15, xr = (tid >> 4) & 3, xg = tid >> 6;
    const int xc = ci0 + xq * 4;
    const bool xcok = xc < a.Ctot;
    // channel quads beyond the sources (first recurrent step: no second source yet) read a valid dummy address of the
    // first source and are zeroed by the select below
    const bool xFromA = xc < a.Ca || !xcok;
    const int xld = xFromA ? a.ldA : a.ldB;
    const int ntAll = a.ntiles * a.groups;                 // tiles of all grouped time steps
    const int xcc = xFromA ? xc : xc - a.Ca;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 rg[4], rx[10];

    auto load_tile = [&](int pt) {
        const int grp = pt / a.ntiles;                     // workgroup-uniform: the time step this tile belongs to
        const float* gsrc = a.g[grp];
        const float* xsrc = xFromA ? a.inA[grp] : a.inB[grp];
        int t = pt - grp * a.ntiles;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        const int n = t / a.tilesY;
        const int oy0 = ty * TH, ox0 = tx * TW;
        {
            const int oy = oy0 + (gpg >> 3), oxb = ox0 + (gpg & 7) * 4;
            // clamped address + select: a per-lane branch around a load makes the compiler wait for it on the spot
            const bool rowok = gcok && oy < a.Ho;
            const float* base = gsrc + ((long long)(n * a.Ho + (rowok ? oy : 0)) * a.Wo) * a.ldG + (gcok ? gco : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ox = oxb + j;
                const bool ok = rowok && ox < a.Wo;
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)(ok ? ox : 0) * a.ldG);
                rg[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        {
            const int iy = oy0 - a.pad + xr, ixb = ox0 - a.pad + xg * 8;
            const bool rowok = xcok && iy >= 0 && iy < a.H;
            const float* base = xsrc + ((long long)(n * a.H + (rowok ? iy : 0)) * a.W) * xld + (xcok ? xcc : 0);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int ix = ixb + j;
                const bool ok = rowok && ix >= 0 && ix < a.W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)(ok ? ix : 0) * xld);
                rx[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x4 h;
            h[0] = (__bf16)rg[0][k]; h[1] = (__bf16)rg[1][k]; h[2] = (__bf16)rg[2][k]; h[3] = (__bf16)rg[3][k];
            *reinterpret_cast<bf16x4*>(sG + (gq * 4 + k) * GP + gpg * 8) = h;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum += rg[j];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bf16x8 h = pack8(rx[s][k], rx[s + 1][k], rx[s + 2][k], rx[s + 3][k], rx[s + 4][k], rx[s + 5][k],
                                       rx[s + 6][k], rx[s + 7][k]);
                *reinterpret_cast<bf16x8*>(sX + (s * CIT + xq * 4 + k) * XP + (xr * 4 + xg) * 16) = h;
            }
    };

    int pt = split;
    if (pt < ntAll) {
        load_tile(pt);
        store_tile();
    }
    __syncthreads();

    const char* gA = sG + (wr * 32 + li) * GP + kh * 16;            // + kstep * 32 bytes
    const char* xB = sX + (wc * 32 + li) * XP + kh * 16;            // + (kx * CIT) * XP + ((r + ky) * 4 + half * 2) * 16

    for (; pt < ntAll; pt += a.nsplit) {
        const bool more = pt + a.nsplit < ntAll;
        if (more) load_tile(pt + a.nsplit);
#pragma unroll
        for (int ks = 0; ks < TH * 2; ++ks) {                        // 16 pixels per step: row ks / 2, columns (ks & 1) * 16 ..
            const int r = ks >> 1, half = ks & 1;
            __builtin_amdgcn_sched_barrier(0);       // keep one step's ten operand reads live at a time (register budget)
            const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gA + ks * 32);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xB + (kx * CIT) * XP + ((r + ky) * 4 + half * 2) * 16);
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, gf, acc[ky * 3 + kx], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- partial slab [split][tap][co][ci]; D[ci][co]: lane li = output channel, register quad q = 4 consecutive ci
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* sl = a.slabs + ((long long)(split * 9 + tap) * a.CoP) * a.CiP;
        const int co = co0 + wr * 32 + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = ci0 + wc * 32 + 8 * q + 4 * kh;
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[tap][4 * q + k];
            f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
            if (a.accum) v += *dst;
            *dst = v;
        }
    }
    if (a.bslabs != nullptr && blockIdx.y == 0) {
        // fixed order (no LDS atomics): xor-shuffle tree over the wave's threads that share gq, then the four waves
        // in sequence; the operand tiles at the start of the LDS are dead here (every wave is past its last read)
        float* sred = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = refid_wave_rows_sum<16>(bsum[k]);
            if ((tid & 63) < 16) sred[(tid >> 6) * COT + gq * 4 + k] = v;
        }
        __syncthreads();
        if (tid < COT) {
            const float tot = ((sred[tid] + sred[COT + tid]) + sred[2 * COT + tid]) + sred[3 * COT + tid];
            float* dst = a.bslabs + (long long)split * a.CoP + co0 + tid;
            *dst = a.accum ? *dst + tot : tot;
        }
    }
}

}  // namespace

int refid_wgrad_bf16_launch(const WgKArgs& a, int nciT, int ncoT, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = refid_lds_attr_once(attr_done, &wgrad_bf16_kernel, LDS_BYTES, "wgrad_bf16")) return rc;
    hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(a.nsplit, nciT, ncoT), dim3(256), LDS_BYTES, st, a);
    REFID_LAUNCH_CHECK("wgrad_bf16");
    return 0;
}
