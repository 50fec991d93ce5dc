// Streaming 1x1 weight gradient (refid_conv2d_wgrad, kh = kw = 1; round 5): EGACA's five 1x1 convs (fusion_modules.py:300-331),
// fuse_two_dir (recurrent_sub_modules.py:291-293) and the image encoder's identity convs (:41-49).
//
//     dW[o][i] (+)= sum_p g[p][o] * x[p][i]          p = pixels of the batch, K of the GEMM
//
// At 64-256 channels this is 8-64 FLOP per byte: an HBM-bound stream, not a matrix-pipe problem (one workgroup with two
// workgroups per CU could consume ~10 TB/s of operands chip-wide).  The register-operand tile it replaces where it applies
// (wgrad_pw_kernel: 4-byte buffer loads straight into MFMA operands) and the 128 x 128 LDS tile sat at 0.26-0.37 of HBM --
// bound by memory instructions in flight, not bytes.  Here the operands travel global -> LDS by LDS-DMA in 1 KB pieces (the LDS
// image IS the NHWC memory layout: [pixel][channels of the tile]), a ring of four buffers keeps three tiles of pixels in flight
// per workgroup, and the matrix cores read the ring with conflict-free ds_read_b32 (lane = channel, MFMA K half = pixel
// parity) -- no vector arithmetic at all in the loop.
//
// Mapping: 256 threads = 4 waves = OW output-channel sub-tiles (32 each) x PH = 4 / OW pixel phases; a wave owns o sub-tile ow
// for all WI input-channel sub-tiles (WI accumulators) and the pixel pairs pp == ph (mod PH) of every buffer.  Channel tile =
// 32 OW (o) x 32 WI (i): 128 x 128 reads every tensor exactly once at c_o = c_i = 128.  Split K over contiguous pixel ranges
// (grid x), slabs [split][co][ci] as the other weight-gradient kernels (phases, grouped time steps, the two-stage deterministic
// reduction of conv_wgrad.hip); pixel phases are added in order through LDS at the end; the bias gradient is the sum of the
// gradient operand as it goes by.  Pixels past the end of a tensor need no test: their offsets lie beyond the buffer
// descriptor's range and the DMA fills zeros.
#include "common.h"
#include "wgrad_args.h"
#include <cstdlib>

namespace {

constexpr int NBUF = 4;
typedef __attribute__((address_space(3))) void* lds_ptr_s;

constexpr int pws_pb(int ow, int wi) { return (ow + wi) <= 4 ? 32 : 16; }                 // pixels per buffer
constexpr int pws_buf_bytes(int ow, int wi) { return pws_pb(ow, wi) * 32 * (ow + wi) * 4; }
constexpr int pws_lds_bytes(int ow, int wi) { return NBUF * pws_buf_bytes(ow, wi) + 1024; }   // + a dump slot for padding pieces
// s_waitcnt with only vmcnt set (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

template <int OW, int WI>
__global__ __launch_bounds__(256, 2) void wgrad_pws_kernel(const WgKArgs a) {
    constexpr int PH = 4 / OW, OT = 32 * OW, IT = 32 * WI, PB = pws_pb(OW, WI);
    constexpr int G_BYTES = PB * OT * 4, X_BYTES = PB * IT * 4, BUF = G_BYTES + X_BYTES;
    constexpr int GP = G_BYTES / 1024, XP = X_BYTES / 1024, NP = GP + XP, PPW = (NP + 3) / 4;      // 1 KB DMA pieces per buffer
    constexpr int GPX = 256 / OT, XPX = 256 / IT;                                               // pixels per piece
    static_assert(G_BYTES % 1024 == 0 && X_BYTES % 1024 == 0 && 2 * PPW <= 63, "piece geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int ow = wave % OW, ph = wave / OW;
    const int co0 = blockIdx.z * OT, ci0 = blockIdx.y * IT;
    const int split = blockIdx.x;

    // the input-channel tile lies in one source (host: c_a % IT == 0 for two sources); a tile beyond the sources (first recurrent
    // step: no second source yet) keeps a valid descriptor, every lane out of range
    const bool xFromA = ci0 < a.Ca || ci0 >= a.Ctot;
    const int xld = xFromA ? a.ldA : a.ldB;
    const long long npix = (long long)a.N * a.H * a.W;
    // (patch form: the sources are the even / odd rows of one tensor of npix / patchW row pairs; a tile of PB pixels lies in one
    //  row -- host: patchW % PB == 0 --, pixels past the end land beyond the descriptor's range as in the dense form)
    const long long spanX = a.patchW ? (npix / a.patchW) * (long long)a.patchRow - (xFromA ? 0 : a.patchRow / 2) : npix * xld;
    const int limG = (int)min(npix * a.ldG * 4, 0x7fffffffLL), limX = (int)min(spanX * 4, 0x7fffffffLL);
    const int tilesPer = (int)((npix + PB - 1) / PB), ntAll = tilesPer * a.groups;
    const int chunk = (ntAll + a.nsplit - 1) / a.nsplit;
    const int t0 = min(split * chunk, ntAll), t1 = min(t0 + chunk, ntAll);

    // per-lane byte offsets inside a piece (pixel of the piece, channel quad); channels beyond the tensors are forced out of range
    const int gq = co0 + (lane % (OT / 4)) * 4, xq = ci0 + (lane % (IT / 4)) * 4;
    const int glc = ((lane / (OT / 4)) * a.ldG + gq) * 4, gbad = gq < a.Co ? 0 : -1;
    const int xlc = ((lane / (IT / 4)) * xld + (xFromA ? xq : xq - a.Ca)) * 4, xbad = xq < a.Ctot ? 0 : -1;

    int qt = t0;                                           // next tile to request
    auto request = [&]() {
        const int grp = qt / tilesPer, p0 = (qt - grp * tilesPer) * PB;
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[grp]), 0, limG, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xFromA ? a.inA[grp] : a.inB[grp]), 0, limX, 0x00020000);
        char* dst = smem + ((qt - t0) % NBUF) * BUF;
        const int xp0 = a.patchW ? ((p0 / a.patchW) * a.patchRow + (p0 % a.patchW) * xld) * 4 : p0 * xld * 4;   // bytes (wave-uniform)
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int q = wave + 4 * k;                    // piece of this wave (wave-uniform)
            if (q < GP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_s)(dst + q * 1024), 16, ((p0 + q * GPX) * a.ldG * 4 + glc) | gbad, 0, 0, 0);
            else if (q < NP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_s)(dst + G_BYTES + (q - GP) * 1024), 16,
                                                         (xp0 + (q - GP) * XPX * xld * 4 + xlc) | xbad, 0, 0, 0);
            else                                           // padding piece: every wave issues PPW pieces per tile (the wait counts them)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_s)(smem + NBUF * BUF), 16, -1, 0, 0, 0);
        }
        ++qt;
    };

    f32x16 acc[WI];
#pragma unroll
    for (int wi = 0; wi < WI; ++wi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wi][r] = 0.f;
    float bs = 0.f;

    const int ntl = t1 - t0;
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k)
        if (k < ntl) request();
    for (int t = 0; t < ntl; ++t) {
        // tiles t+1 and t+2 may still be in flight: wait until this wave's pieces of tile t have landed
        const int ahead = min(ntl - 1 - t, NBUF - 2);
        if (ahead >= 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * PPW));
        else if (ahead == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(PPW));
        else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody is done with tile t-1's buffer
        __builtin_amdgcn_sched_barrier(0);
        if (t + NBUF - 1 < ntl) request();                 // tile t+3 into the buffer of tile t-1
        const float* sG = reinterpret_cast<const float*>(smem + (t % NBUF) * BUF) + ow * 32 + li;
        const float* sX = reinterpret_cast<const float*>(smem + (t % NBUF) * BUF + G_BYTES) + li;
#pragma unroll
        for (int pp = 0; pp < PB / 2 / PH; ++pp) {
            const int px = 2 * (pp * PH + ph) + kh;
            const float gv = sG[px * OT];
            float xv[WI];
#pragma unroll
            for (int wi = 0; wi < WI; ++wi) xv[wi] = sX[px * IT + wi * 32];
#pragma unroll
            for (int wi = 0; wi < WI; ++wi) acc[wi] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[wi], gv, acc[wi], 0, 0, 0);
            bs += gv;
        }
    }

    // ---- pixel phases in order through LDS, then the slab: [split][co][ci]; D[ci][co]: lane li = output channel -----------------
    __syncthreads();                                       // every wave is past its last read of the ring
    float* ex = reinterpret_cast<float*>(smem);            // [wave - OW][wi][16][64], then the bias partials
    constexpr int EXW = (4 - OW) * WI * 16 * 64;
    static_assert((EXW + (4 - OW) * 64) * 4 <= NBUF * BUF, "phase exchange fits the ring");
    if (PH > 1 && ph > 0) {
#pragma unroll
        for (int wi = 0; wi < WI; ++wi)
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[(((wave - OW) * WI + wi) * 16 + r) * 64 + lane] = acc[wi][r];
        ex[EXW + (wave - OW) * 64 + lane] = bs;
    }
    if (PH > 1) __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int p2 = 1; p2 < PH; ++p2) {
            const int w2 = p2 * OW + ow - OW;
#pragma unroll
            for (int wi = 0; wi < WI; ++wi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[wi][r] += ex[((w2 * WI + wi) * 16 + r) * 64 + lane];
            bs += ex[EXW + w2 * 64 + lane];
        }
        float* sl = a.slabs + (long long)split * a.CoP * a.CiP;
        const int co = co0 + ow * 32 + li;
#pragma unroll
        for (int wi = 0; wi < WI; ++wi)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ci = ci0 + wi * 32 + 8 * qd + 4 * kh;
                f32x4 vv;
#pragma unroll
                for (int k = 0; k < 4; ++k) vv[k] = acc[wi][4 * qd + k];
                f32x4* dst = reinterpret_cast<f32x4*>(sl + (long long)co * a.CiP + ci);
                if (a.accum) vv += *dst;
                *dst = vv;
            }
        if (a.bslabs != nullptr && blockIdx.y == 0) {
            const float tot = bs + __shfl_xor(bs, 32, 64);             // the two pixel parities
            if (kh == 0) {
                float* dst = a.bslabs + (long long)split * a.CoP + co;
                *dst = a.accum ? *dst + tot : tot;
            }
        }
    }
}

const bool USE_PWS = !(getenv("REFID_PWS_WGRAD") && getenv("REFID_PWS_WGRAD")[0] == '0');

struct PwsPlan { int ow, wi; };
bool pws_plan(const refid_wgrad_desc* d, PwsPlan& p) {
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;
    const int ci = ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b;
    p.ow = d->c_o >= 128 ? 4 : 2;
    p.wi = ci >= 128 ? 4 : (ci >= 64 ? 2 : 1);
    while (d->c_b && p.wi > 1 && d->c_a % (32 * p.wi)) p.wi /= 2;     // an input-channel tile must lie in one source
    if (p.ow == 4 && p.wi == 1) {                          // (instantiated: 2x1 2x2 2x4 4x2 4x4)
        if (d->c_b && d->c_a % 64) return false;
        p.wi = 2;
    }
    return true;
}

template <int OW, int WI>
int launch_pws(const WgKArgs& a, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (int rc = refid_lds_attr_once(done, &wgrad_pws_kernel<OW, WI>, pws_lds_bytes(OW, WI), "wgrad_pws")) return rc;
    hipLaunchKernelGGL((wgrad_pws_kernel<OW, WI>), grid, dim3(256), pws_lds_bytes(OW, WI), st, a);
    REFID_LAUNCH_CHECK("wgrad_pws");
    return 0;
}

}  // namespace

bool refid_wgrad_pws_ok(const refid_wgrad_desc* d) {
    if (!USE_PWS || d->algo != 0 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return false;
    if (d->c_o < 64 || d->c_o % 32 || d->c_a % 32 || d->c_b % 32 || (d->i_total - d->i_base) % 32) return false;
    if (d->ld_g % 4 || d->ld_a % 4 || (d->c_b && d->ld_b % 4)) return false;
    PwsPlan p;
    if (!pws_plan(d, p)) return false;
    const long long npix = (long long)d->n * d->h * d->w, lim = 0x7fffffffLL;
    return npix * d->ld_g * 4 < lim && npix * d->ld_a * 4 < lim && (!d->c_b || npix * d->ld_b * 4 < lim);
}

// pixels per ring buffer of the tile this geometry takes (the patch form's row width must be a multiple of it); 0: not eligible
int refid_wgrad_pws_pixels_per_buffer(const refid_wgrad_desc* d) {
    PwsPlan p;
    if (!pws_plan(d, p)) return 0;
    return pws_pb(p.ow, p.wi);
}

void refid_wgrad_pws_geo(const refid_wgrad_desc* d, int* ncoT, int* nciT, int* nsplit, int* CoP, int* CiP) {
    PwsPlan p;
    pws_plan(d, p);
    const int ci_geo = (d->phase != 0) ? d->i_total - d->i_base : d->c_a + d->c_b;
    const int ci = ci_geo > d->c_a + d->c_b ? ci_geo : d->c_a + d->c_b;
    *ncoT = cdiv(d->c_o, 32 * p.ow);
    *nciT = cdiv(ci, 32 * p.wi);
    const long long npix = (long long)d->n * d->h * d->w;
    const long long tiles = (npix + pws_pb(p.ow, p.wi) - 1) / pws_pb(p.ow, p.wi);
    int want = cdiv(512, *ncoT * *nciT);                    // two workgroups per CU
    if (want >= 8) want = want / 8 * 8;
    if (want > tiles) want = (int)tiles;
    if (want < 1) want = 1;
    *nsplit = want;
    *CoP = *ncoT * 32 * p.ow;
    *CiP = *nciT * 32 * p.wi;
}

int refid_wgrad_pws_launch(const refid_wgrad_desc* d, const WgKArgs& a, int nciT, int ncoT, hipStream_t st) {
    PwsPlan p;
    pws_plan(d, p);
    for (int k = 0; k < a.groups; ++k)
        REFID_CHECK(((uintptr_t)a.g[k] | (uintptr_t)a.inA[k] | (uintptr_t)(d->c_b ? a.inB[k] : nullptr)) % 16 == 0,
                    "wgrad (1x1 streaming): tensors must be 16-byte aligned (group %d)", k);
    REFID_CHECK(!a.patchW || a.patchW % pws_pb(p.ow, p.wi) == 0, "wgrad (streaming, patch form): the row width must be a multiple of %d pixels",
                pws_pb(p.ow, p.wi));
    const dim3 grid(a.nsplit, nciT, ncoT);
    if (p.ow == 2 && p.wi == 1) return launch_pws<2, 1>(a, grid, st);
    if (p.ow == 2 && p.wi == 2) return launch_pws<2, 2>(a, grid, st);
    if (p.ow == 2 && p.wi == 4) return launch_pws<2, 4>(a, grid, st);
    if (p.ow == 4 && p.wi == 2) return launch_pws<4, 2>(a, grid, st);
    return launch_pws<4, 4>(a, grid, st);
}
