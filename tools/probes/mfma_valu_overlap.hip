// Hardware characterisation (gfx950): how much independent VALU / LDS work can hide under fp32 MFMA issue?
// One "phase" = 32 x v_mfma_f32_32x32x2_f32 on 8 independent accumulators (the Winograd tile's K chunk), plus
// NV independent VALU ops and NL ds_read_b128 per phase.  Reports cycles per phase per SIMD for 1 and 2 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/probes/mfma_valu_overlap.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, bool DEP, bool ILV = false>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    __shared__ f32x4 lds[1024];
    lds[threadIdx.x] = f32x4{seed, seed, seed, seed};
    lds[threadIdx.x + 256] = f32x4{seed, 1.f, seed, 2.f};
    __syncthreads();
    f32x16 acc[8];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + k;
    f32x4 l = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (NL > 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k) l += lds[(threadIdx.x + 17 * k + it) & 511];
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k & 7] = v[k & 7] * 1.0001f + b;       // independent of the MFMAs unless DEP
        const float bb = DEP ? (v[0] + l[0]) : b;
        if (ILV) {          // consecutive MFMAs hit different accumulators (no back-to-back dependency)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        acc[g * 4 + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[g * 4 + x], 0, 0, 0);
        } else {
#pragma unroll
            for (int x = 0; x < 8; ++x)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[x], 0, 0, 0);
        }
    }
    float s = l[0] + l[1] + l[2] + l[3];
    for (int k = 0; k < 8; ++k) s += v[k];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NV, int NL, bool DEP, bool ILV = false>
void run(const char* name, float* d) {
    for (int wps = 1; wps <= 2; ++wps) {
        const int iters = 2000, blocks = 256 * wps;       // 4 waves per block -> wps waves per SIMD on 256 CUs
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<NV, NL, DEP, ILV><<<blocks, 256>>>(d, 10, 1.f);
        hipEventRecord(e0);
        probe<NV, NL, DEP, ILV><<<blocks, 256>>>(d, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / iters;     // cycles per loop iteration (all resident waves in parallel)
        printf("%-34s waves/SIMD %d: %8.0f cycles/phase  (MFMA floor %d)  util %.2f\n", name, wps, cyc, 2048 * wps,
               2048.0 * wps / cyc);
    }
}

int main() {
    float* d; hipMalloc(&d, 1024);
    run<0, 0, false>("32 MFMA only (chains of 4)", d);
    run<0, 0, false, true>("32 MFMA only (interleaved accs)", d);
    run<128, 16, true, true>("+128 VALU + 16 LDS, dep, interleaved", d);
    run<64, 0, false>("+ 64 independent VALU", d);
    run<128, 0, false>("+128 independent VALU", d);
    run<256, 0, false>("+256 independent VALU", d);
    run<128, 16, false>("+128 VALU + 16 ds_read_b128", d);
    run<128, 16, true>("+128 VALU + 16 LDS, MFMA depends", d);
    return 0;
}
