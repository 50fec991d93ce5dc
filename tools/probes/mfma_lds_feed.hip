// Hardware characterisation (gfx950): can ds_read_b128 fragment reads hide under bf16 MFMAs when nothing depends on
// them but the NEXT iteration's MFMAs?  One iteration = NM x v_mfma_f32_32x32x16_bf16 on 4 accumulators fed from
// registers read one iteration earlier (software pipelined, immediate-offset addresses: no VALU), NR reads per
// iteration.  Prints matrix-pipe busy fraction for 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe2 tools/probes/mfma_lds_feed.hip && /tmp/probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NR, int NM, bool PIPE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    extern __shared__ f32x4 lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        if (seed == 1.f) lds[i] = f32x4{seed, seed, 0.f, 0.f};
        else {              // pseudo-random bf16 pairs of magnitude ~1 (toggle power as with real data)
            unsigned h = i * 2654435761u + blockIdx.x * 40503u;
            unsigned w[4];
            for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; w[k] = (h & 0x807f807fu) | 0x3f003f00u; }
            lds[i] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
    }
    __syncthreads();
    f32x16 acc[4];
    for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    constexpr int NF = NR > 0 ? NR : 1;
    f32x4 fa[NF], fb[NF];
    const f32x4* p = lds + (threadIdx.x & 63) + (threadIdx.x >> 6) * 64;
    for (int k = 0; k < NF; ++k) { fa[k] = p[k * 256]; fb[k] = p[k * 256 + 128]; }
    auto mm = [&](const f32x4 (&f)[NF]) {
#pragma unroll
        for (int k = 0; k < NM; ++k)
            acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[k % NF]),
                                                                  __builtin_bit_cast(bf16x8, f[(k + 1) % NF]), acc[k & 3], 0, 0, 0);
    };
    for (int it = 0; it < iters; it += 2) {
        if (PIPE) {
            if (NR > 0) {
#pragma unroll
                for (int k = 0; k < NR; ++k) fb[k] = p[k * 256 + ((it & 2) ? 64 : 0)];
            }
            mm(fa);
            if (NR > 0) {
#pragma unroll
                for (int k = 0; k < NR; ++k) fa[k] = p[k * 256 + 128 + ((it & 2) ? 64 : 0)];
            }
            mm(fb);
        } else {
            if (NR > 0) {
#pragma unroll
                for (int k = 0; k < NR; ++k) fa[k] = p[k * 256 + ((it & 2) ? 64 : 0)];
            }
            mm(fa);
            if (NR > 0) {
#pragma unroll
                for (int k = 0; k < NR; ++k) fb[k] = p[k * 256 + 128 + ((it & 2) ? 64 : 0)];
            }
            mm(fb);
        }
    }
    float s = 0.f;
    for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NR, int NM, bool PIPE>
void run(const char* name, float* d, float seed = 1.f) {
    hipFuncSetAttribute((const void*)&probe<NR, NM, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int wps = 1; wps <= 2; ++wps) {
        const int iters = 2000, blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<NR, NM, PIPE><<<blocks, 256, 65536>>>(d, 10, 1.f);
        hipEventRecord(e0);
        probe<NR, NM, PIPE><<<blocks, 256, 65536>>>(d, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ideal = (double)iters * NM * 32 * wps / 2.4e9 * 1e3;      // ms at 2.4 GHz
        printf("%-34s waves/SIMD %d: %.3f ms  ideal(2.4GHz) %.3f ms  -> pipe %.2f\n", name, wps, ms, ideal, ideal / ms);
    }
}

int main() {
    float* d; hipMalloc(&d, 4);
    run<0, 24, false>("24 MFMA, no reads", d);
    run<6, 24, false>("24 MFMA +  6 reads, read->use", d);
    run<6, 24, true>("24 MFMA +  6 reads, pipelined", d);
    run<12, 24, false>("24 MFMA + 12 reads, read->use", d);
    run<12, 24, true>("24 MFMA + 12 reads, pipelined", d);
    run<12, 12, true>("12 MFMA + 12 reads, pipelined", d);
    run<0, 24, false>("24 MFMA, no reads, RANDOM data", d, 2.f);
    run<12, 24, false>("24 MFMA + 12 reads, RANDOM data", d, 2.f);
    return 0;
}
