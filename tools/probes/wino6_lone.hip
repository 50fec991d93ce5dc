// Hardware question behind a producer / consumer redesign of the Winograd x six tile: can ONE compute
// wave per SIMD -- 2 x 2 register blocking, 16 accumulators = 256 AGPRs, U fragments re-used by two tile groups (half the
// fragment bytes per MFMA) -- keep the bf16 matrix pipe as busy as the product tile's TWO waves per SIMD do, when the U
// fragments come from global memory (L2) like in the real kernel and only the raw halo is in LDS?  If yes, helper waves
// could take over staging and the epilogue and the compute waves would never enter a memory phase.
//   MT     tile groups per wave (1: 8 accumulators, the product tile; 2: 16 accumulators)
//   WPS    compute waves per SIMD (2 = two workgroups per CU; 1 = one)
//   UD     U prefetch distance in columns (1 = the product tile's: column j+1 requested before column j's MFMAs; 2, 3)
//   KCH    chunks of 16 input channels the walk cycles through (4 = a 64-channel layer, 32 = 512 channels)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/wino6_lone tools/probes/wino6_lone.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

constexpr int HWD = 34;
constexpr int U_XI = 3 * 64 * 16 * 2;          // bytes per transform point: [plane][cout 64][16 channels] bf16
constexpr int U_CHUNK = 16 * U_XI;             // 96 KB per chunk of 16 channels

template <int MT, int WPS, int UD, int KCH, int VAR>
__global__ __launch_bounds__(256, WPS) void probe(const float* __restrict__ ug, float* out, int chunks) {
    extern __shared__ f32x4 lds[];
    constexpr int HP = (4 * MT + 2) * HWD;
    constexpr int RAW = 4 * HP;
    f32x4* sR = lds;
    for (int i = threadIdx.x; i < 2 * RAW; i += 256) {
        unsigned h = i * 2654435761u + blockIdx.x * 40503u;
        float w[4];
        for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; w[k] = ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23)); }
        lds[i] = f32x4{w[0], w[1], w[2], w[3]};
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kh = lane >> 5;
    const int ti = wave;
    f32x16 acc[4][MT][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][m][n][r] = 0.f;
    const int rowP = (ti == 0) ? 0 : ((ti == 2) ? 2 : 1);
    const int rowM = (ti == 0) ? 2 : ((ti == 1) ? 2 : ((ti == 2) ? 1 : 3));
    const float sgn = (ti == 1) ? 1.f : -1.f;
    const int hp0 = (2 * (li >> 4)) * HWD + 2 * (li & 15);
    const int offP = hp0 + rowP * HWD, offM = hp0 + rowM * HWD;
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ug), 0, KCH * U_CHUNK, 0x00020000);
    const int voU0 = ti * 4 * U_XI + (li * 16 + kh * 8) * 2, voU1 = voU0 + 32 * 16 * 2;
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};
    // column c (global index: chunk = c / 4, j = c % 4) -> register set c % (UD + 1)
    f32x4 us[UD + 1][3][2];
    auto load_u = [&](int c, f32x4 (&dst)[3][2]) {
        const int so = ((c >> 2) % KCH) * U_CHUNK + (c & 3) * U_XI;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            dst[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voU0, so + p * (64 * 16 * 2), 0));
            dst[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, voU1, so + p * (64 * 16 * 2), 0));
        }
    };
#pragma unroll
    for (int c = 0; c < UD; ++c) load_u(c, us[c]);
    const int ncol = chunks * 4;
    // the column loop is unrolled over (UD + 1) * 4 / gcd columns so that register sets are static: walk in groups of
    // lcm(4, UD + 1) columns
    constexpr int G = (UD + 1) % 4 == 0 ? (UD + 1) : ((UD + 1) % 2 == 0 ? 2 * (UD + 1) : 4 * (UD + 1));
    f32x4 t[MT][2][4];
    for (int c0 = 0; c0 < ncol; c0 += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = c0 + g, j = g & 3;
            if (j == 0) {
                if (VAR != 9) __syncthreads();                           // raw buffer hand-over (one barrier per chunk)
                const f32x4* r = sR + (((c0 + g) >> 2) & 1) * RAW + 2 * kh * HP;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            t[m][q][b] = r[q * HP + offP + m * 4 * HWD + b] + r[q * HP + offM + m * 4 * HWD + b] * sgn;
            }
            load_u(c + UD, us[(g + UD) % (UD + 1)]);
            const f32x4 (&uf)[3][2] = us[g % (UD + 1)];
            f32x4 pl[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    v[q] = (j == 0) ? t[m][q][0] - t[m][q][2] : (j == 1) ? t[m][q][1] + t[m][q][2]
                         : (j == 2) ? t[m][q][2] - t[m][q][1] : t[m][q][1] - t[m][q][3];
                split8(v[0], v[1], pl[m]);
            }
#pragma unroll
            for (int e = 0; e < 6; ++e)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[j][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, uf[TB[e]][n]), __builtin_bit_cast(bf16x8, pl[m][TA[e]]), acc[j][m][n], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[j][m][n][r];
    if (s == 12345.678f) out[0] = s;
}

template <int MT, int WPS, int UD, int KCH, int VAR>
void run(const char* name, const float* ug, float* d) {
    constexpr int HP = (4 * MT + 2) * HWD;
    int bytes = 2 * 4 * HP * 16;
    const int floorB = WPS == 1 ? 90000 : 70000;          // pin the residency (160 KB of LDS per CU)
    if (bytes < floorB) bytes = floorB;
    (void)hipFuncSetAttribute((const void*)&probe<MT, WPS, UD, KCH, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const int chunks = 384, blocks = 256 * WPS;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MT, WPS, UD, KCH, VAR><<<blocks, 256, bytes>>>(ug, d, 12);
    (void)hipEventRecord(e0);
    probe<MT, WPS, UD, KCH, VAR><<<blocks, 256, bytes>>>(ug, d, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ideal = (double)chunks * 48 * MT * 32 * WPS / 2.4e9 * 1e3;      // ms at 2.4 GHz
    printf("%-74s %.3f ms  mfma-ideal %.3f ms -> pipe %.2f (%s)\n", name, ms, ideal, ideal / ms, hipGetErrorString(hipGetLastError()));
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    const size_t ub = (size_t)32 * U_CHUNK;
    unsigned short* hu = (unsigned short*)malloc(ub);
    unsigned h = 12345u;
    for (size_t i = 0; i < ub / 2; ++i) { h = h * 1664525u + 1013904223u; hu[i] = (unsigned short)((h >> 16 & 0x807f) | 0x3f00); }   // bf16 ~ +-1
    float* ug; (void)hipMalloc(&ug, ub); (void)hipMemcpy(ug, hu, ub, hipMemcpyHostToDevice);
    printf("U fragments from global memory (buffer loads, L2), raw halo in LDS, one barrier per chunk; random operands\n");
    run<1, 2, 1, 4, 0>("product shape: 8 acc, 2 waves/SIMD, U one column ahead, K = 64", ug, d);
    run<1, 2, 1, 32, 0>("product shape: 8 acc, 2 waves/SIMD, U one column ahead, K = 512", ug, d);
    run<1, 2, 2, 32, 0>("8 acc, 2 waves/SIMD, U two columns ahead, K = 512", ug, d);
    run<1, 1, 1, 32, 0>("8 acc, ONE wave/SIMD (a lone product wave), U one column ahead, K = 512", ug, d);
    run<1, 1, 2, 32, 0>("8 acc, ONE wave/SIMD, U two columns ahead, K = 512", ug, d);
    run<1, 1, 3, 32, 0>("8 acc, ONE wave/SIMD, U three columns ahead, K = 512", ug, d);
    run<1, 1, 3, 4, 0>("8 acc, ONE wave/SIMD, U three columns ahead, K = 64", ug, d);
    run<1, 1, 3, 32, 9>("8 acc, ONE wave/SIMD, U three columns ahead, K = 512, no barrier", ug, d);
    run<1, 1, 5, 32, 0>("8 acc, ONE wave/SIMD, U five columns ahead, K = 512", ug, d);
    run<2, 1, 1, 4, 0>("16 acc, ONE wave/SIMD, U one column ahead, K = 64", ug, d);
    run<2, 1, 1, 32, 0>("16 acc, ONE wave/SIMD, U one column ahead, K = 512", ug, d);
    run<2, 1, 2, 32, 0>("16 acc, ONE wave/SIMD, U two columns ahead, K = 512", ug, d);
    run<2, 1, 3, 32, 0>("16 acc, ONE wave/SIMD, U three columns ahead, K = 512", ug, d);
    run<2, 1, 3, 4, 0>("16 acc, ONE wave/SIMD, U three columns ahead, K = 64", ug, d);
    run<2, 1, 3, 32, 9>("16 acc, ONE wave/SIMD, U three columns ahead, K = 512, no barrier", ug, d);
    return 0;
}
