// Hardware characterisation (gfx950) for the Winograd x six-product tile: can the transform + three-plane bf16 split of
// V (~5 VALU per MFMA) hide under v_mfma_f32_32x32x16_bf16 when the MFMAs are fed from LDS (raw halo + U fragments)?
// One "chunk" of a wave = 16 input channels of one transform row i: 16 raw ds_read_b128, 32 VALU (row transform),
// then per column j: 8 VALU (column transform) + 44 VALU (split) + 6 U reads + 12*MT MFMAs.
//   WAVES = 4 (256 threads, two workgroups per CU) or 8 (512 threads, one workgroup per CU): both two waves per SIMD;
//   MT = 2 with WAVES = 4 and one workgroup per CU = one wave per SIMD, 16 accumulators.
//   VAR: 0 = everything, 1 = no split / transform VALU (operands straight from LDS), 2 = VALU + reads only (no MFMA)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino6_loop tools/probes/wino6_loop.hip && /tmp/wino6_loop
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

// The same three planes with v_dot2c_f32_bf16 (gfx950: D += A.lo * B.lo + A.hi * B.hi on packed bf16): the residual of a
// pair is taken straight from the PACKED plane -- r_a = a + dot2(h, {-1, 0}), r_b = b + dot2(h, {0, -1}) -- so the two
// unpack instructions per value (shift / mask) disappear: 7 instead of 11 VALU per pair of values.  The difference is
// exactly representable in fp32, so any correctly aligned adder returns it exactly (checked below on 2^26 values).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float sub_lo(float a, unsigned h) {
    const bf16x2 k = {(__bf16)-1.0f, (__bf16)0.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), k, a, false);
}
__device__ __forceinline__ float sub_hi(float b, unsigned h) {
    const bf16x2 k = {(__bf16)0.0f, (__bf16)-1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), k, b, false);
}
__device__ __forceinline__ void split8_dot2(const f32x4& v0, const f32x4& v1, f32x4 (&pl)[3]) {
    unsigned p[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = k < 2 ? v0[2 * k] : v1[2 * k - 4], b = k < 2 ? v0[2 * k + 1] : v1[2 * k - 3];
        const unsigned h = cvt_pk(a, b);
        a = sub_lo(a, h); b = sub_hi(b, h);
        const unsigned m = cvt_pk(a, b);
        a = sub_lo(a, m); b = sub_hi(b, m);
        p[0][k] = h; p[1][k] = m; p[2][k] = cvt_pk(a, b);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) pl[q] = __builtin_bit_cast(f32x4, *reinterpret_cast<uint4*>(p[q]));
}

// exactness: both splits on n pseudo-random fp32 values of every exponent; counts planes that differ in any bit
__global__ void split_check(unsigned long long* bad, unsigned seed, int mode) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    float w[8];
    for (int k = 0; k < 8; ++k) {
        h = h * 1664525u + 1013904223u;
        unsigned bits = h;
        if (mode == 1) bits = (h & 0x807fffffu) | ((100u + (h >> 23) % 56u) << 23);    // 2^-27 .. 2^28: the tile's range
        if (mode == 2) bits = (h & 0x80ffffffu);                                          // denormals and the smallest normals
        float v = __uint_as_float(bits);
        if (v != v || fabsf(v) > 3e38f) v = 1.f;
        w[k] = v;
    }
    f32x4 a[3], b[3];
    split8(f32x4{w[0], w[1], w[2], w[3]}, f32x4{w[4], w[5], w[6], w[7]}, a);
    split8_dot2(f32x4{w[0], w[1], w[2], w[3]}, f32x4{w[4], w[5], w[6], w[7]}, b);
    int n = 0;
    for (int q = 0; q < 3; ++q)
        for (int k = 0; k < 4; ++k) n += __float_as_uint(a[q][k]) != __float_as_uint(b[q][k]);
    if (n) atomicAdd(bad, (unsigned long long)n);
}

constexpr int HWD = 34;
template <int WAVES, int MT, int VAR, int WPS, int NSLOT>
__global__ __launch_bounds__(WAVES * 64, WPS) void probe(float* out, int chunks) {
    extern __shared__ f32x4 lds[];
    constexpr int HP = (4 * MT * (WAVES / 4) + 2) * HWD;
    constexpr int RAW = 4 * HP;                 // [quad][pixel]
    constexpr int USLOT = 4 * 3 * 2 * 64;       // [i][plane][nt][32 x 2] 16-byte fragments of one column j
    f32x4* sR = lds;                            // 2 buffers
    f32x4* sU = lds + 2 * RAW;                  // NSLOT column slots
    for (int i = threadIdx.x; i < 2 * RAW + NSLOT * USLOT; i += WAVES * 64) {
        unsigned h = i * 2654435761u + blockIdx.x * 40503u;
        float w[4];
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            if (i < 2 * RAW) w[k] = ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));       // fp32 in [-1, 1)
            else w[k] = __uint_as_float((h & 0x807f807fu) | 0x3f003f00u);                    // bf16 pairs ~1
        }
        lds[i] = f32x4{w[0], w[1], w[2], w[3]};
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kh = lane >> 5;
    const int ti = wave & 3, pt = wave >> 2;
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
    const int hp0 = (pt * 4 * MT + 2 * (li >> 4)) * HWD + 2 * (li & 15);
    const int offP = hp0 + rowP * HWD, offM = hp0 + rowM * HWD;
    const f32x4* pU = sU + ti * (3 * 2 * 64) + li * 2 + kh;
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1};
    constexpr int TB[6] = {0, 1, 0, 2, 0, 1};
    for (int ch = 0; ch < chunks; ++ch) {
        const f32x4* r = sR + (ch & 1) * RAW + 2 * kh * HP;
        f32x4 t[MT][2][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    t[m][q][b] = r[q * HP + offP + m * 4 * HWD + b] + r[q * HP + offM + m * 4 * HWD + b] * sgn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 uf[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int n = 0; n < 2; ++n) uf[p][n] = pU[(j % NSLOT) * USLOT + (p * 2 + n) * 64];
            f32x4 pl[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    v[q] = (j == 0) ? t[m][q][0] - t[m][q][2] : (j == 1) ? t[m][q][1] + t[m][q][2]
                         : (j == 2) ? t[m][q][2] - t[m][q][1] : t[m][q][1] - t[m][q][3];
                if (VAR == 1) { pl[m][0] = v[0]; pl[m][1] = v[1]; pl[m][2] = t[m][0][j]; }
                else if (VAR == 3) split8_dot2(v[0], v[1], pl[m]);
                else split8(v[0], v[1], pl[m]);
            }
            if (VAR == 2) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) acc[j][m][0][p] += pl[m][p][0] + uf[p][0][1] + uf[p][1][2];
            } else {
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[j][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, uf[TB[e]][n]), __builtin_bit_cast(bf16x8, pl[m][TA[e]]),
                                acc[j][m][n], 0, 0, 0);
            }
        }
        __syncthreads();
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

template <int WAVES, int MT, int VAR, int WPS, int NSLOT>
void run(const char* name, float* d) {
    constexpr int HP = (4 * MT * (WAVES / 4) + 2) * HWD;
    const int wgPerCu = (WPS * 4) / WAVES;
    int bytes = (2 * 4 * HP + NSLOT * 4 * 3 * 2 * 64) * 16;
    const int floorB = wgPerCu == 1 ? 90000 : 70000;   // pin the residency (160 KB of LDS per CU)
    if (bytes < floorB) bytes = floorB;
    hipFuncSetAttribute((const void*)&probe<WAVES, MT, VAR, WPS, NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const int chunks = 400, blocks = 256 * wgPerCu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<WAVES, MT, VAR, WPS, NSLOT><<<blocks, WAVES * 64, bytes>>>(d, 4);
    hipEventRecord(e0);
    probe<WAVES, MT, VAR, WPS, NSLOT><<<blocks, WAVES * 64, bytes>>>(d, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ideal = (double)chunks * 48 * MT * 32 * WPS / 2.4e9 * 1e3;      // ms at 2.4 GHz
    printf("%-52s lds %6d B: %.3f ms  mfma-ideal %.3f ms -> pipe %.2f (%s)\n", name, bytes, ms, ideal, ideal / ms,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float* d; hipMalloc(&d, 4);
    {
        unsigned long long* bad; hipMalloc(&bad, 8);
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(bad, 0, 8);
            for (int r = 0; r < 8; ++r) split_check<<<4096, 256>>>(bad, 977u * r + 13u, mode);
            unsigned long long hb; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
            printf("split via v_dot2c_f32_bf16 vs shift/mask/sub, %s: %llu of %llu plane words differ\n",
                   mode == 0 ? "any finite fp32" : mode == 1 ? "2^-27..2^28" : "denormals / smallest normals", hb,
                   8ull * 4096 * 256 * 12);
        }
    }
    run<4, 1, 3, 2, 1>("4 waves x2 WG/CU, full, dot2c split", d);
    run<4, 1, 0, 2, 1>("4 waves x2 WG/CU, full", d);
    run<4, 1, 1, 2, 1>("4 waves x2 WG/CU, no VALU", d);
    run<4, 1, 2, 2, 1>("4 waves x2 WG/CU, no MFMA", d);
    run<8, 1, 0, 2, 4>("8 waves x1 WG/CU, full", d);
    run<8, 1, 1, 2, 4>("8 waves x1 WG/CU, no VALU", d);
    run<8, 1, 2, 2, 4>("8 waves x1 WG/CU, no MFMA", d);
    run<4, 2, 0, 1, 4>("4 waves x1 WG/CU, 16 acc (1 wave/SIMD), full", d);
    run<4, 2, 1, 1, 4>("4 waves x1 WG/CU, 16 acc (1 wave/SIMD), no VALU", d);
    run<4, 2, 2, 1, 4>("4 waves x1 WG/CU, 16 acc (1 wave/SIMD), no MFMA", d);
    return 0;
}
