// Callers either side of the hot path (SURVEY.md section 8f "next" rows 1-3):
//   * event voxelisation     basicsr/data/event_util.py:6-66  (events_to_voxel_grid)
//   * validation tail        basicsr/utils/img_util.py:90-117 (tensor2img quantisation) +
//                            basicsr/metrics/psnr_ssim.py:48-63 (calculate_psnr, float64 MSE)
//   * tile overlap-averaging basicsr/models/twoImage_event_recurrent_model.py:252-268 (grids_inverse)
// HBM-bound streaming kernels; the scalar sums (PSNR / SSIM) are two-stage and deterministic, only the event scatter-add
// uses (fp32) atomics, as np.add.at's order is unspecified too.
#include "common.h"

namespace {

// voxel[(ti) * H*W + y*W + x] += pol*(1-dt) ; voxel[(ti+1)...] += pol*dt   (bilinear in time)
__global__ __launch_bounds__(256) void voxel_kernel(const double* __restrict__ ts, const int* __restrict__ xs,
                                                   const int* __restrict__ ys, const float* __restrict__ ps,
                                                   long long n, int bins, int W, int H, double first, double deltaT,
                                                   float* __restrict__ voxel) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double t = (double)(bins - 1) * (ts[i] - first) / deltaT;       // event_util.py:36
        const int ti = (int)t;                                                // astype(int): truncation
        const double dt = t - (double)ti;
        double pol = (double)ps[i];
        if (pol == 0.0) pol = -1.0;                                           // event_util.py:41
        const int x = xs[i], y = ys[i];
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        const long long base = (long long)y * W + x;
        if (ti >= 0 && ti < bins) atomicAdd(voxel + (long long)ti * W * H + base, (float)(pol * (1.0 - dt)));
        if (ti + 1 >= 0 && ti + 1 < bins) atomicAdd(voxel + (long long)(ti + 1) * W * H + base, (float)(pol * dt));
    }
}

__device__ __forceinline__ float quant255(float v) {          // tensor2img: clamp [0,1], *255, round (half to even)
    v = fminf(fmaxf(v, 0.f), 1.f);
    return rintf(v * 255.f);
}

// sq[f][chunk] = sum over the chunk's share of the frame of (q(a) - q(b))^2 ; grid = (chunks, frames)
__global__ __launch_bounds__(256) void sqerr_u8_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      long long frame_elems, double* __restrict__ sq) {
    __shared__ double sh[4];
    const long long f = blockIdx.y;
    const float* pa = a + f * frame_elems;
    const float* pb = b + f * frame_elems;
    double acc = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < frame_elems; i += (long long)gridDim.x * 256) {
        const float d = quant255(pa[i]) - quant255(pb[i]);
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sq[f * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];   // partial [frame][chunk]
}

// acc[c][i0+y][j0+x] += tile[c][y][x] ; cnt[i0+y][j0+x] += 1     (one tile)
__global__ __launch_bounds__(256) void tile_add_kernel(const float* __restrict__ tile, float* __restrict__ acc,
                                                      float* __restrict__ cnt, int C, int th, int tw, int H, int W,
                                                      int i0, int j0) {
    const long long total = (long long)C * th * tw;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int x = e % tw; long long r = e / tw;
        const int y = r % th; const int c = r / th;
        const long long d = ((long long)c * H + i0 + y) * W + j0 + x;
        acc[d] += tile[e];
        if (c == 0) cnt[(long long)(i0 + y) * W + j0 + x] += 1.f;
    }
}

__global__ __launch_bounds__(256) void tile_norm_kernel(float* __restrict__ acc, const float* __restrict__ cnt, int C,
                                                       long long HW) {
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < C * HW; e += (long long)gridDim.x * 256)
        acc[e] /= cnt[e % HW];
}

// ---- SSIM of the reference's _ssim_3d (metrics/psnr_ssim.py:135-182): 11x11x11 separable Gaussian
// (sigma 1.5) over (H, W, C=3) with REPLICATE padding in all three axes, fp32, on the uint8-quantised
// frames; ssim map mean over H*W*3.  One block = 16x16 output pixels x 3 channels.
struct SsimConst { float g[11]; float mc[3][3]; };

__global__ __launch_bounds__(256) void ssim3d_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    int H, int W, const SsimConst k, double* __restrict__ out) {
    constexpr int T = 16, R = 5, HS = T + 2 * R;          // 26
    __shared__ float sA[3][HS][HS + 1], sB[3][HS][HS + 1];
    __shared__ float sH[5][3][HS][T + 1];
    __shared__ double sred[4];
    const int f = blockIdx.z;
    const int y0 = blockIdx.y * T, x0 = blockIdx.x * T;
    const float* pa = a + (long long)f * 3 * H * W;
    const float* pb = b + (long long)f * 3 * H * W;
    for (int e = threadIdx.x; e < 3 * HS * HS; e += 256) {
        const int c = e / (HS * HS), r = (e / HS) % HS, q = e % HS;
        const int y = min(max(y0 + r - R, 0), H - 1), x = min(max(x0 + q - R, 0), W - 1);   // replicate
        sA[c][r][q] = quant255(pa[((long long)c * H + y) * W + x]);
        sB[c][r][q] = quant255(pb[((long long)c * H + y) * W + x]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * HS * T; e += 256) {          // horizontal pass, 5 fields
        const int c = e / (HS * T), r = (e / T) % HS, q = e % T;
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const float va = sA[c][r][q + j], vb = sB[c][r][q + j], g = k.g[j];
            m1 += g * va; m2 += g * vb; s11 += g * va * va; s22 += g * vb * vb; s12 += g * va * vb;
        }
        sH[0][c][r][q] = m1; sH[1][c][r][q] = m2; sH[2][c][r][q] = s11; sH[3][c][r][q] = s22; sH[4][c][r][q] = s12;
    }
    __syncthreads();
    const int ty = threadIdx.x / T, tx = threadIdx.x % T;
    float v[5][3];
#pragma unroll
    for (int fi = 0; fi < 5; ++fi)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 11; ++j) s += k.g[j] * sH[fi][c][ty + j][tx];
            v[fi][c] = s;
        }
    double acc = 0.0;
    if (y0 + ty < H && x0 + tx < W) {
        const float C1 = (0.01f * 255.f) * (0.01f * 255.f), C2 = (0.03f * 255.f) * (0.03f * 255.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {                               // channel-axis pass (replicate-padded)
            float w[5];
#pragma unroll
            for (int fi = 0; fi < 5; ++fi) w[fi] = k.mc[c][0] * v[fi][0] + k.mc[c][1] * v[fi][1] + k.mc[c][2] * v[fi][2];
            const float mu1 = w[0], mu2 = w[1];
            const float s1 = w[2] - mu1 * mu1, s2 = w[3] - mu2 * mu2, s12 = w[4] - mu1 * mu2;
            acc += (double)(((2.f * mu1 * mu2 + C1) * (2.f * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)                                           // partial [frame][tile row][tile column]
        out[((long long)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

int nb(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

extern "C" int refid_events_to_voxel(const double* ts, const int* xs, const int* ys, const float* ps,
                                     long long n_events, int num_bins, int width, int height, double first_stamp,
                                     double last_stamp, float* voxel, void* stream) {
    REFID_CHECK(ts && xs && ys && ps && voxel && n_events > 0 && num_bins > 0 && width > 0 && height > 0,
                "events_to_voxel: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(voxel, 0, sizeof(float) * (size_t)num_bins * width * height, st);
    REFID_CHECK(e == hipSuccess, "events_to_voxel: memset failed: %s", hipGetErrorString(e));
    double deltaT = last_stamp - first_stamp;
    if (deltaT == 0) deltaT = 1.0;                                            // event_util.py:33-34
    hipLaunchKernelGGL(voxel_kernel, dim3(nb(n_events)), dim3(256), 0, st, ts, xs, ys, ps, n_events, num_bins, width,
                       height, first_stamp, deltaT, voxel);
    REFID_LAUNCH_CHECK("events_to_voxel");
    return 0;
}

static int sqerr_chunks(long long frame_elems) {
    const int c = nb(frame_elems);
    return c > 256 ? 256 : c;
}

extern "C" int refid_sqerr_u8_parts(int n_frames, long long frame_elems) {
    return (n_frames > 0 && frame_elems > 0) ? n_frames * sqerr_chunks(frame_elems) : 0;
}

extern "C" int refid_sqerr_u8(const float* a, const float* b, int n_frames, long long frame_elems, double* sq,
                              double* parts, void* stream) {
    REFID_CHECK(a && b && sq && parts && n_frames > 0 && frame_elems > 0, "sqerr_u8: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = sqerr_chunks(frame_elems);
    hipLaunchKernelGGL(sqerr_u8_kernel, dim3(chunks, n_frames), dim3(256), 0, st, a, b, frame_elems, parts);
    REFID_LAUNCH_CHECK("sqerr_u8");
    return refid_launch_sum_rows_f64(parts, n_frames, chunks, sq, st);
}

extern "C" int refid_tile_add(const float* tile, float* acc, float* cnt, int c, int th, int tw, int h, int w, int i0,
                              int j0, void* stream) {
    REFID_CHECK(tile && acc && cnt && c > 0 && th > 0 && tw > 0 && i0 >= 0 && j0 >= 0 && i0 + th <= h && j0 + tw <= w,
                "tile_add: tile (%d,%d)+(%d,%d) outside %dx%d", i0, j0, th, tw, h, w);
    hipLaunchKernelGGL(tile_add_kernel, dim3(nb((long long)c * th * tw)), dim3(256), 0, (hipStream_t)stream, tile, acc,
                       cnt, c, th, tw, h, w, i0, j0);
    REFID_LAUNCH_CHECK("tile_add");
    return 0;
}

extern "C" int refid_tile_normalize(float* acc, const float* cnt, int c, int h, int w, void* stream) {
    REFID_CHECK(acc && cnt && c > 0 && h > 0 && w > 0, "tile_normalize: bad arguments");
    hipLaunchKernelGGL(tile_norm_kernel, dim3(nb((long long)c * h * w)), dim3(256), 0, (hipStream_t)stream, acc, cnt, c,
                       (long long)h * w);
    REFID_LAUNCH_CHECK("tile_normalize");
    return 0;
}

extern "C" int refid_ssim3d_u8_parts(int n_frames, int h, int w) {
    return (n_frames > 0 && h > 0 && w > 0) ? n_frames * cdiv(w, 16) * cdiv(h, 16) : 0;
}

extern "C" int refid_ssim3d_u8(const float* a, const float* b, int n_frames, int h, int w, double* sum_out,
                               double* parts, void* stream) {
    REFID_CHECK(a && b && sum_out && parts && n_frames > 0 && h > 0 && w > 0, "ssim3d_u8: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    SsimConst k;
    double g[11], s = 0.0;                                  // cv2.getGaussianKernel(11, 1.5)
    for (int i = 0; i < 11; ++i) { g[i] = exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < 11; ++i) { g[i] /= s; k.g[i] = (float)g[i]; }
    for (int c = 0; c < 3; ++c) {                           // 11 taps along the 3-channel axis, replicate padding
        double m[3] = {0, 0, 0};
        for (int j = 0; j < 11; ++j) { int cc = c + j - 5; cc = cc < 0 ? 0 : (cc > 2 ? 2 : cc); m[cc] += g[j]; }
        for (int q = 0; q < 3; ++q) k.mc[c][q] = (float)m[q];
    }
    dim3 grid(cdiv(w, 16), cdiv(h, 16), n_frames);
    hipLaunchKernelGGL(ssim3d_kernel, grid, dim3(256), 0, st, a, b, h, w, k, parts);
    REFID_LAUNCH_CHECK("ssim3d_u8");
    return refid_launch_sum_rows_f64(parts, n_frames, (int)(grid.x * grid.y), sum_out, st);
}
