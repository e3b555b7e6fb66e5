// minibatch_stddev_layer for gfx950 (forward with fused concat, and backward).
//
// Behavioural contract: training/networks_stylegan2.py:132-144 with
// num_new_features == 1: the minibatch is viewed as [G, M, C, H, W] (sample
// n = g*M + m), stddev over g with eps 1e-8, mean over (c,h,w) -> stat[m], tiled
// back to [N, 1, H, W] (sample n receives stat[n % M]) and appended as channel C.
// Design: tensors are channel-minor [N, H, W, C]; the H*W*C positions of a group m are cut into slices, one
// workgroup each (the layer is 8192 positions x 6 samples: with one workgroup per group it ran on 4 of the 256
// CUs as a chain of dependent loads, 57 + 106 us forward + backward), each lane holding the G samples of its
// position in registers (two-pass mean / variance like the reference); a wave-shuffle + LDS tree gives the
// slice's partial statistic, and the consumer adds the slices in fixed order.  A second kernel writes the
// concatenated [N,H,W,C+1] output (row stride C+1 is odd, so scalar accesses).
#include "igan_common.h"

namespace {

constexpr int MB_MAXG = 16;

__device__ __forceinline__ float block_sum_256(float s, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

constexpr int MB_MAXS = 64;     // slices of the position axis

__host__ __device__ inline int mb_slices(int P) { return P >= 256 * MB_MAXS ? MB_MAXS : (P + 255) / 256; }

// partial[m][slice] = sum over the slice's positions of sqrt(var_g + 1e-8)
__global__ __launch_bounds__(256) void mbstd_stat_kernel(const float* x, float* partial, int M, int G, int P /*H*W*C*/) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const int S = gridDim.y;
    const int per = (P + S - 1) / S;
    const int p0 = blockIdx.y * per, p1 = min(p0 + per, P);
    float s = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        float v[MB_MAXG];
        float mean = 0.f;
        for (int g = 0; g < G; g++) {
            v[g] = x[((long long)(g * M + m)) * P + p];
            mean += v[g];
        }
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; g++) {
            const float d = v[g] - mean;
            var += d * d;
        }
        var /= (float)G;
        s += sqrtf(var + 1e-8f);
    }
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[m * S + blockIdx.y] = tot;
}

__device__ __forceinline__ float mb_stat(const float* partial, int m, int S, int P) {
    float s = 0.f;
    for (int k = 0; k < S; k++) s += partial[m * S + k];       // fixed order
    return s / (float)P;
}

__global__ __launch_bounds__(256) void mbstd_concat_kernel(const float* x, const float* partial, float* y, int N, int HW, int C, int M, int S) {
    const long long total = (long long)N * HW * (C + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (C + 1));
        const long long pix = i / (C + 1);
        const int n = (int)(pix / HW);
        y[i] = (c < C) ? x[pix * C + c] : mb_stat(partial, n % M, S, HW * C);
    }
}

// dstat[m] = sum over the group's samples and pixels of dy[..., C] ; then
// dx[n,p] = dy[n,p(c<C)] + dstat[m] * (x - mean) / (G * sigma * P).
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const float* x, const float* dy, float* dx, int M, int G, int HW, int C) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const int P = HW * C;
    const int S = gridDim.y;
    const int per = (P + S - 1) / S;
    const int p0 = blockIdx.y * per, p1 = min(p0 + per, P);
    float s = 0.f;
    for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {      // every slice recomputes the group's dstat (G*HW values)
        const int g = i / HW;
        const int pix = i - g * HW;
        s += dy[(((long long)(g * M + m)) * HW + pix) * (C + 1) + C];
    }
    const float dstat = block_sum_256(s, red);
    const float scale = dstat / ((float)G * (float)P);
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const int pix = p / C;
        const int c = p - pix * C;
        float v[MB_MAXG];
        float mean = 0.f;
        for (int g = 0; g < G; g++) {
            v[g] = x[((long long)(g * M + m)) * P + p];
            mean += v[g];
        }
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; g++) {
            const float d = v[g] - mean;
            var += d * d;
        }
        var /= (float)G;
        const float inv_sigma = 1.0f / sqrtf(var + 1e-8f);
        for (int g = 0; g < G; g++) {
            const long long n = g * M + m;
            const float pass = dy[(n * HW + pix) * (C + 1) + c];
            dx[n * P + p] = pass + scale * (v[g] - mean) * inv_sigma;
        }
    }
}

int check_args(int N, int H, int W, int C, int G) {
    IGAN_REQUIRE(N >= 1 && H >= 1 && W >= 1 && C >= 1, "mbstd: dims must be positive");
    IGAN_REQUIRE(G >= 1 && G <= MB_MAXG, "mbstd: group size must be in [1,%d]", MB_MAXG);
    IGAN_REQUIRE(N % G == 0, "mbstd: minibatch must be divisible by group size");
    IGAN_REQUIRE((long long)N * H * W * (C + 1) <= INT32_MAX, "mbstd: tensor too large");
    return IGAN_OK;
}

}  // namespace

extern "C" size_t igan_mbstd_workspace_floats(int N, int H, int W, int C, int G) {
    if (N < 1 || G < 1 || N % G != 0 || H < 1 || W < 1 || C < 1) return 0;
    return (size_t)(N / G) * mb_slices(H * W * C);
}

extern "C" int igan_mbstd_fwd(igan_stream_t stream_, const float* x, float* y, float* workspace,
                              int N, int H, int W, int C, int G) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(x && y && workspace, "mbstd: null buffer");
    if (int rc = check_args(N, H, W, C, G)) return rc;
    const int M = N / G;
    const int HW = H * W;
    const int S = mb_slices(HW * C);
    hipLaunchKernelGGL(mbstd_stat_kernel, dim3(M, S), dim3(256), 0, stream, x, workspace, M, G, HW * C);
    const long long total = (long long)N * HW * (C + 1);
    const int grid = (int)std::min<long long>(ceil_div_ll(total, 256), 2048);
    hipLaunchKernelGGL(mbstd_concat_kernel, dim3(grid), dim3(256), 0, stream, x, (const float*)workspace, y, N, HW, C, M, S);
    IGAN_LAUNCH_CHECK("mbstd_fwd launch");
    return IGAN_OK;
}

extern "C" int igan_mbstd_bwd(igan_stream_t stream_, const float* x, const float* dy, float* dx,
                              int N, int H, int W, int C, int G) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(x && dy && dx, "mbstd: null buffer");
    if (int rc = check_args(N, H, W, C, G)) return rc;
    const int M = N / G;
    hipLaunchKernelGGL(mbstd_bwd_kernel, dim3(M, mb_slices(H * W * C)), dim3(256), 0, stream, x, dy, dx, M, G, H * W, C);
    IGAN_LAUNCH_CHECK("mbstd_bwd launch");
    return IGAN_OK;
}
