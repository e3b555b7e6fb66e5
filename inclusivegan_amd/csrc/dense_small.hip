// Small-batch dense layers on gfx950: y[M,N] = x[M,K] . w  for M <= 32 rows (the mapping network and the
// per-layer style / demodulation matmuls, networks_stylegan2.py:41-46,107,117: 38 of them per generator
// pass at batch 3..24), and their data / weight gradients.
//
// These are not MFMA work: 12 MFLOP against a 1 MB weight matrix.  On the implicit-GEMM path a call
// cost a 32x128-tile kernel cut 8 ways along K plus a reduce kernel (about 13 us of launches for
// 0.3 us of memory traffic).  Here one launch streams the weight matrix once:
//   * a workgroup owns 8 output channels; its 256 lanes are 32 reduction groups x 8 channels, every
//     group takes 4 consecutive k out of each 128 (so the x operand is one ds_read_b128 per row,
//     broadcast to the 8 channel lanes, and the weights are 32 B (w[k][n]) or 16 B-per-lane (w[n][k])
//     contiguous pieces);
//   * x (at most 32 x 256 floats per tile) is staged in LDS once per workgroup, rows beyond M zero;
//   * the 32 group partials meet in LDS and are added in fixed order: bit-reproducible, no atomics.
// The weight gradient is an M-term outer product per element: one thread per 4 output channels.
// Dispatched from igan_conv2d / igan_conv2d_wgrad when the geometry is 1x1 on a 1x1 map without
// scales; everything else stays on the MFMA kernels.
#include "igan_common.h"

namespace {

constexpr int DS_COLS = 8;                   // output channels per workgroup
constexpr int DS_GROUPS = 32;                // reduction groups per workgroup
constexpr int DS_KT = 256;                   // reduction tile staged in LDS (floats)

template <int MB, bool WT>
__global__ __launch_bounds__(256) void dense_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int M, int K, int N, float alpha) {
    constexpr int RSTR = MB * DS_COLS + 8;   // group stride in the partial-sum image (bank-spread)
    __shared__ __attribute__((aligned(16))) float xs[MB * DS_KT];
    __shared__ float red[DS_GROUPS * RSTR];
    const int tid = threadIdx.x, c = tid & (DS_COLS - 1), g = tid / DS_COLS;
    const int j = blockIdx.x * DS_COLS + c;
    const bool jok = j < N;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; m++) acc[m] = 0.f;

    for (int k0 = 0; k0 < K; k0 += DS_KT) {
        for (int v = tid; v < MB * (DS_KT / 4); v += 256) {
            const int m = v / (DS_KT / 4), kv = v - m * (DS_KT / 4);
            const int k = k0 + 4 * kv;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && k < K) t = *reinterpret_cast<const float4*>(x + (size_t)m * K + k);
            *reinterpret_cast<float4*>(xs + m * DS_KT + 4 * kv) = t;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < DS_KT / (4 * DS_GROUPS); it++) {
            const int i = (it * DS_GROUPS + g) * 4;
            const int k = k0 + i;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jok && k < K) {      // K % 4 == 0: the four k are in range together
                if constexpr (WT) wv = *reinterpret_cast<const float4*>(w + (size_t)j * K + k);
                else {
                    wv.x = w[(size_t)(k + 0) * N + j]; wv.y = w[(size_t)(k + 1) * N + j];
                    wv.z = w[(size_t)(k + 2) * N + j]; wv.w = w[(size_t)(k + 3) * N + j];
                }
            }
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + m * DS_KT + i);
                acc[m] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[m]))));
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MB; m++) red[g * RSTR + m * DS_COLS + c] = acc[m];
    __syncthreads();
    for (int o = tid; o < MB * DS_COLS; o += 256) {
        const int m = o / DS_COLS, cc = o - m * DS_COLS;
        float s = 0.f;
#pragma unroll 8
        for (int gg = 0; gg < DS_GROUPS; gg++) s += red[gg * RSTR + o];
        const int jj = blockIdx.x * DS_COLS + cc;
        if (m < M && jj < N) y[(size_t)m * N + jj] = s * alpha;
    }
}

// dw[k][n] = sum_m x[m][k] * dy[m][n]
__global__ __launch_bounds__(256) void dense_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dw, int M, int K, int N, float alpha) {
    const int nv = N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= K * nv) return;
    const int k = idx / nv, n4 = idx - k * nv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; m++) {
        const float xv = x[(size_t)m * K + k];
        const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)m * N + 4 * n4);
        acc.x = fmaf(xv, d.x, acc.x); acc.y = fmaf(xv, d.y, acc.y);
        acc.z = fmaf(xv, d.z, acc.z); acc.w = fmaf(xv, d.w, acc.w);
    }
    *reinterpret_cast<float4*>(dw + (size_t)k * N + 4 * n4) = make_float4(acc.x * alpha, acc.y * alpha, acc.z * alpha, acc.w * alpha);
}

template <int MB>
void launch_dense(hipStream_t stream, const float* x, const float* w, float* y, int M, int K, int N, bool wt, float alpha) {
    const dim3 grid(igan::ceil_div(N, DS_COLS));
    if (wt) hipLaunchKernelGGL((dense_small_kernel<MB, true>), grid, dim3(256), 0, stream, x, w, y, M, K, N, alpha);
    else hipLaunchKernelGGL((dense_small_kernel<MB, false>), grid, dim3(256), 0, stream, x, w, y, M, K, N, alpha);
}

}  // namespace

namespace igan {

bool dense_small_ok(int M, int K, const void* x, const void* w, bool wt) {
    return M >= 1 && M <= 32 && (K % 4) == 0 && (((uintptr_t)x) & 15) == 0 && (!wt || (((uintptr_t)w) & 15) == 0);
}

int dense_small_rows(int M) { return M <= 8 ? 8 : (M <= 16 ? 16 : 32); }

void dense_small(hipStream_t stream, const float* x, const float* w, float* y, int M, int K, int N, bool wt, float alpha) {
    const int mb = dense_small_rows(M);
    if (mb == 8) launch_dense<8>(stream, x, w, y, M, K, N, wt, alpha);
    else if (mb == 16) launch_dense<16>(stream, x, w, y, M, K, N, wt, alpha);
    else launch_dense<32>(stream, x, w, y, M, K, N, wt, alpha);
}

bool dense_small_wgrad_ok(int M, int N, const void* dy, const void* dw) {
    return M >= 1 && M <= 32 && (N % 4) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dw)) & 15) == 0;
}

void dense_small_wgrad(hipStream_t stream, const float* x, const float* dy, float* dw, int M, int K, int N, float alpha) {
    const int total = K * (N >> 2);
    hipLaunchKernelGGL(dense_small_wgrad_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, x, dy, dw, M, K, N, alpha);
}

}  // namespace igan
