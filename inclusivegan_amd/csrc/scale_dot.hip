// Per-sample channel dot products for the modulated-conv backward on gfx950.
//
// Behavioural contract: the gradients of modulated_conv2d_layer w.r.t. the style and the
// demodulation coefficients in its non-fused form (training/networks_stylegan2.py:112,126):
//     y = d[n,co] * conv(x * s[n,ci], w)
//     ds[n,ci] = sum_{h,w} x[n,h,w,ci] * g[n,h,w,ci],      g = dgrad(dy * d, w)      (and dx = g * s)
//     dd[n,co] = sum_{h,w} dy[n,h,w,co] * conv(x*s, w)[n,h,w,co] = sum_{h,w} dy * y / d
// which the reference obtains from tf.gradients of the broadcast multiplies (reduce_sum over H, W).
// MI355X design: one streaming pass over the two channel-minor tensors:
//     dot[n,c] = sum_rows a[n,row,c] * b[n,row,c];   optionally  out[n,row,c] = b[n,row,c] * s[n,c]
// (out may alias b).  Same thread layout as the fused epilogue backward (row lanes x float4 columns,
// column sums in registers, LDS fold, per-block partials, fixed-order final add: deterministic).
#include "igan_common.h"

namespace {

struct SdArgs {
    const float* a;
    const float* b;
    const float* s;   // [N, C] or NULL
    float* out;       // b * s, or NULL
    float* partial;   // [N][blocks][C]
    int HW, C, blocks;
};

__global__ __launch_bounds__(256) void scale_dot_kernel(SdArgs p) {
    __shared__ float4 red[256];
    const int cv = p.C >> 2;
    const int cvt = min(cv, 256);
    const int rl = 256 / cvt;
    const int col = threadIdx.x % cvt + blockIdx.z * 256;
    const int lane_r = threadIdx.x / cvt;
    const int n = blockIdx.y;
    const bool active = (col < cv) && (lane_r < rl);
    const int rows_per_block = (p.HW + p.blocks - 1) / p.blocks;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, p.HW);
    const size_t base = (size_t)n * p.HW * cv;
    const float4* a4 = reinterpret_cast<const float4*>(p.a) + base;
    const float4* b4 = reinterpret_cast<const float4*>(p.b) + base;
    float4* o4 = p.out ? reinterpret_cast<float4*>(p.out) + base : nullptr;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (active && p.s) sc = *reinterpret_cast<const float4*>(p.s + (size_t)n * p.C + (size_t)col * 4);
    if (active) {
        // 4 rows per iteration with their 8 loads in flight together (a single row per iteration leaves too few bytes in
        // flight per CU to reach HBM speed); the sum keeps its row order.
        for (int rb = r0 + lane_r; rb < r1; rb += 4 * rl) {
            float4 x[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const size_t i = (size_t)min(rb + u * rl, r1 - 1) * cv + col;     // clamped, masked below
                x[u] = a4[i];
                g[u] = b4[i];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = rb + u * rl;
                if (r < r1) {
                    acc.x += x[u].x * g[u].x; acc.y += x[u].y * g[u].y; acc.z += x[u].z * g[u].z; acc.w += x[u].w * g[u].w;
                    if (o4) o4[(size_t)r * cv + col] = make_float4(g[u].x * sc.x, g[u].y * sc.y, g[u].z * sc.z, g[u].w * sc.w);
                }
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < cvt && col < cv) {
        float4 t = red[threadIdx.x];
        for (int j = 1; j < rl; j++) {
            const float4 u = red[threadIdx.x + j * cvt];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float* q = p.partial + ((size_t)n * p.blocks + blockIdx.x) * p.C + (size_t)col * 4;
        q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
    }
}

// dot[n][c] = sum_j partial[n][j][c]; grid = (ceil(C/16), N), 16 columns x 16 groups.
__global__ __launch_bounds__(256) void scale_dot_final_kernel(const float* partial, float* dot, int blocks, int C) {
    __shared__ float red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const int grp = threadIdx.x >> 4;
    const int n = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        // four independent partial sums: four loads in flight per lane instead of a chain of dependent L2 round trips
        const float* q = partial + (size_t)n * blocks * C + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int j = grp;
        for (; j + 48 < blocks; j += 64) {
            s0 += q[(size_t)j * C];        s1 += q[(size_t)(j + 16) * C];
            s2 += q[(size_t)(j + 32) * C]; s3 += q[(size_t)(j + 48) * C];
        }
        for (; j < blocks; j += 16) s0 += q[(size_t)j * C];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) t += red[g * 16 + threadIdx.x];
        dot[(size_t)n * C + c] = t;
    }
}

int sd_blocks(int N, int HW, int C) {
    const int cvt = std::min(C / 4, 256);
    const int rl = 256 / cvt;
    int b = HW / (rl * 8);                       // >= 8 rows per row lane
    const int want = std::max(1, 1024 / std::max(N, 1));   // ~1024 blocks over the batch
    return std::max(1, std::min(b, want));
}

}  // namespace

extern "C" size_t igan_scale_dot_workspace_floats(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C < 4) return 0;
    return (size_t)N * sd_blocks(N, HW, C) * C;
}

extern "C" int igan_scale_dot(igan_stream_t stream_, const float* a, const float* b, const float* s, float* out,
                              float* dot, float* workspace, int N, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(a && b && dot && workspace, "scale_dot: null buffer");
    IGAN_REQUIRE(N >= 1 && HW >= 1 && C >= 4 && (C % 4) == 0, "scale_dot: needs channel-minor data with C %% 4 == 0");
    IGAN_REQUIRE((long long)N * HW * C <= INT32_MAX, "scale_dot: tensor too large");
    IGAN_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)s) & 15) == 0, "scale_dot: buffers must be 16-byte aligned");
    IGAN_REQUIRE(N <= 65535, "scale_dot: batch too large");
    const int blocks = sd_blocks(N, HW, C);
    SdArgs p{a, b, s, out, workspace, HW, C, blocks};
    dim3 grid(blocks, N, ceil_div(C / 4, 256));
    hipLaunchKernelGGL(scale_dot_kernel, grid, dim3(256), 0, (hipStream_t)stream_, p);
    hipLaunchKernelGGL(scale_dot_final_kernel, dim3(ceil_div(C, 16), N), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)workspace, dot, blocks, C);
    IGAN_LAUNCH_CHECK("scale_dot launch");
    return IGAN_OK;
}
