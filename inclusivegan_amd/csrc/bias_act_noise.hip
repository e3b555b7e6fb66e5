// Fused synthesis-layer epilogue for gfx950 and its one-pass backward.
//
// Behavioural contract: the tail of `layer()` in training/networks_stylegan2.py:351-357
//     x += noise[n,1,h,w] * noise_strength;  y = fused_bias_act(x + bias, act) * gain
// (with noise == NULL it is plain `apply_bias_act`, networks_stylegan2.py:66-68), and the gradients
// the reference assembles from FusedBiasAct(grad=1) + two tf.reduce_sum passes
// (dnnlib/tflib/ops/fused_bias_act.py:132-146) + the broadcast-multiply gradients of the noise term.
// Only activations whose derivative kernel takes ref = y and has zero second derivative
// (linear / relu / lrelu: fused_bias_act.py:21-23) are offered here; the general op stays in
// fused_bias_act.hip.
//
// MI355X design: channel-minor activations [rows = N*H*W][C]; both kernels are single HBM streams.
//   forward : read x (16 B/lane), write y; noise is one float per row, bias one float4 per lane.
//   backward: read dy and y, write dx, and in the same pass accumulate
//                 db[c]     = sum_rows dx[row][c]
//                 dstrength = sum_rows noise[row] * sum_c dx[row][c]
//             Threads are laid out (row lane) x (float4 column) so a thread's column is fixed while it
//             walks rows: column sums stay in registers, row-lanes are folded through LDS once per
//             block, per-block partials go to a caller workspace and a second tiny kernel adds them
//             in fixed order (bit-reproducible; no float atomics).
#include "igan_common.h"

namespace {

struct BanArgs {
    const float* x;        // fwd: input        bwd: dy
    const float* ref;      // bwd: y
    const float* noise;    // [rows] or NULL
    const float* strength; // device scalar or NULL
    const float* b;        // [C] or NULL (fwd)
    float* y;              // fwd: output       bwd: dx
    float* partial;        // bwd: [blocks][C + 1]
    int rows, C;
    int act;               // 1 linear, 2 relu, 3 lrelu
    float alpha, gain;
};

__device__ __forceinline__ float act_fwd(int act, float v, float alpha) {
    if (act == 2) return v > 0.f ? v : 0.f;
    if (act == 3) return v > 0.f ? v : v * alpha;
    return v;
}
__device__ __forceinline__ float act_bwd(int act, float dy, float y, float alpha) {
    // fused_bias_act.cu:69,74,79 with ref = y / gain (same sign as y for gain > 0)
    if (act == 2) return y > 0.f ? dy : 0.f;
    if (act == 3) return y > 0.f ? dy : dy * alpha;
    return dy;
}

__global__ __launch_bounds__(256) void ban_fwd_kernel(BanArgs a) {
    const int cv = a.C >> 2;
    const long long n4 = (long long)a.rows * cv;
    const float s = (a.noise && a.strength) ? *a.strength : 0.f;
    const float4* x4 = reinterpret_cast<const float4*>(a.x);
    float4* y4 = reinterpret_cast<float4*>(a.y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / cv);
        const int c = (int)(i - (long long)row * cv) << 2;
        float4 v = x4[i];
        float add = a.noise ? a.noise[row] * s : 0.f;
        float4 bb = a.b ? *reinterpret_cast<const float4*>(a.b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = act_fwd(a.act, v.x + add + bb.x, a.alpha) * a.gain;
        v.y = act_fwd(a.act, v.y + add + bb.y, a.alpha) * a.gain;
        v.z = act_fwd(a.act, v.z + add + bb.z, a.alpha) * a.gain;
        v.w = act_fwd(a.act, v.w + add + bb.w, a.alpha) * a.gain;
        y4[i] = v;
    }
}

// blockDim = 256 = RL row lanes x CV float4 columns (CV = min(C/4, 256) per column tile; grid.y tiles C).
__global__ __launch_bounds__(256) void ban_bwd_kernel(BanArgs a) {
    __shared__ float4 red[256];
    __shared__ float reds[256];
    const int cv = a.C >> 2;
    const int cvt = min(cv, 256);               // columns handled by this block
    const int rl = 256 / cvt;                   // row lanes
    const int col = threadIdx.x % cvt + blockIdx.y * 256;   // float4 column
    const int lane_r = threadIdx.x / cvt;
    const bool active = (col < cv) && (lane_r < rl);
    const int rows_per_block = (a.rows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, a.rows);
    const float4* dy4 = reinterpret_cast<const float4*>(a.x);
    const float4* y4 = reinterpret_cast<const float4*>(a.ref);
    float4* dx4 = reinterpret_cast<float4*>(a.y);
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
    float accs = 0.f;
    if (active) {
        // 4 rows per iteration, their 8 loads in flight together: with one row per iteration a CU holds ~16 KB of
        // reads in flight, short of what 6 TB/s needs (4.0 TB/s measured); the sums keep their row order.
        for (int rb = r0 + lane_r; rb < r1; rb += 4 * rl) {
            float4 g[4], yy[4];
            float nz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = min(rb + u * rl, r1 - 1);        // clamped: always a valid row, masked below
                const long long i = (long long)r * cv + col;
                g[u] = dy4[i];
                yy[u] = y4[i];
                nz[u] = a.noise ? a.noise[r] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = rb + u * rl;
                if (r < r1) {
                    float4 d;
                    d.x = act_bwd(a.act, g[u].x, yy[u].x, a.alpha) * a.gain;
                    d.y = act_bwd(a.act, g[u].y, yy[u].y, a.alpha) * a.gain;
                    d.z = act_bwd(a.act, g[u].z, yy[u].z, a.alpha) * a.gain;
                    d.w = act_bwd(a.act, g[u].w, yy[u].w, a.alpha) * a.gain;
                    dx4[(long long)r * cv + col] = d;
                    accb.x += d.x; accb.y += d.y; accb.z += d.z; accb.w += d.w;
                    accs += nz[u] * ((d.x + d.y) + (d.z + d.w));
                }
            }
        }
    }
    red[threadIdx.x] = accb;
    reds[threadIdx.x] = accs;
    __syncthreads();
    // fold row lanes (fixed order)
    if (threadIdx.x < cvt && col < cv) {
        float4 t = red[threadIdx.x];
        for (int j = 1; j < rl; j++) {
            const float4 u = red[threadIdx.x + j * cvt];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float* p = a.partial + (size_t)blockIdx.x * (a.C + 1) + (size_t)col * 4;
        p[0] = t.x; p[1] = t.y; p[2] = t.z; p[3] = t.w;
    }
    if (a.noise && blockIdx.y == 0) {
        // block sum of accs in fixed order: wave shuffle tree then 4 waves
        float s = reds[threadIdx.x];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * (a.C + 1) + a.C] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    } else if (!a.noise && blockIdx.y == 0 && threadIdx.x == 0) {
        a.partial[(size_t)blockIdx.x * (a.C + 1) + a.C] = 0.f;
    }
}

// db[c] = sum_j partial[j][c]; dstrength = sum_j partial[j][C].  One block per 16 columns: 16 columns x
// 16 groups of partial rows (each thread adds <= blocks/16 values, 64 B coalesced per 16 lanes), folded
// through LDS in fixed order.
__global__ __launch_bounds__(256) void ban_final_kernel(const float* partial, float* db, float* dstrength, int blocks, int C) {
    __shared__ float red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const int grp = threadIdx.x >> 4;
    float s = 0.f;
    if (c <= C) {
        // four independent partial sums: four loads in flight per lane instead of a chain of dependent L2 round trips
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int j = grp;
        for (; j + 48 < blocks; j += 64) {
            s0 += partial[(size_t)j * (C + 1) + c];        s1 += partial[(size_t)(j + 16) * (C + 1) + c];
            s2 += partial[(size_t)(j + 32) * (C + 1) + c]; s3 += partial[(size_t)(j + 48) * (C + 1) + c];
        }
        for (; j < blocks; j += 16) s0 += partial[(size_t)j * (C + 1) + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && c <= C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) t += red[g * 16 + threadIdx.x];
        if (c < C) { if (db) db[c] = t; }
        else if (dstrength) *dstrength = t;
    }
}

int ban_blocks(int rows, int C) {
    const int cvt = std::min(C / 4, 256);
    const int rl = 256 / cvt;
    // aim for >= 8 rows per row-lane per block, at most 512 blocks
    int b = rows / (rl * 8);
    return std::max(1, std::min(b, 512));
}

int ban_check(const char* who, int rows, int C, int act, float gain) {
    IGAN_REQUIRE(gain > 0.0f, "%s: gain must be positive (the derivative is keyed on the sign of y)", who);
    IGAN_REQUIRE(rows >= 1 && C >= 4 && (C % 4) == 0, "%s: needs channel-minor data with C %% 4 == 0", who);
    IGAN_REQUIRE((long long)rows * C <= INT32_MAX, "%s: x is too large", who);
    IGAN_REQUIRE(act >= 1 && act <= 3, "%s: only linear / relu / lrelu (act 1..3) are offered here", who);
    return IGAN_OK;
}

}  // namespace

extern "C" size_t igan_bias_act_noise_workspace_floats(int rows, int C) {
    if (rows <= 0 || C < 4) return 0;
    return (size_t)ban_blocks(rows, C) * (size_t)(C + 1);
}

extern "C" int igan_bias_act_noise_fwd(igan_stream_t stream_, const float* x, const float* noise, const float* strength,
                                       const float* b, float* y, int rows, int C, int act, float alpha, float gain) {
    using namespace igan;
    IGAN_REQUIRE(x && y, "bias_act_noise_fwd: null buffer");
    IGAN_REQUIRE((noise == nullptr) == (strength == nullptr), "bias_act_noise_fwd: noise and strength go together");
    if (int rc = ban_check("bias_act_noise_fwd", rows, C, act, gain)) return rc;
    IGAN_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)b) & 15) == 0, "bias_act_noise_fwd: buffers must be 16-byte aligned");
    BanArgs a{x, nullptr, noise, strength, b, y, nullptr, rows, C, act, alpha, gain};
    const long long n4 = (long long)rows * (C / 4);
    const int grid = (int)std::min<long long>(ceil_div_ll(n4, 256), 256 * 16);
    hipLaunchKernelGGL(ban_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, a);
    IGAN_LAUNCH_CHECK("bias_act_noise_fwd launch");
    return IGAN_OK;
}

extern "C" int igan_bias_act_noise_bwd(igan_stream_t stream_, const float* dy, const float* y, const float* noise,
                                       float* dx, float* db, float* dstrength, float* workspace,
                                       int rows, int C, int act, float alpha, float gain) {
    using namespace igan;
    IGAN_REQUIRE(dy && y && dx && workspace, "bias_act_noise_bwd: null buffer");
    IGAN_REQUIRE((noise == nullptr) || (dstrength != nullptr), "bias_act_noise_bwd: noise given without dstrength");
    if (int rc = ban_check("bias_act_noise_bwd", rows, C, act, gain)) return rc;
    IGAN_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15) == 0, "bias_act_noise_bwd: buffers must be 16-byte aligned");
    BanArgs a{dy, y, noise, nullptr, nullptr, dx, workspace, rows, C, act, alpha, gain};
    const int blocks = ban_blocks(rows, C);
    dim3 grid(blocks, ceil_div(C / 4, 256));
    hipLaunchKernelGGL(ban_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream_, a);
    hipLaunchKernelGGL(ban_final_kernel, dim3(ceil_div(C + 1, 16)), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)workspace, db, noise ? dstrength : nullptr, blocks, C);
    IGAN_LAUNCH_CHECK("bias_act_noise_bwd launch");
    return IGAN_OK;
}
