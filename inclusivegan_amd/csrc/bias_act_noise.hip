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

// Backward of the epilogue when it was fused into a modulated convolution, y = act(d[n,c] * z + noise * strength + b) * gain
// (z = the un-demodulated convolution output, never stored): besides dx (the gradient w.r.t. d * z), db and dstrength it returns
//     dd[n][c] = sum_{pixels of n} dx * z = (sum_pixels dx * (pre - b[c] - noise * strength)) / d[n][c],
// the demodulation gradient, from the SAME pass -- pre, the pre-activation value, is recovered from y (linear / lrelu are
// invertible: pre = y / gain for y >= 0, y / (gain * alpha) below).  Replaces ban_bwd + a separate scale-dot over (dx, d * z).
// Blocks never straddle samples: block = (sample, slice of its pixels); partial[block] = [C db | 1 dstrength | C dd].
__global__ __launch_bounds__(256) void ban_bwd_dd_kernel(BanArgs a, const float* __restrict__ strength_p, int HW, int bps) {
    __shared__ float4 red[256];
    __shared__ float4 red2[256];
    __shared__ float reds[256];
    const int cv = a.C >> 2;
    const int cvt = min(cv, 256);
    const int rl = 256 / cvt;
    const int col = threadIdx.x % cvt + blockIdx.y * 256;
    const int lane_r = threadIdx.x / cvt;
    const bool active = (col < cv) && (lane_r < rl);
    const int n = blockIdx.x / bps, j = blockIdx.x - n * bps;
    const int per = (HW + bps - 1) / bps;
    const int r0 = n * HW + j * per;
    const int r1 = min(r0 + per, (n + 1) * HW);
    const float4* dy4 = reinterpret_cast<const float4*>(a.x);
    const float4* y4 = reinterpret_cast<const float4*>(a.ref);
    float4* dx4 = reinterpret_cast<float4*>(a.y);
    const float st = (a.noise && strength_p) ? strength_p[0] : 0.f;
    const float inv_gain = 1.0f / a.gain;
    const float inv_neg = (a.act == 3) ? 1.0f / (a.gain * a.alpha) : inv_gain;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active && a.b) bb = *reinterpret_cast<const float4*>(a.b + 4 * col);
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f), accd = make_float4(0.f, 0.f, 0.f, 0.f);
    float accs = 0.f;
    if (active) {
        for (int rb = r0 + lane_r; rb < r1; rb += 4 * rl) {
            float4 g[4], yy[4];
            float nz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = min(rb + u * rl, r1 - 1);
                const long long i = (long long)r * cv + col;
                g[u] = dy4[i];
                yy[u] = y4[i];
                nz[u] = a.noise ? a.noise[a.rows < 0 ? r - n * HW : r] : 0.f;
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
                    const float off = nz[u] * st;
                    accd.x += d.x * (yy[u].x * (yy[u].x >= 0.f ? inv_gain : inv_neg) - bb.x - off);
                    accd.y += d.y * (yy[u].y * (yy[u].y >= 0.f ? inv_gain : inv_neg) - bb.y - off);
                    accd.z += d.z * (yy[u].z * (yy[u].z >= 0.f ? inv_gain : inv_neg) - bb.z - off);
                    accd.w += d.w * (yy[u].w * (yy[u].w >= 0.f ? inv_gain : inv_neg) - bb.w - off);
                }
            }
        }
    }
    red[threadIdx.x] = accb;
    red2[threadIdx.x] = accd;
    reds[threadIdx.x] = accs;
    __syncthreads();
    const int W = 2 * a.C + 1;
    if (threadIdx.x < cvt && col < cv) {
        float4 t = red[threadIdx.x], t2 = red2[threadIdx.x];
        for (int k = 1; k < rl; k++) {
            const float4 u = red[threadIdx.x + k * cvt], u2 = red2[threadIdx.x + k * cvt];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            t2.x += u2.x; t2.y += u2.y; t2.z += u2.z; t2.w += u2.w;
        }
        float* p = a.partial + (size_t)blockIdx.x * W + (size_t)col * 4;
        p[0] = t.x; p[1] = t.y; p[2] = t.z; p[3] = t.w;
        float* q = p + a.C + 1;
        q[0] = t2.x; q[1] = t2.y; q[2] = t2.z; q[3] = t2.w;
    }
    if (blockIdx.y == 0) {
        float s = a.noise ? reds[threadIdx.x] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * W + a.C] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    }
}

// db[c] = sum_blocks partial[.][c]; dstrength = sum_blocks partial[.][C]; dd[n][c] = (sum_{blocks of n} partial[.][C + 1 + c]) / d[n][c].
// grid = (ceil((C + 1) / 16), 1 + N): y == 0 does db / dstrength over all blocks, y == 1 + n does dd of sample n.  Fixed order.
__global__ __launch_bounds__(256) void ban_dd_final_kernel(const float* partial, const float* dscale, float* db, float* dstrength, float* dd,
                                                           int N, int bps, int C) {
    __shared__ float red[256];
    const int W = 2 * C + 1;
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const int grp = threadIdx.x >> 4;
    const bool sample = blockIdx.y > 0;
    const int n = (int)blockIdx.y - 1;
    const int b0 = sample ? n * bps : 0, b1 = sample ? (n + 1) * bps : N * bps;
    const int off = sample ? C + 1 : 0;
    const int cmax = sample ? C - 1 : C;
    float s = 0.f;
    if (c <= cmax) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int j = b0 + grp;
        for (; j + 48 < b1; j += 64) {
            s0 += partial[(size_t)j * W + off + c];        s1 += partial[(size_t)(j + 16) * W + off + c];
            s2 += partial[(size_t)(j + 32) * W + off + c]; s3 += partial[(size_t)(j + 48) * W + off + c];
        }
        for (; j < b1; j += 16) s0 += partial[(size_t)j * W + off + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && c <= cmax) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) t += red[g * 16 + threadIdx.x];
        if (sample) { if (dd) dd[(size_t)n * C + c] = t / dscale[(size_t)n * C + c]; }
        else if (c < C) { if (db) db[c] = t; }
        else if (dstrength) *dstrength = t;
    }
}

int ban_dd_bps(int N, int HW, int C) {
    const int cvt = std::min(C / 4, 256);
    const int rl = 256 / cvt;
    // aim for >= 8 rows per row-lane per block and about 512 blocks in all
    return std::max(1, std::min(std::max(512 / std::max(N, 1), 1), std::max(HW / (rl * 8), 1)));
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

extern "C" size_t igan_bias_act_noise_dd_workspace_floats(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C < 4) return 0;
    return (size_t)N * ban_dd_bps(N, HW, C) * (size_t)(2 * C + 1);
}

extern "C" int igan_bias_act_noise_bwd_dd(igan_stream_t stream_, const float* dy, const float* y, const float* noise, const float* strength,
                                          const float* b, const float* dscale, float* dx, float* db, float* dstrength, float* dd,
                                          float* workspace, int noise_bcast, int N, int HW, int C, int act, float alpha, float gain) {
    using namespace igan;
    IGAN_REQUIRE(dy && y && dx && workspace && dscale && dd, "bias_act_noise_bwd_dd: null buffer");
    IGAN_REQUIRE((noise == nullptr) || (strength != nullptr && dstrength != nullptr), "bias_act_noise_bwd_dd: noise given without strength / dstrength");
    IGAN_REQUIRE(N >= 1 && HW >= 1 && (long long)N * HW <= INT32_MAX, "bias_act_noise_bwd_dd: bad sizes");
    if (int rc = ban_check("bias_act_noise_bwd_dd", N * HW, C, act, gain)) return rc;
    IGAN_REQUIRE(noise == nullptr || C <= 1024, "bias_act_noise_bwd_dd: with noise C must be <= 1024 (the noise-strength partial is reduced by the first block column only)");
    IGAN_REQUIRE(act == 1 || act == 3, "bias_act_noise_bwd_dd: the pre-activation value is recovered from y: linear or lrelu only");
    IGAN_REQUIRE(act != 3 || alpha > 0.0f, "bias_act_noise_bwd_dd: lrelu slope must be positive");
    IGAN_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)b) & 15) == 0, "bias_act_noise_bwd_dd: buffers must be 16-byte aligned");
    // rows < 0 in the kernel's argument block means: the noise has one sample, shared by the batch
    BanArgs a{dy, y, noise, nullptr, b, dx, workspace, noise_bcast ? -1 : N * HW, C, act, alpha, gain};
    const int bps = ban_dd_bps(N, HW, C);
    dim3 grid(N * bps, ceil_div(C / 4, 256));
    hipLaunchKernelGGL(ban_bwd_dd_kernel, grid, dim3(256), 0, (hipStream_t)stream_, a, strength, HW, bps);
    hipLaunchKernelGGL(ban_dd_final_kernel, dim3(ceil_div(C + 1, 16), 1 + N), dim3(256), 0, (hipStream_t)stream_,
                       (const float*)workspace, dscale, db, noise ? dstrength : nullptr, dd, N, bps, C);
    IGAN_LAUNCH_CHECK("bias_act_noise_bwd_dd launch");
    return IGAN_OK;
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
