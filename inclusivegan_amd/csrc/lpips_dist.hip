// LPIPS per-layer distance on gfx950: channel unit-normalisation, squared difference, non-negative
// 1x1 "lin" weighting and spatial sum, fused into one streaming pass (and one for the gradient).
//
// Behavioural contract: what `lpips.get_output_for(a, b)` computes per feature layer
// (training/loss.py:31,41; Zhang et al. 2018 -- the reference's own network is the absent pickle
// metrics/vgg16_zhang_perceptual.pkl, so this is a restatement, parity unpinned):
//     u = fa / (||fa||_c + 1e-10),  v = fb / (||fb||_c + 1e-10)        (per pixel, over channels)
//     d[n] = sum_{h,w} sum_c lin_c (u_c - v_c)^2                          (caller divides by H*W)
// Gradient w.r.t. fa (fb is obtained by swapping the arguments; d is symmetric):
//     q_c = 2 lin_c (u_c - v_c);  dfa_k = g[n] * (q_k - u_k * sum_c q_c u_c) / (||fa|| + eps)
// (d u_c / d fa_k = delta_ck / (r+eps) - fa_c fa_k / (r (r+eps)^2);  r ~ r + eps to 1e-10.)
// MI355X design: features are channel-minor [N, HW, C]; a group of C/4 lanes owns one pixel (a wave
// covers 64 / (C/4) pixels for C <= 256, or one pixel with two float4 per lane for C = 512), channel
// sums are XOR-butterfly shuffles inside the group, so both tensors are read exactly once with 16 B
// per lane and nothing but the per-block partial sums (forward) or dfa (backward) is written.
#include "igan_common.h"

namespace {

struct LpArgs {
    const float* fa;
    const float* fb;
    const float* lin;     // [C], already non-negative
    const float* g;       // backward: [N] upstream gradient (already divided by HW by the caller)
    float* out;           // forward: partial [N][stride] (columns 0..blocks-1 written); backward: dfa [*,HW,C]
    int HW, C, blocks;
    const int* ia;        // pair tables (nullable = identity): row n compares sample ia[n] of fa with sample ib[n] of fb;
    const int* ib;        // the gradient of row n goes to sample ia[n] of dfa
    int stride;           // forward: floats between consecutive rows of partial
    int accumulate;       // backward: dfa += instead of dfa =
};

template <int V>   // float4 per lane per pixel: 1 (C <= 256) or 2 (C == 512)
__device__ __forceinline__ void load_px(const float* base, size_t pix, int C, int lane_c, float4 (&x)[V]) {
    const float4* p = reinterpret_cast<const float4*>(base + pix * C);
#pragma unroll
    for (int v = 0; v < V; v++) x[v] = p[lane_c + v * 64];
}

__device__ __forceinline__ float group_sum(float s, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

template <int V, bool BWD>
__global__ __launch_bounds__(256) void lpips_kernel(LpArgs a) {
    __shared__ float red[4];
    const int cv = a.C >> 2;                         // float4 per pixel
    const int gw = (V == 2) ? 64 : cv;               // lanes per pixel group
    const int ppw = 64 / gw;                         // pixels per wave per step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lane_c = lane % gw;                    // float4 column (first of V)
    const int sub = lane / gw;                       // pixel within the wave step
    const int n = blockIdx.y;
    const int na = a.ia ? a.ia[n] : n, nb = a.ib ? a.ib[n] : n;
    const int per_block = (a.HW + a.blocks - 1) / a.blocks;
    const int p0 = blockIdx.x * per_block;
    const int p1 = min(p0 + per_block, a.HW);
    float4 lw[V];
#pragma unroll
    for (int v = 0; v < V; v++) lw[v] = *reinterpret_cast<const float4*>(a.lin + 4 * (lane_c + v * 64));
    const float gn = BWD ? a.g[n] : 0.f;
    float acc = 0.f;
    for (int pb = p0 + wave * ppw; pb < p1; pb += 4 * ppw) {   // wave-uniform trip count
        const int p = pb + sub;
        const bool ok = p < p1;                      // tail lanes still take part in the shuffles
        const size_t pix = (size_t)na * a.HW + (ok ? p : p0);
        float4 xa[V], xb[V];
        load_px<V>(a.fa, pix, a.C, lane_c, xa);
        load_px<V>(a.fb, (size_t)nb * a.HW + (ok ? p : p0), a.C, lane_c, xb);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int v = 0; v < V; v++) {
            sa += xa[v].x * xa[v].x + xa[v].y * xa[v].y + xa[v].z * xa[v].z + xa[v].w * xa[v].w;
            sb += xb[v].x * xb[v].x + xb[v].y * xb[v].y + xb[v].z * xb[v].z + xb[v].w * xb[v].w;
        }
        sa = group_sum(sa, gw);
        sb = group_sum(sb, gw);
        const float ia = 1.0f / (sqrtf(sa) + 1e-10f), ib = 1.0f / (sqrtf(sb) + 1e-10f);
        if (!BWD) {
            float d = 0.f;
#pragma unroll
            for (int v = 0; v < V; v++) {
                const float e0 = xa[v].x * ia - xb[v].x * ib, e1 = xa[v].y * ia - xb[v].y * ib;
                const float e2 = xa[v].z * ia - xb[v].z * ib, e3 = xa[v].w * ia - xb[v].w * ib;
                d += lw[v].x * e0 * e0 + lw[v].y * e1 * e1 + lw[v].z * e2 * e2 + lw[v].w * e3 * e3;
            }
            if (ok) acc += d;
        } else {
            float4 q[V], u[V];
            float dot = 0.f;
#pragma unroll
            for (int v = 0; v < V; v++) {
                u[v] = make_float4(xa[v].x * ia, xa[v].y * ia, xa[v].z * ia, xa[v].w * ia);
                q[v].x = 2.f * lw[v].x * (u[v].x - xb[v].x * ib);
                q[v].y = 2.f * lw[v].y * (u[v].y - xb[v].y * ib);
                q[v].z = 2.f * lw[v].z * (u[v].z - xb[v].z * ib);
                q[v].w = 2.f * lw[v].w * (u[v].w - xb[v].w * ib);
                dot += q[v].x * u[v].x + q[v].y * u[v].y + q[v].z * u[v].z + q[v].w * u[v].w;
            }
            dot = group_sum(dot, gw);
            if (ok) {
                const float sc = gn * ia;
                float4* o = reinterpret_cast<float4*>(a.out + pix * a.C);
#pragma unroll
                for (int v = 0; v < V; v++) {
                    float4 r = make_float4(sc * (q[v].x - u[v].x * dot), sc * (q[v].y - u[v].y * dot),
                                           sc * (q[v].z - u[v].z * dot), sc * (q[v].w - u[v].w * dot));
                    if (a.accumulate) {
                        const float4 old = o[lane_c + v * 64];
                        r = make_float4(old.x + r.x, old.y + r.y, old.z + r.z, old.w + r.w);
                    }
                    o[lane_c + v * 64] = r;
                }
            }
        }
    }
    if (!BWD) {
        // block sum of acc (fixed order): full-wave butterfly, then the 4 waves through LDS
        acc = group_sum(acc, 64);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) a.out[(size_t)n * a.stride + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

int lp_blocks(int N, int HW) {
    const int want = std::max(1, 1024 / std::max(N, 1));
    return std::max(1, std::min(want, HW / 64 > 0 ? HW / 64 : 1));
}

int lp_check(int N, int HW, int C) {
    IGAN_REQUIRE(N >= 1 && N <= 65535 && HW >= 1, "lpips_layer: bad batch / spatial size");
    IGAN_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, "lpips_layer: C must be 64, 128, 256 or 512 (VGG16 feature widths)");
    IGAN_REQUIRE((long long)N * HW * C <= INT32_MAX, "lpips_layer: tensor too large");
    return IGAN_OK;
}

}  // namespace

extern "C" int igan_lpips_layer_blocks(int N, int HW) { return (N >= 1 && HW >= 1) ? lp_blocks(N, HW) : 0; }

extern "C" int igan_lpips_layer_fwd(igan_stream_t stream_, const float* fa, const float* fb, const float* lin,
                                    float* partial, int N, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(fa && fb && lin && partial, "lpips_layer_fwd: null buffer");
    if (int rc = lp_check(N, HW, C)) return rc;
    IGAN_REQUIRE((((uintptr_t)fa | (uintptr_t)fb | (uintptr_t)lin) & 15) == 0, "lpips_layer_fwd: buffers must be 16-byte aligned");
    const int blocks = lp_blocks(N, HW);
    LpArgs a{fa, fb, lin, nullptr, partial, HW, C, blocks, nullptr, nullptr, blocks, 0};
    dim3 grid(a.blocks, N);
    if (C == 512) hipLaunchKernelGGL((lpips_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((lpips_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    IGAN_LAUNCH_CHECK("lpips_layer_fwd launch");
    return IGAN_OK;
}

extern "C" int igan_lpips_pairs_fwd(igan_stream_t stream_, const float* fa, const float* fb, const float* lin, const int* ia,
                                    const int* ib, float* partial, int partial_stride, int blocks, int P, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(fa && fb && lin && partial, "lpips_pairs_fwd: null buffer");
    if (int rc = lp_check(P, HW, C)) return rc;
    IGAN_REQUIRE(blocks >= 1 && blocks <= HW && partial_stride >= blocks, "lpips_pairs_fwd: need 1 <= blocks <= HW and partial_stride >= blocks");
    IGAN_REQUIRE((((uintptr_t)fa | (uintptr_t)fb | (uintptr_t)lin) & 15) == 0, "lpips_pairs_fwd: buffers must be 16-byte aligned");
    LpArgs a{fa, fb, lin, nullptr, partial, HW, C, blocks, ia, ib, partial_stride, 0};
    dim3 grid(blocks, P);
    if (C == 512) hipLaunchKernelGGL((lpips_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((lpips_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    IGAN_LAUNCH_CHECK("lpips_pairs_fwd launch");
    return IGAN_OK;
}

extern "C" int igan_lpips_layer_bwd(igan_stream_t stream_, const float* fa, const float* fb, const float* lin,
                                    const float* g, float* dfa, int N, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(fa && fb && lin && g && dfa, "lpips_layer_bwd: null buffer");
    if (int rc = lp_check(N, HW, C)) return rc;
    IGAN_REQUIRE((((uintptr_t)fa | (uintptr_t)fb | (uintptr_t)lin | (uintptr_t)dfa) & 15) == 0, "lpips_layer_bwd: buffers must be 16-byte aligned");
    const int blocks = lp_blocks(N, HW);
    LpArgs a{fa, fb, lin, g, dfa, HW, C, blocks, nullptr, nullptr, blocks, 0};
    dim3 grid(a.blocks, N);
    if (C == 512) hipLaunchKernelGGL((lpips_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((lpips_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    IGAN_LAUNCH_CHECK("lpips_layer_bwd launch");
    return IGAN_OK;
}

extern "C" int igan_lpips_pairs_bwd(igan_stream_t stream_, const float* fa, const float* fb, const float* lin, const int* ia,
                                    const int* ib, const float* g, float* dfa, int accumulate, int P, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(fa && fb && lin && g && dfa, "lpips_pairs_bwd: null buffer");
    if (int rc = lp_check(P, HW, C)) return rc;
    IGAN_REQUIRE((((uintptr_t)fa | (uintptr_t)fb | (uintptr_t)lin | (uintptr_t)dfa) & 15) == 0, "lpips_pairs_bwd: buffers must be 16-byte aligned");
    const int blocks = lp_blocks(P, HW);
    LpArgs a{fa, fb, lin, g, dfa, HW, C, blocks, ia, ib, blocks, accumulate ? 1 : 0};
    dim3 grid(blocks, P);
    if (C == 512) hipLaunchKernelGGL((lpips_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((lpips_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    IGAN_LAUNCH_CHECK("lpips_pairs_bwd launch");
    return IGAN_OK;
}
