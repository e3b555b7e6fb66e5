// Flat-bucket optimizer kernels for gfx950: finite check, Adam, EMA.
//
// Behavioural contract: dnnlib/tflib/optimizer.py:237-239 (skip the whole update
// when any gradient is non-finite) and :318-332 (SimpleAdam arithmetic, "behaves
// identically" to tf.train.AdamOptimizer); dnnlib/tflib/network.py:341-351 (EMA).
// Design: the reference issues one Adam kernel chain and one NCCL call per
// variable (96 for G); here all trainables of a network live in ONE contiguous
// fp32 bucket, so a step is one finite-check stream + one update stream over
// ~24.5 M floats (HBM-bound: 4 reads + 3 writes per element), with the skip
// decision and the beta-power bookkeeping kept on the device so that the host
// never synchronises (hipGraph-capturable).
#include "igan_common.h"

namespace {

__global__ __launch_bounds__(256) void finite_check_kernel(const float* g, int n, int* flag) {
    const int n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        // (v - v) is 0 for finite v and NaN for +-inf / NaN.
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        bad |= !((v - v) == 0.0f);
    }
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
}

struct AdamArgs {
    float* w;
    const float* g;
    float* m;
    float* v;
    int n;
    float lr, beta1, beta2, eps;
    const float* pow_state;
    const int* skip_flag;
};

__device__ __forceinline__ void adam_one(float& w, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    // optimizer.py:327-329
    const float m_new = b1 * m + (1.0f - b1) * g;
    const float v_new = b2 * v + (1.0f - b2) * (g * g);
    const float delta = lr_t * m_new / (sqrtf(v_new) + eps);
    m = m_new;
    v = v_new;
    w -= delta;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    if (a.skip_flag && *a.skip_flag) return;
    // optimizer.py:318-322
    const float b1pow = a.pow_state[0] * a.beta1;
    const float b2pow = a.pow_state[1] * a.beta2;
    const float lr_t = a.lr * sqrtf(1.0f - b2pow) / (1.0f - b1pow);
    const int n4 = a.n >> 2;
    float4* w4 = reinterpret_cast<float4*>(a.w);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 w = w4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_one(w.x, g.x, m.x, v.x, lr_t, a.beta1, a.beta2, a.eps);
        adam_one(w.y, g.y, m.y, v.y, lr_t, a.beta1, a.beta2, a.eps);
        adam_one(w.z, g.z, m.z, v.z, lr_t, a.beta1, a.beta2, a.eps);
        adam_one(w.w, g.w, m.w, v.w, lr_t, a.beta1, a.beta2, a.eps);
        w4[i] = w; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int i = (n4 << 2) + threadIdx.x;
        adam_one(a.w[i], a.g[i], a.m[i], a.v[i], lr_t, a.beta1, a.beta2, a.eps);
    }
}

__global__ void adam_advance_kernel(float* pow_state, float beta1, float beta2, const int* skip_flag) {
    if (skip_flag && *skip_flag) return;
    pow_state[0] *= beta1;
    pow_state[1] *= beta2;
}

__global__ __launch_bounds__(256) void ema_kernel(float* dst, const float* src, int n, float beta) {
    const int n4 = n >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 d = d4[i];
        const float4 s = s4[i];
        // tfutil.lerp(src, dst, beta) = src + (dst - src) * beta
        d.x = s.x + (d.x - s.x) * beta;
        d.y = s.y + (d.y - s.y) * beta;
        d.z = s.z + (d.z - s.z) * beta;
        d.w = s.w + (d.w - s.w) * beta;
        d4[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int i = (n4 << 2) + threadIdx.x;
        dst[i] = src[i] + (dst[i] - src[i]) * beta;
    }
}

inline int stream_grid(int n4) { return std::max(1, std::min(igan::ceil_div(n4, 256), 256 * 8)); }

// Running mean bookkeeping of a training scalar (dnnlib/tflib/autosummary.py:45-74: [count, sum] of the finite values, :64):
// acc[0] += number of finite x[i], acc[1] += their sum, in double, one workgroup, fixed-order tree (bit-reproducible).
__global__ __launch_bounds__(256) void summary_accumulate_kernel(const float* __restrict__ x, int n, double* __restrict__ acc) {
    __shared__ double s_cnt[256];
    __shared__ double s_sum[256];
    double cnt = 0.0, sum = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        if (isfinite(v)) { cnt += 1.0; sum += (double)v; }
    }
    s_cnt[threadIdx.x] = cnt;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { acc[0] += s_cnt[0]; acc[1] += s_sum[0]; }
}

}  // namespace

extern "C" int igan_summary_accumulate(igan_stream_t stream_, const float* x, int n, double* acc) {
    using namespace igan;
    IGAN_REQUIRE(x && acc, "summary_accumulate: null buffer");
    IGAN_REQUIRE(n >= 0, "summary_accumulate: negative size");
    if (n == 0) return IGAN_OK;
    hipLaunchKernelGGL(summary_accumulate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, x, n, acc);
    IGAN_LAUNCH_CHECK("summary_accumulate launch");
    return IGAN_OK;
}

extern "C" int igan_finite_check(igan_stream_t stream_, const float* g, int n, int* flag) {
    using namespace igan;
    IGAN_REQUIRE(g && flag, "finite_check: null buffer");
    IGAN_REQUIRE(n >= 0, "finite_check: negative size");
    IGAN_REQUIRE(((uintptr_t)g & 15) == 0, "finite_check: buffer must be 16-byte aligned");
    if (n == 0) return IGAN_OK;
    hipLaunchKernelGGL(finite_check_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream_, g, n, flag);
    IGAN_LAUNCH_CHECK("finite_check launch");
    return IGAN_OK;
}

extern "C" int igan_adam_step(igan_stream_t stream_, float* w, const float* g, float* m, float* v,
                              int n, float lr, float beta1, float beta2, float eps,
                              float* pow_state, const int* skip_flag) {
    using namespace igan;
    IGAN_REQUIRE(w && g && m && v && pow_state, "adam_step: null buffer");
    IGAN_REQUIRE(n >= 0, "adam_step: negative size");
    IGAN_REQUIRE((((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return IGAN_OK;
    AdamArgs a{w, g, m, v, n, lr, beta1, beta2, eps, pow_state, skip_flag};
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, pow_state, beta1, beta2, skip_flag);
    IGAN_LAUNCH_CHECK("adam_step launch");
    return IGAN_OK;
}

extern "C" int igan_ema(igan_stream_t stream_, float* dst, const float* src, int n, float beta) {
    using namespace igan;
    IGAN_REQUIRE(dst && src, "ema: null buffer");
    IGAN_REQUIRE(n >= 0, "ema: negative size");
    IGAN_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "ema: buffers must be 16-byte aligned");
    if (n == 0) return IGAN_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream_, dst, src, n, beta);
    IGAN_LAUNCH_CHECK("ema launch");
    return IGAN_OK;
}
