// fused_bias_act for gfx950: y = act(x + b) * gain and its 1st / 2nd derivative
// forms, plus the bias-gradient reduction.
//
// Behavioural contract: dnnlib/tflib/ops/fused_bias_act.cu:42-116 (per-element
// formulas selected by act*10+grad, `ref` = y/gain or x) and :139-171 (argument
// checks); bias gradient = fused_bias_act.py:137-146.
// Design: a pure HBM stream -- 16 B per lane per access, grid-stride over float4s;
// the act/grad switch is resolved at compile time (template) so the streaming loop
// has no branches.  With channel-minor activations stepB == 1, so a lane's float4
// maps to four consecutive bias entries and the bias load is one (L1/L2-resident)
// float4 as well.
#include "igan_common.h"
#include <hip/hip_fp16.h>

namespace {

struct FbaArgs {
    const float* x;
    const float* b;
    const float* ref;
    float* y;
    float alpha;
    float gain;
    int sizeX;
    int sizeB;
    int stepB;
};

template <int ACT, int GRAD>
__device__ __forceinline__ float fba_eval(float x, float ref, float alpha) {
    const float expRange = 80.0f;
    const float halfExpRange = 40.0f;
    const float seluScale = 1.0507009873554804934193349852946f;
    const float seluAlpha = 1.6732632423543772848170429916717f;
    constexpr int sel = ACT * 10 + GRAD;
    float y;
    if constexpr (sel == 10 || sel == 11) y = x;
    else if constexpr (sel == 12) y = 0.0f;
    else if constexpr (sel == 20) y = (x > 0.0f) ? x : 0.0f;
    else if constexpr (sel == 21) y = (ref > 0.0f) ? x : 0.0f;
    else if constexpr (sel == 22) y = 0.0f;
    else if constexpr (sel == 30) y = (x > 0.0f) ? x : x * alpha;
    else if constexpr (sel == 31) y = (ref > 0.0f) ? x : x * alpha;
    else if constexpr (sel == 32) y = 0.0f;
    else if constexpr (sel == 40) { float c = expf(x); float d = 1.0f / c; y = (x < -expRange) ? -1.0f : (x > expRange) ? 1.0f : (c - d) / (c + d); }
    else if constexpr (sel == 41) y = x * (1.0f - ref * ref);
    else if constexpr (sel == 42) y = x * (1.0f - ref * ref) * (-2.0f * ref);
    else if constexpr (sel == 50) y = (x < -expRange) ? 0.0f : 1.0f / (expf(-x) + 1.0f);
    else if constexpr (sel == 51) y = x * ref * (1.0f - ref);
    else if constexpr (sel == 52) y = x * ref * (1.0f - ref) * (1.0f - 2.0f * ref);
    else if constexpr (sel == 60) y = (x >= 0.0f) ? x : expf(x) - 1.0f;
    else if constexpr (sel == 61) y = (ref >= 0.0f) ? x : x * (ref + 1.0f);
    else if constexpr (sel == 62) y = (ref >= 0.0f) ? 0.0f : x * (ref + 1.0f);
    else if constexpr (sel == 70) y = (x >= 0.0f) ? seluScale * x : (seluScale * seluAlpha) * (expf(x) - 1.0f);
    else if constexpr (sel == 71) y = (ref >= 0.0f) ? x * seluScale : x * (ref + seluScale * seluAlpha);
    else if constexpr (sel == 72) y = (ref >= 0.0f) ? 0.0f : x * (ref + seluScale * seluAlpha);
    else if constexpr (sel == 80) y = (x > expRange) ? x : logf(expf(x) + 1.0f);
    else if constexpr (sel == 81) y = x * (1.0f - expf(-ref));
    else if constexpr (sel == 82) { float c = expf(-ref); y = x * c * (1.0f - c); }
    else if constexpr (sel == 90) y = (x < -expRange) ? 0.0f : x / (expf(-x) + 1.0f);
    else if constexpr (sel == 91) { float c = expf(ref); float d = c + 1.0f; y = (ref > halfExpRange) ? x : x * c * (ref + d) / (d * d); }
    else if constexpr (sel == 92) { float c = expf(ref); float d = c + 1.0f; y = (ref > halfExpRange) ? 0.0f : x * c * (ref * (2.0f - d) + 2.0f * d) / (d * d * d); }
    else y = x;
    return y;
}

template <int ACT, int GRAD>
__device__ __forceinline__ float fba_one(const FbaArgs& a, float x, float bias, float ref) {
    x += bias;
    // fused_bias_act.cu:59-60: ref is stored post-gain (except swish, which keeps x).
    if (a.gain != 0.0f && ACT != 9) ref /= a.gain;
    return fba_eval<ACT, GRAD>(x, ref, a.alpha) * a.gain;
}

// VEC = true: sizeX % 4 == 0, 16 B aligned, and the four elements of a float4 either
// share one bias entry (stepB % 4 == 0) or map to four consecutive ones
// (stepB == 1 && sizeB % 4 == 0).
template <int ACT, int GRAD, bool VEC>
__global__ __launch_bounds__(256) void fba_kernel(FbaArgs a) {
    if constexpr (VEC) {
        const int n4 = a.sizeX >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(a.x);
        const float4* r4 = reinterpret_cast<const float4*>(a.ref);
        float4* y4 = reinterpret_cast<float4*>(a.y);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
            float4 x = x4[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (GRAD != 0) r = r4[i];
            float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.b) {
                const int e = i << 2;
                if (a.stepB == 1) {
                    bb = *reinterpret_cast<const float4*>(a.b + (e % a.sizeB));
                } else {
                    const float bv = a.b[(e / a.stepB) % a.sizeB];
                    bb = make_float4(bv, bv, bv, bv);
                }
            }
            float4 y;
            y.x = fba_one<ACT, GRAD>(a, x.x, bb.x, r.x);
            y.y = fba_one<ACT, GRAD>(a, x.y, bb.y, r.y);
            y.z = fba_one<ACT, GRAD>(a, x.z, bb.z, r.z);
            y.w = fba_one<ACT, GRAD>(a, x.w, bb.w, r.w);
            y4[i] = y;
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.sizeX; i += gridDim.x * blockDim.x) {
            const float bias = a.b ? a.b[(i / a.stepB) % a.sizeB] : 0.0f;
            const float ref = (GRAD != 0) ? a.ref[i] : 0.0f;
            a.y[i] = fba_one<ACT, GRAD>(a, a.x[i], bias, ref);
        }
    }
}

template <int ACT, int GRAD>
void fba_launch(hipStream_t stream, const FbaArgs& a, bool vec) {
    if (vec) {
        const int n4 = a.sizeX >> 2;
        const int grid = std::min(igan::ceil_div(n4, 256), 256 * 16);
        hipLaunchKernelGGL((fba_kernel<ACT, GRAD, true>), dim3(grid), dim3(256), 0, stream, a);
    } else {
        const int grid = std::min(igan::ceil_div(a.sizeX, 256), 256 * 16);
        hipLaunchKernelGGL((fba_kernel<ACT, GRAD, false>), dim3(grid), dim3(256), 0, stream, a);
    }
}

// The reference registers the op for float and half (fused_bias_act.cu:185-186); its kernel loads T, computes in float and
// stores (T)y (:56-61,113).  Same here for T = half: x / b / ref / y hold IEEE halves, 8 of them per 16 B lane when aligned.
template <int ACT, int GRAD>
__global__ __launch_bounds__(256) void fba_f16_kernel(FbaArgs a) {
    const __half* x = reinterpret_cast<const __half*>(a.x);
    const __half* b = reinterpret_cast<const __half*>(a.b);
    const __half* ref = reinterpret_cast<const __half*>(a.ref);
    __half* y = reinterpret_cast<__half*>(a.y);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.sizeX; i += gridDim.x * blockDim.x) {
        const float bias = b ? __half2float(b[(i / a.stepB) % a.sizeB]) : 0.0f;
        const float r = (GRAD != 0) ? __half2float(ref[i]) : 0.0f;
        y[i] = __float2half(fba_one<ACT, GRAD>(a, __half2float(x[i]), bias, r));
    }
}

template <int ACT>
void fba_f16_dispatch_grad(hipStream_t stream, const FbaArgs& a, int grad) {
    const int grid = std::min(igan::ceil_div(a.sizeX, 256), 256 * 16);
    switch (grad) {
        case 0: hipLaunchKernelGGL((fba_f16_kernel<ACT, 0>), dim3(grid), dim3(256), 0, stream, a); break;
        case 1: hipLaunchKernelGGL((fba_f16_kernel<ACT, 1>), dim3(grid), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((fba_f16_kernel<ACT, 2>), dim3(grid), dim3(256), 0, stream, a); break;
    }
}

template <int ACT>
void fba_dispatch_grad(hipStream_t stream, const FbaArgs& a, int grad, bool vec) {
    switch (grad) {
        case 0: fba_launch<ACT, 0>(stream, a, vec); break;
        case 1: fba_launch<ACT, 1>(stream, a, vec); break;
        default: fba_launch<ACT, 2>(stream, a, vec); break;
    }
}

// ---- bias gradient -------------------------------------------------------------
// Layout view: x[outer][sizeB][stepB]; db[c] = sum_{o,s} dx[o][c][s].
// Pass 1: block (chunk j, channel tile) sums its rows into partial[j][c];
// pass 2: db[c] = sum_j partial[j][c] in fixed order.
constexpr int BG_ROWS_PER_BLOCK = 256;

// stepB == 1: rows of sizeB contiguous channels. thread <-> channel (coalesced
// across the wave), loop over the block's rows.
__global__ __launch_bounds__(256) void bias_grad_rows_kernel(const float* dx, float* partial, int rows, int sizeB) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = blockIdx.y * BG_ROWS_PER_BLOCK;
    const int r1 = min(r0 + BG_ROWS_PER_BLOCK, rows);
    if (c >= sizeB) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += dx[(long long)(r + 0) * sizeB + c];
        s1 += dx[(long long)(r + 1) * sizeB + c];
        s2 += dx[(long long)(r + 2) * sizeB + c];
        s3 += dx[(long long)(r + 3) * sizeB + c];
    }
    for (; r < r1; r++) s0 += dx[(long long)r * sizeB + c];
    partial[(long long)blockIdx.y * sizeB + c] = (s0 + s1) + (s2 + s3);
}

// general stepB (> 1): one block per (outer index, channel): wave-shuffle + LDS tree.
__global__ __launch_bounds__(256) void bias_grad_planes_kernel(const float* dx, float* partial, int sizeB, int stepB) {
    const int c = blockIdx.x;
    const int o = blockIdx.y;
    const float* src = dx + ((long long)o * sizeB + c) * stepB;
    float s = 0.f;
    for (int i = threadIdx.x; i < stepB; i += blockDim.x) s += src[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)o * sizeB + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float* partial, float* db, int chunks, int sizeB) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sizeB) return;
    // four partial sums (fixed order): four loads in flight instead of `chunks` dependent L2 round trips
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int j = 0;
    for (; j + 4 <= chunks; j += 4) {
        s0 += partial[(long long)j * sizeB + c];       s1 += partial[(long long)(j + 1) * sizeB + c];
        s2 += partial[(long long)(j + 2) * sizeB + c]; s3 += partial[(long long)(j + 3) * sizeB + c];
    }
    for (; j < chunks; j++) s0 += partial[(long long)j * sizeB + c];
    db[c] = (s0 + s1) + (s2 + s3);
}

int bias_grad_chunks(int sizeX, int sizeB, int stepB) {
    const int outer = sizeX / (sizeB * stepB);
    if (stepB == 1) return igan::ceil_div(outer, BG_ROWS_PER_BLOCK);
    return outer;
}

}  // namespace

extern "C" int igan_fused_bias_act(igan_stream_t stream_, const igan_fused_bias_act_params* p) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(p != nullptr, "fused_bias_act: null params");
    IGAN_REQUIRE(p->x && p->y, "fused_bias_act: null buffer");
    // fused_bias_act.cu:134-136
    IGAN_REQUIRE(p->grad >= 0, "grad must be non-negative");
    IGAN_REQUIRE(p->act >= 0, "act must be non-negative");
    IGAN_REQUIRE(p->grad <= 2, "fused_bias_act: third-order gradients are not supported");
    IGAN_REQUIRE(p->sizeX >= 0, "x is too large");
    // fused_bias_act.cu:150-153
    if (p->b) {
        IGAN_REQUIRE(p->sizeB >= 1 && p->stepB >= 1, "b has wrong number of elements");
        IGAN_REQUIRE(p->sizeX % (p->sizeB * p->stepB) == 0, "b has wrong number of elements");
    }
    IGAN_REQUIRE((p->ref != nullptr) == (p->grad != 0), "ref has wrong number of elements");
    if (p->sizeX == 0) return IGAN_OK;

    FbaArgs a;
    a.x = p->x; a.b = p->b; a.ref = p->ref; a.y = p->y;
    a.alpha = p->alpha; a.gain = p->gain;
    a.sizeX = p->sizeX; a.sizeB = p->b ? p->sizeB : 1; a.stepB = p->b ? p->stepB : 1;

    uintptr_t al = (uintptr_t)p->x | (uintptr_t)p->y | (uintptr_t)p->ref;
    bool vec = (p->sizeX % 4 == 0) && ((al & 15) == 0);
    if (vec && p->b) {
        const bool consecutive = (a.stepB == 1) && (a.sizeB % 4 == 0) && (((uintptr_t)p->b & 15) == 0);
        const bool shared = (a.stepB % 4 == 0);
        vec = consecutive || shared;
    }
    switch (p->act) {
        case 1: fba_dispatch_grad<1>(stream, a, p->grad, vec); break;
        case 2: fba_dispatch_grad<2>(stream, a, p->grad, vec); break;
        case 3: fba_dispatch_grad<3>(stream, a, p->grad, vec); break;
        case 4: fba_dispatch_grad<4>(stream, a, p->grad, vec); break;
        case 5: fba_dispatch_grad<5>(stream, a, p->grad, vec); break;
        case 6: fba_dispatch_grad<6>(stream, a, p->grad, vec); break;
        case 7: fba_dispatch_grad<7>(stream, a, p->grad, vec); break;
        case 8: fba_dispatch_grad<8>(stream, a, p->grad, vec); break;
        case 9: fba_dispatch_grad<9>(stream, a, p->grad, vec); break;
        default: fba_dispatch_grad<1>(stream, a, p->grad, vec); break;  // fused_bias_act.cu:67 `default:` == linear
    }
    IGAN_LAUNCH_CHECK("fused_bias_act launch");
    return IGAN_OK;
}

extern "C" int igan_fused_bias_act_f16(igan_stream_t stream_, const igan_fused_bias_act_params* p) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(p != nullptr, "fused_bias_act_f16: null params");
    IGAN_REQUIRE(p->x && p->y, "fused_bias_act_f16: null buffer");
    IGAN_REQUIRE(p->grad >= 0, "grad must be non-negative");
    IGAN_REQUIRE(p->grad <= 2, "fused_bias_act: third-order gradients are not supported");
    IGAN_REQUIRE(p->sizeX >= 0, "x is too large");
    if (p->b) {
        IGAN_REQUIRE(p->sizeB >= 1 && p->stepB >= 1, "b has wrong number of elements");
        IGAN_REQUIRE(p->sizeX % (p->sizeB * p->stepB) == 0, "b has wrong number of elements");
    }
    IGAN_REQUIRE((p->ref != nullptr) == (p->grad != 0), "ref has wrong number of elements");
    if (p->sizeX == 0) return IGAN_OK;
    FbaArgs a;
    a.x = p->x; a.b = p->b; a.ref = p->ref; a.y = p->y;       // halves behind the float-typed fields (see include/igan_hip.h)
    a.alpha = p->alpha; a.gain = p->gain;
    a.sizeX = p->sizeX; a.sizeB = p->b ? p->sizeB : 1; a.stepB = p->b ? p->stepB : 1;
    switch (p->act) {
        case 2: fba_f16_dispatch_grad<2>(stream, a, p->grad); break;
        case 3: fba_f16_dispatch_grad<3>(stream, a, p->grad); break;
        case 4: fba_f16_dispatch_grad<4>(stream, a, p->grad); break;
        case 5: fba_f16_dispatch_grad<5>(stream, a, p->grad); break;
        case 6: fba_f16_dispatch_grad<6>(stream, a, p->grad); break;
        case 7: fba_f16_dispatch_grad<7>(stream, a, p->grad); break;
        case 8: fba_f16_dispatch_grad<8>(stream, a, p->grad); break;
        case 9: fba_f16_dispatch_grad<9>(stream, a, p->grad); break;
        default: fba_f16_dispatch_grad<1>(stream, a, p->grad); break;
    }
    IGAN_LAUNCH_CHECK("fused_bias_act_f16 launch");
    return IGAN_OK;
}

extern "C" size_t igan_bias_grad_workspace_floats(int sizeX, int sizeB, int stepB) {
    if (sizeX <= 0 || sizeB <= 0 || stepB <= 0) return 0;
    return (size_t)bias_grad_chunks(sizeX, sizeB, stepB) * (size_t)sizeB;
}

extern "C" int igan_bias_grad(igan_stream_t stream_, const float* dx, float* db, float* partial,
                              int sizeX, int sizeB, int stepB) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(dx && db && partial, "bias_grad: null buffer");
    IGAN_REQUIRE(sizeX >= 1 && sizeB >= 1 && stepB >= 1, "bias_grad: sizes must be positive");
    IGAN_REQUIRE(sizeX % (sizeB * stepB) == 0, "b has wrong number of elements");
    const int chunks = bias_grad_chunks(sizeX, sizeB, stepB);
    if (stepB == 1) {
        const int rows = sizeX / sizeB;
        dim3 grid(ceil_div(sizeB, 256), chunks);
        hipLaunchKernelGGL(bias_grad_rows_kernel, grid, dim3(256), 0, stream, dx, partial, rows, sizeB);
    } else {
        dim3 grid(sizeB, chunks);
        hipLaunchKernelGGL(bias_grad_planes_kernel, grid, dim3(256), 0, stream, dx, partial, sizeB, stepB);
    }
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(ceil_div(sizeB, 256)), dim3(256), 0, stream, partial, db, chunks, sizeB);
    IGAN_LAUNCH_CHECK("bias_grad launch");
    return IGAN_OK;
}
