// upfirdn2d for gfx950: pad -> zero-insert upsample -> FIR -> decimate on
// channel-minor tensors [majorDim, inH, inW, minorDim].
//
// Behavioural contract: dnnlib/tflib/ops/upfirdn_2d.cu:64-117 (index math of the
// general kernel) and :232-307 (argument checks, output size).  The design is not
// the reference's: the reference runs with minorDim == 1 (NCHW reshaped to
// [N*C,H,W,1], upfirdn_2d.py:357-361) and stages one channel plane per block in
// shared memory.  Here the channel axis is minor and contiguous, so the hot case
// (up = down = 1, <= 4x4 taps: every conv pre/post filter of G and D and their
// gradients) is a pure HBM stream: each lane owns a float4 of channels and a
// vertical strip of TY output rows, keeps the TY partial sums in registers and
// reads each input row of its column window exactly once (4 x 16 B loads per row,
// horizontally adjacent lanes share them through L1).  A wave covers 64
// consecutive float4 = 1 KiB of one or more adjacent pixels, so global loads and
// stores are fully coalesced.
#include "igan_common.h"
#include <hip/hip_fp16.h>
#include <cstdlib>

namespace {

struct FirTaps {
    float k[64];  // flipped taps, row-major [kernelH][kernelW] (or 4x4 zero-extended)
};

struct UpfirdnArgs {
    const float* x;
    float* y;
    int upx, upy, downx, downy;
    int padx0, pady0;
    int majorDim, inH, inW, minorDim;
    int kernelH, kernelW;
    int outH, outW;
    int xcd_remap;          // XCD-aware block order in the FIR fast path (A/B switch IGAN_FIR_XCD)
    // optional epilogue of the FIR fast path (igan_upfirdn2d_ban): y = act(fir + noise[m, oy, ox] * strength + bias[c]) * gain
    const float* noise;
    const float* strength;
    const float* bias;
    int noise_bcast, act;
    float act_alpha, act_gain;
};

__host__ __device__ __forceinline__ int floor_div(int a, int b) {
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

__device__ __forceinline__ float io_load(const float* p) { return *p; }
__device__ __forceinline__ float io_load(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void io_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void io_store(__half* p, float v) { *p = __float2half(v); }

// General path: one thread per output element, channel index fastest.  Written from the op's definition (upfirdn_2d.py:19-60):
//     y[m, oy, ox, c] = sum_{jy, jx} Z[m, oy * downy + jy - pady0, ox * downx + jx - padx0, c] * taps[jy][jx]
// where Z is x with (up - 1) zeros inserted after every sample (Z[u] = x[u / up] when u >= 0, u % up == 0 and u / up < in; else 0:
// negative pads crop, positive pads add zeros) and `taps` is the filter in correlation order (the host flips it once).  Per axis only
// every up-th tap meets a sample: the first one is the smallest j >= 0 whose Z coordinate is a non-negative multiple of `up`, and
// from there tap j += up walks sample i += 1.  Same terms in the same order (input rows, then columns, ascending) as the reference's
// kernels (upfirdn_2d.cu:64-207), so the float accumulation gives the same value.  T = float or half (the reference registers both,
// upfirdn_2d.cu:323-324; loads are widened to float, the accumulation is float, the store rounds to T: :101,114).
struct TapWalk { int j0, i0, n; };      // first tap that lands on a sample, that sample, how many (tap, sample) pairs follow

__device__ __forceinline__ TapWalk tap_walk(int o, int down, int up, int pad0, int taps, int in) {
    const int u0 = o * down - pad0;                                 // Z coordinate under tap 0
    const int j0 = (u0 >= 0) ? (up - u0 % up) % up : -u0;           // u0 + j0 is the first non-negative multiple of up
    const int i0 = (u0 + j0) / up;
    const int by_taps = (j0 < taps) ? (taps - j0 + up - 1) / up : 0;
    const int by_samples = (i0 < in) ? in - i0 : 0;
    return TapWalk{j0, i0, min(by_taps, by_samples)};
}

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(UpfirdnArgs a, FirTaps taps) {
    const T* xin = reinterpret_cast<const T*>(a.x);
    T* yout = reinterpret_cast<T*>(a.y);
    const long long total = (long long)a.majorDim * a.outH * a.outW * a.minorDim;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int c = (int)(t % a.minorDim); t /= a.minorDim;
        const int ox = (int)(t % a.outW);    t /= a.outW;
        const int oy = (int)(t % a.outH);    t /= a.outH;
        const int m = (int)t;
        const TapWalk wy = tap_walk(oy, a.downy, a.upy, a.pady0, a.kernelH, a.inH);
        const TapWalk wx = tap_walk(ox, a.downx, a.upx, a.padx0, a.kernelW, a.inW);
        float v = 0.0f;
        for (int sy = 0; sy < wy.n; sy++) {
            const T* xrow = xin + (((long long)m * a.inH + wy.i0 + sy) * a.inW + wx.i0) * a.minorDim + c;
            const float* krow = taps.k + (wy.j0 + sy * a.upy) * a.kernelW + wx.j0;
            for (int sx = 0; sx < wx.n; sx++)
                v += io_load(xrow + (long long)sx * a.minorDim) * krow[sx * a.upx];
        }
        io_store(yout + idx, v);
    }
}

// Fast path: up = down = 1, taps zero-extended to 4x4, minorDim % 4 == 0.
//   y[m,oy,ox,c] = sum_{ky,kx<4} x[m, oy+ky-pady0, ox+kx-padx0, c] * kf[ky][kx]
template <int TY, int TX, bool EPI = false>
__global__ __launch_bounds__(256) void upfirdn2d_fir4_kernel(UpfirdnArgs a, FirTaps taps) {
    // a lane owns TY output rows x TX adjacent output columns of one channel quad: (TY+3) x (TX+3) input loads feed
    // TY*TX outputs (5.5 loads per output at 8x1, 3.4 at 8x2: the kernel is L1-request bound, not HBM bound)
    const int cvecs = a.minorDim >> 2;
    const int strips = (a.outH + TY - 1) / TY;
    const int xgroups = (a.outW + TX - 1) / TX;
    const long long total = (long long)a.majorDim * strips * xgroups * cvecs;
    // XCD-aware block order (as in conv2d_mfma.hip): workgroups are dealt round-robin to the 8 XCDs, so with the plain order the
    // blocks that share halo rows (neighbouring strips) sit on eight different L2s and every halo row is fetched from HBM twice;
    // here XCD x works through one contiguous range of logical blocks.  Placement only: no result depends on it.
    int bid = blockIdx.x;
    if (a.xcd_remap) {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const long long idx = (long long)bid * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long long t = idx;
    const int cv = (int)(t % cvecs); t /= cvecs;
    const int ox0 = (int)(t % xgroups) * TX; t /= xgroups;
    const int strip = (int)(t % strips); t /= strips;
    const int m = (int)t;
    const int oy0 = strip * TY;

    float4 acc[TY][TX];
#pragma unroll
    for (int i = 0; i < TY; i++)
#pragma unroll
        for (int j = 0; j < TX; j++) acc[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int ix0 = ox0 - a.padx0;
    const float4* xbase = reinterpret_cast<const float4*>(a.x) + (long long)m * a.inH * a.inW * cvecs + cv;

#pragma unroll
    for (int r = 0; r < TY + 3; r++) {
        const int iy = oy0 + r - a.pady0;
        float4 v[TX + 3];
        const bool rowok = (iy >= 0) & (iy < a.inH);
#pragma unroll
        for (int c = 0; c < TX + 3; c++) {
            const int ix = ix0 + c;
            const bool ok = rowok & (ix >= 0) & (ix < a.inW);
            v[c] = ok ? xbase[((long long)iy * a.inW + ix) * cvecs] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const int o = r - ky;  // output row (relative) fed by this input row via tap ky
            if (o >= 0 && o < TY) {
#pragma unroll
                for (int j = 0; j < TX; j++)
#pragma unroll
                    for (int kx = 0; kx < 4; kx++) {
                        const float kv = taps.k[ky * 4 + kx];
                        acc[o][j].x += v[j + kx].x * kv;
                        acc[o][j].y += v[j + kx].y * kv;
                        acc[o][j].z += v[j + kx].z * kv;
                        acc[o][j].w += v[j + kx].w * kv;
                    }
            }
        }
    }

    float4* ybase = reinterpret_cast<float4*>(a.y) + (long long)m * a.outH * a.outW * cvecs + cv;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    float st = 0.f;
    if constexpr (EPI) {
        if (a.bias) bb = *reinterpret_cast<const float4*>(a.bias + 4 * cv);
        if (a.noise) st = a.strength[0];
    }
#pragma unroll
    for (int i = 0; i < TY; i++) {
        const int oy = oy0 + i;
#pragma unroll
        for (int j = 0; j < TX; j++)
            if (oy < a.outH && ox0 + j < a.outW) {
                float4 v = acc[i][j];
                if constexpr (EPI) {    // the layer epilogue that follows the FIR (networks_stylegan2.py:351-357)
                    const float nz = a.noise ? a.noise[((long long)(a.noise_bcast ? 0 : m) * a.outH + oy) * a.outW + ox0 + j] * st : 0.f;
                    const float e[4] = {v.x + nz + bb.x, v.y + nz + bb.y, v.z + nz + bb.z, v.w + nz + bb.w};
                    float o[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float t = e[q];
                        if (a.act == 2) t = t > 0.f ? t : 0.f;
                        if (a.act == 3) t = t > 0.f ? t : t * a.act_alpha;
                        o[q] = t * a.act_gain;
                    }
                    v = make_float4(o[0], o[1], o[2], o[3]);
                }
                ybase[((long long)oy * a.outW + ox0 + j) * cvecs] = v;
            }
    }
}

}  // namespace

namespace {
struct FirEpilogue { const float* noise; const float* strength; const float* bias; int noise_bcast, act; float alpha, gain; };
int upfirdn_launch(hipStream_t stream, const igan_upfirdn2d_params* p, const FirEpilogue* epi, bool half_io = false);
}  // namespace

extern "C" int igan_upfirdn2d(igan_stream_t stream_, const igan_upfirdn2d_params* p) {
    return upfirdn_launch((hipStream_t)stream_, p, nullptr);
}

extern "C" int igan_upfirdn2d_f16(igan_stream_t stream_, const igan_upfirdn2d_params* p) {
    return upfirdn_launch((hipStream_t)stream_, p, nullptr, true);
}

extern "C" int igan_upfirdn2d_ban(igan_stream_t stream_, const igan_upfirdn2d_params* p, const float* noise, const float* strength,
                                  int noise_bcast, const float* bias, int act, float alpha, float gain) {
    using namespace igan;
    IGAN_REQUIRE(act >= 1 && act <= 3, "upfirdn2d_ban: act must be 1 (linear), 2 (relu) or 3 (lrelu)");
    IGAN_REQUIRE(gain > 0.0f, "upfirdn2d_ban: gain must be positive");
    IGAN_REQUIRE((noise == nullptr) == (strength == nullptr), "upfirdn2d_ban: noise and strength go together");
    IGAN_REQUIRE(((uintptr_t)bias & 15) == 0, "upfirdn2d_ban: bias must be 16-byte aligned");
    FirEpilogue e{noise, strength, bias, noise_bcast, act, alpha, gain};
    return upfirdn_launch((hipStream_t)stream_, p, &e);
}

namespace {
int upfirdn_launch(hipStream_t stream, const igan_upfirdn2d_params* p, const FirEpilogue* epi, bool half_io) {
    using namespace igan;
    IGAN_REQUIRE(p != nullptr, "upfirdn2d: null params");
    IGAN_REQUIRE(p->x && p->k && p->y, "upfirdn2d: null buffer");
    // upfirdn_2d.cu:228-229
    IGAN_REQUIRE(p->upx >= 1 && p->upy >= 1, "upx and upy must be at least 1x1");
    IGAN_REQUIRE(p->downx >= 1 && p->downy >= 1, "downx and downy must be at least 1x1");
    // upfirdn_2d.cu:252
    IGAN_REQUIRE(p->kernelW >= 1 && p->kernelH >= 1, "kernel must be at least 1x1");
    IGAN_REQUIRE(p->kernelW * p->kernelH <= 64, "kernel too large (max 64 taps)");
    IGAN_REQUIRE(p->majorDim >= 1 && p->inH >= 1 && p->inW >= 1 && p->minorDim >= 1, "input must have rank 4 with positive dims");
    // upfirdn_2d.cu:254-256
    const int outW = (p->inW * p->upx + p->padx0 + p->padx1 - p->kernelW + p->downx) / p->downx;
    const int outH = (p->inH * p->upy + p->pady0 + p->pady1 - p->kernelH + p->downy) / p->downy;
    IGAN_REQUIRE(outW >= 1 && outH >= 1, "output must be at least 1x1");
    IGAN_REQUIRE(outW == p->outW && outH == p->outH, "upfirdn2d: outH/outW (%d,%d) do not match derived (%d,%d)", p->outH, p->outW, outH, outW);
    // upfirdn_2d.cu:243,266
    const long long in_elems = (long long)p->majorDim * p->inH * p->inW * p->minorDim;
    const long long out_elems = (long long)p->majorDim * outH * outW * p->minorDim;
    IGAN_REQUIRE(in_elems <= INT32_MAX, "input too large");
    IGAN_REQUIRE(out_elems <= INT32_MAX, "output too large");

    UpfirdnArgs a;
    a.x = p->x; a.y = p->y;
    a.upx = p->upx; a.upy = p->upy; a.downx = p->downx; a.downy = p->downy;
    a.padx0 = p->padx0; a.pady0 = p->pady0;
    a.majorDim = p->majorDim; a.inH = p->inH; a.inW = p->inW; a.minorDim = p->minorDim;
    a.kernelH = p->kernelH; a.kernelW = p->kernelW;
    a.outH = outH; a.outW = outW;
    static const bool fir_xcd = !(getenv("IGAN_FIR_XCD") && atoi(getenv("IGAN_FIR_XCD")) == 0);     // A/B switch
    a.xcd_remap = fir_xcd ? 1 : 0;
    a.noise = epi ? epi->noise : nullptr; a.strength = epi ? epi->strength : nullptr; a.bias = epi ? epi->bias : nullptr;
    a.noise_bcast = epi ? epi->noise_bcast : 0; a.act = epi ? epi->act : 0;
    a.act_alpha = epi ? epi->alpha : 0.f; a.act_gain = epi ? epi->gain : 1.f;

    const bool aligned = (((uintptr_t)p->x | (uintptr_t)p->y) & 15) == 0;
    const bool fast = !half_io && p->upx == 1 && p->upy == 1 && p->downx == 1 && p->downy == 1 &&
                      p->kernelH <= 4 && p->kernelW <= 4 && (p->minorDim % 4) == 0 && aligned;
    FirTaps taps;
    for (int i = 0; i < 64; i++) taps.k[i] = 0.0f;
    if (fast) {
        // flipped taps, zero-extended to 4x4 (extra taps read padding-or-data times 0)
        for (int ky = 0; ky < p->kernelH; ky++)
            for (int kx = 0; kx < p->kernelW; kx++)
                taps.k[ky * 4 + kx] = p->k[(p->kernelH - 1 - ky) * p->kernelW + (p->kernelW - 1 - kx)];
        const int cvecs = p->minorDim / 4;
        static const int tx = getenv("IGAN_FIR_TX") ? atoi(getenv("IGAN_FIR_TX")) : 2;   // A/B switch
        if (outH >= 8 && outW >= 16 && tx >= 2) {        // 8x2 outputs per lane (8x4 and 4x4 were slower: registers)
            const long long total = (long long)p->majorDim * ceil_div(outH, 8) * ceil_div(outW, 2) * cvecs;
            if (epi) hipLaunchKernelGGL((upfirdn2d_fir4_kernel<8, 2, true>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
            else hipLaunchKernelGGL((upfirdn2d_fir4_kernel<8, 2>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
        } else if (outH >= 8) {
            const long long total = (long long)p->majorDim * ceil_div(outH, 8) * outW * cvecs;
            if (epi) hipLaunchKernelGGL((upfirdn2d_fir4_kernel<8, 1, true>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
            else hipLaunchKernelGGL((upfirdn2d_fir4_kernel<8, 1>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
        } else {
            const long long total = (long long)p->majorDim * ceil_div(outH, 2) * outW * cvecs;
            if (epi) hipLaunchKernelGGL((upfirdn2d_fir4_kernel<2, 1, true>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
            else hipLaunchKernelGGL((upfirdn2d_fir4_kernel<2, 1>), dim3((int)ceil_div_ll(total, 256)), dim3(256), 0, stream, a, taps);
        }
    } else {
        if (epi) return fail(IGAN_ERR_UNSUPPORTED, "upfirdn2d_ban: the epilogue rides on the FIR fast path only (up = down = 1, taps <= 4x4, minorDim %% 4 == 0, 16-byte aligned)");
        for (int ky = 0; ky < p->kernelH; ky++)
            for (int kx = 0; kx < p->kernelW; kx++)
                taps.k[ky * p->kernelW + kx] = p->k[(p->kernelH - 1 - ky) * p->kernelW + (p->kernelW - 1 - kx)];
        const int grid = (int)std::min<long long>(ceil_div_ll(out_elems, 256), 256 * 32);
        if (half_io) {
            // the reference's half instantiation holds the taps in half too (k: T, upfirdn_2d.cu:312): round them the same way
            for (int i = 0; i < 64; i++) taps.k[i] = (float)(_Float16)taps.k[i];
            hipLaunchKernelGGL(upfirdn2d_generic_kernel<__half>, dim3(grid), dim3(256), 0, stream, a, taps);
        } else {
            hipLaunchKernelGGL(upfirdn2d_generic_kernel<float>, dim3(grid), dim3(256), 0, stream, a, taps);
        }
    }
    IGAN_LAUNCH_CHECK("upfirdn2d launch");
    return IGAN_OK;
}
}  // namespace
