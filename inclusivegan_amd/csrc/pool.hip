// 2x2 max-pool between the VGG blocks of the LPIPS network (the pooling of metrics/vgg16_zhang_perceptual.pkl's
// VGG16, used through lpips.get_output_for at training/loss.py:31,41 -- the pickle is absent, so the network is a restatement
// of Zhang et al. 2018 / Simonyan & Zisserman, parity unpinned), channel-minor, and its gradient.
//
// The feature map that is pooled is also an LPIPS tap (relu1_2 ... relu4_3 feed both the distance and the next block), so the
// gradient kernel takes the tap's gradient as a second input and writes
//     dx[n, 2i+a, 2j+b, c] = dskip[n, 2i+a, 2j+b, c] + (argmax(n, i, j, c) == (a, b) ? dy[n, i, j, c] : 0)
// in one pass: the pooled gradient is never materialised at full resolution and the two-consumer sum is not a separate pass.
// Ties (frequent after ReLU: all four zeros) go to the first maximum in window order (0,0) (0,1) (1,0) (1,1), NaN wins: the rule
// of the framework pooling the oracle uses.
// MI355X design: HBM-bound streaming; a lane owns 4 channels (16 B) of one output pixel, reads the four input pixels with
// 16 B loads (lanes of a wave cover consecutive channels, then consecutive output pixels: 128..2048 B contiguous per row).
#include "igan_common.h"

namespace {

__device__ __forceinline__ void pick(float v, int k, float& m, int& am) {
    if (v > m || v != v) { m = v; am = k; }
}

template <bool BWD>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ dskip, float* __restrict__ out,
                                                         int OH, int OW, int C4, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const long long n = r / OH;
    const int W = 2 * OW;
    const size_t row = (size_t)C4 * W;                                  // float4 per input row
    const size_t base = ((size_t)n * 2 * OH + 2 * oy) * row + (size_t)(2 * ox) * C4 + c4;
    const float4* xp = reinterpret_cast<const float4*>(x);
    const float4 v00 = xp[base], v01 = xp[base + C4], v10 = xp[base + row], v11 = xp[base + row + C4];
    float m[4] = {v00.x, v00.y, v00.z, v00.w};
    int am[4] = {0, 0, 0, 0};
    pick(v01.x, 1, m[0], am[0]); pick(v01.y, 1, m[1], am[1]); pick(v01.z, 1, m[2], am[2]); pick(v01.w, 1, m[3], am[3]);
    pick(v10.x, 2, m[0], am[0]); pick(v10.y, 2, m[1], am[1]); pick(v10.z, 2, m[2], am[2]); pick(v10.w, 2, m[3], am[3]);
    pick(v11.x, 3, m[0], am[0]); pick(v11.y, 3, m[1], am[1]); pick(v11.z, 3, m[2], am[2]); pick(v11.w, 3, m[3], am[3]);
    if constexpr (!BWD) {
        reinterpret_cast<float4*>(out)[i] = make_float4(m[0], m[1], m[2], m[3]);
    } else {
        const float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float gv[4] = {g.x, g.y, g.z, g.w};
        const float4* sp = reinterpret_cast<const float4*>(dskip);
        float4* op = reinterpret_cast<float4*>(out);
        const size_t off[4] = {base, base + C4, base + row, base + row + C4};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float4 s = dskip ? sp[off[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
            s.x += (am[0] == k) ? gv[0] : 0.f;
            s.y += (am[1] == k) ? gv[1] : 0.f;
            s.z += (am[2] == k) ? gv[2] : 0.f;
            s.w += (am[3] == k) ? gv[3] : 0.f;
            op[off[k]] = s;
        }
    }
}

int check_dims(const void* x, const void* y, int N, int H, int W, int C) {
    IGAN_REQUIRE(x != nullptr && y != nullptr, "maxpool2x2: null pointer");
    IGAN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2: sizes must be positive");
    IGAN_REQUIRE(H % 2 == 0 && W % 2 == 0, "maxpool2x2: H and W must be even (got %d x %d)", H, W);
    IGAN_REQUIRE(C % 4 == 0, "maxpool2x2: C must be a multiple of 4 (got %d)", C);
    IGAN_REQUIRE((long long)N * H * W * C < (1LL << 31), "maxpool2x2: too many elements");
    return IGAN_OK;
}

}  // namespace

extern "C" int igan_maxpool2x2_fwd(igan_stream_t stream_, const float* x, float* y, int N, int H, int W, int C) {
    if (int rc = check_dims(x, y, N, H, W, C)) return rc;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL((maxpool2x2_kernel<false>), dim3((unsigned)igan::ceil_div_ll(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                       x, nullptr, nullptr, y, H / 2, W / 2, C / 4, total);
    IGAN_LAUNCH_CHECK("maxpool2x2_fwd");
    return IGAN_OK;
}

extern "C" int igan_maxpool2x2_bwd(igan_stream_t stream_, const float* x, const float* dy, const float* dskip, float* dx,
                                   int N, int H, int W, int C) {
    if (int rc = check_dims(x, dx, N, H, W, C)) return rc;
    IGAN_REQUIRE(dy != nullptr, "maxpool2x2_bwd: null dy");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL((maxpool2x2_kernel<true>), dim3((unsigned)igan::ceil_div_ll(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                       x, dy, dskip, dx, H / 2, W / 2, C / 4, total);
    IGAN_LAUNCH_CHECK("maxpool2x2_bwd");
    return IGAN_OK;
}
