// MEASURE-ONLY probe (not part of libigan_hip.so, not on any product path): how fast would the headline convolution's GEMM run
// if its fp32 operands were split into bf16 pieces and multiplied on the bf16 matrix pipe -- and how far from fp32 are the results?
//
//   x = x0 + x1 (+ x2), xi = bf16(x - x0 - ... - x(i-1));   a*b ~ a0*b0 + a0*b1 + a1*b0            ("bf16x3", 3 MFMAs per product)
//                                                           ... + a0*b2 + a2*b0 + a1*b1            ("bf16x6", 6 MFMAs)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (32 cycles per 32x32x16 = 16x the FLOP/clk of v_mfma_f32_32x32x2_f32).
//
// Problem = the north-star shape as a GEMM with the convolution's reuse pattern: M = 98304 "pixels" (6 x 128 x 128), N = 128,
// K = 9 taps x 128 channels; A[m][tap*128 + c] = X[(m + shift[tap]) mod M][c] with shift = dy*128 + dx, dy,dx in {-1,0,1} (every
// X row is read by nine taps, mostly out of L2, as in the implicit-GEMM kernel; the image-border zero padding is left out: it does
// not change the work).  B = W[n][k] (k contiguous).  Output C[m][n] fp32.
//
// Kernel: 128x128 tile, 8 waves (2 x 4 of 64x32), K chunks of 32 fp32; register-staged: global fp32 -> split in registers (each
// element once per workgroup) -> bf16 planes in LDS (row pitch 80 B: conflict-free ds_read_b128 fragments) -> MFMA; two LDS stages,
// global loads two chunks ahead (the MFMA phase of a chunk is only ~400 cycles: one chunk of prefetch does not cover the L2 latency).
// Reports: time per launch, fp32-equivalent TFLOP/s (2*M*N*K / t), and the error of sampled outputs against fp64 for the split
// forms AND for an fp32 FMA chain in K order (what v_mfma_f32_32x32x2_f32 computes), relative to sum_k |a_k b_k|.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bf16_split_probe tools/bf16_split_probe.hip      run: tools/bf16_split_probe [seconds]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int IMG = 128, CH = 128, TAPS = 9, KTOT = TAPS * CH;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 40;                      // bf16 elements per LDS row (32 + 8 pad = 80 B)

// Split two fp32 values into P bf16 pieces each, packed pairwise (piece p of both values in one dword): the plain casts compile
// to v_cvt_pk_bf16_f32 (round-to-nearest-even, one instruction per pair), bf16 -> f32 is a shift / mask.
template <int P>
__device__ __forceinline__ void split2(float x, float y, unsigned (&o)[P]) {
    float rx = x, ry = y;
#pragma unroll
    for (int p = 0; p < P; p++) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 h;
        h[0] = (__bf16)rx; h[1] = (__bf16)ry;
        const unsigned u = __builtin_bit_cast(unsigned, h);
        o[p] = u;
        if (p + 1 < P) {
            rx -= __uint_as_float(u << 16);
            ry -= __uint_as_float(u & 0xFFFF0000u);
        }
    }
}

// P = planes (2: bf16x3, 3: bf16x6)
template <int P>
__global__ __launch_bounds__(512) void gemm_split_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ C, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    // layout: stage s: A planes [P][BM][PITCH], B planes [P][BN][PITCH]
    constexpr int PLANE = BM * PITCH;
    constexpr int STAGE = 2 * P * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * BM;
    // staging role: thread -> row (tid >> 2), 8 consecutive k at 8 * (tid & 3)
    const int srow = tid >> 2, sk = (tid & 3) * 8;
    float4 ra2[2][2], rb2[2][2];          // two register sets: the loads of chunk c+2 are in flight during the whole of iteration c+1
    auto gload = [&](int chunk, float4 (&ra)[2], float4 (&rb)[2]) {
        const int tap = chunk >> 2, c0 = (chunk & 3) * BK + sk;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        long long m = (long long)m0 + srow + dy * IMG + dx;
        m = (m % M + M) % M;
        const float4* pa = reinterpret_cast<const float4*>(X + (size_t)m * CH + c0);
        ra[0] = pa[0]; ra[1] = pa[1];
        const float4* pb = reinterpret_cast<const float4*>(W + (size_t)srow * KTOT + tap * CH + c0);
        rb[0] = pb[0]; rb[1] = pb[1];
    };
    auto sstore = [&](int stage, const float4 (&ra)[2], const float4 (&rb)[2]) {
        unsigned short* base = lds + stage * STAGE;
        const float av[8] = {ra[0].x, ra[0].y, ra[0].z, ra[0].w, ra[1].x, ra[1].y, ra[1].z, ra[1].w};
        const float bv[8] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w, rb[1].x, rb[1].y, rb[1].z, rb[1].w};
        unsigned ap[P][4], bp[P][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned t[P];
            split2<P>(av[2 * j], av[2 * j + 1], t);
#pragma unroll
            for (int p = 0; p < P; p++) ap[p][j] = t[p];
            split2<P>(bv[2 * j], bv[2 * j + 1], t);
#pragma unroll
            for (int p = 0; p < P; p++) bp[p][j] = t[p];
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            *reinterpret_cast<uint4*>(base + p * PLANE + srow * PITCH + sk) = make_uint4(ap[p][0], ap[p][1], ap[p][2], ap[p][3]);
            *reinterpret_cast<uint4*>(base + (P + p) * PLANE + srow * PITCH + sk) = make_uint4(bp[p][0], bp[p][1], bp[p][2], bp[p][3]);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    auto mfmas = [&](int cur) {
        const unsigned short* base = lds + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {      // two k-steps of 16 per chunk
            bf16x8 af[2][P], bf[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
#pragma unroll
                for (int t = 0; t < 2; t++)
                    af[t][p] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + (wm * 64 + t * 32 + l31) * PITCH + ks * 16 + 8 * h);
                bf[p] = *reinterpret_cast<const bf16x8*>(base + (P + p) * PLANE + (wn * 32 + l31) * PITCH + ks * 16 + 8 * h);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                // smallest terms first
                if constexpr (P == 3) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], bf[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], bf[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][2], bf[0], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], bf[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], bf[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], bf[0], acc[t], 0, 0, 0);
            }
        }
    };
    constexpr int CHUNKS = KTOT / BK;        // 36 (even)
    gload(0, ra2[0], rb2[0]);
    sstore(0, ra2[0], rb2[0]);
    gload(1, ra2[1], rb2[1]);
    __syncthreads();
    for (int c = 0; c < CHUNKS; c += 2) {
        // even chunk: computes stage 0; set 0 is free (chunk c was stored from it), set 1 holds chunk c+1
        if (c + 2 < CHUNKS) gload(c + 2, ra2[0], rb2[0]);
        mfmas(0);
        sstore(1, ra2[1], rb2[1]);
        __syncthreads();
        // odd chunk
        if (c + 3 < CHUNKS) gload(c + 3, ra2[1], rb2[1]);
        mfmas(1);
        if (c + 2 < CHUNKS) sstore(0, ra2[0], rb2[0]);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            C[(size_t)(m0 + row) * BN + wn * 32 + l31] = acc[t][r];
        }
}

static float host_bf16(float x, float* rest) {
    unsigned u; memcpy(&u, &x, 4);
    unsigned hbits = ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16) << 16;
    float hi; memcpy(&hi, &hbits, 4);
    *rest = x - hi;
    return hi;
}

template <int P>
static void run(const char* name, const float* dX, const float* dW, float* dC, int M, double seconds, const std::vector<float>& X,
                const std::vector<float>& W) {
    const size_t lds_bytes = 2 * (2 * P * BM * PITCH) * sizeof(unsigned short);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid(M / BM), block(512);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm-up: the clock needs a few hundred ms of load
    double warmed = 0;
    while (warmed < seconds * 0.4) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(gemm_split_kernel<P>, grid, block, lds_bytes, 0, dX, dW, dC, M);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); warmed += ms * 1e-3;
    }
    const int reps = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(gemm_split_kernel<P>, grid, block, lds_bytes, 0, dX, dW, dC, M);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 2.0 * M * BN * KTOT;
    std::vector<float> C((size_t)M * BN);
    CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    // sampled check against fp64; the fp32 FMA chain in K order for comparison
    double worst = 0, rms = 0, worst32 = 0, rms32 = 0;
    int cnt = 0;
    srand(7);
    for (int s = 0; s < 400; s++) {
        const int m = rand() % M, n = rand() % BN;
        double ref = 0, mag = 0;
        float chain = 0.f;
        for (int k = 0; k < KTOT; k++) {
            const int tap = k / CH, c = k % CH;
            long long mm = ((long long)m + (tap / 3 - 1) * IMG + (tap % 3 - 1)) % M; if (mm < 0) mm += M;
            const float a = X[(size_t)mm * CH + c], b = W[(size_t)n * KTOT + k];
            ref += (double)a * b; mag += fabs((double)a * b);
            chain = fmaf(a, b, chain);
        }
        const double e = fabs(C[(size_t)m * BN + n] - ref) / mag, e32 = fabs((double)chain - ref) / mag;
        worst = fmax(worst, e); rms += e * e; worst32 = fmax(worst32, e32); rms32 += e32 * e32; cnt++;
    }
    printf("%-8s %7.1f us per launch = %6.1f fp32-equivalent TFLOP/s | error / sum|a b| over %d sampled outputs: max %.2e rms %.2e   (fp32 FMA chain: max %.2e rms %.2e)\n",
           name, us, flops / us * 1e-6, cnt, worst, sqrt(rms / cnt), worst32, sqrt(rms32 / cnt));
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    const int M = 6 * IMG * IMG;
    std::vector<float> X((size_t)M * CH), W((size_t)BN * KTOT);
    srand(1);
    auto gauss = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v)); };
    for (auto& v : X) v = fmaxf(gauss(), -0.2f * 1.0f) ;       // activation-like: mostly positive (post-lrelu flavour)
    for (auto& v : W) v = gauss() / 34.f;
    float *dX, *dW, *dC;
    CHECK(hipMalloc(&dX, X.size() * 4)); CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * BN * 4));
    CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    printf("GEMM M=%d N=%d K=%d (headline conv shape, 9-tap reuse pattern), fp32 in / fp32 out; reference point: the product's exact-fp32 MFMA kernel runs this shape in ~238 us (122 TFLOP/s)\n", M, BN, KTOT);
    run<2>("bf16x3", dX, dW, dC, M, seconds, X, W);
    run<3>("bf16x6", dX, dW, dC, M, seconds, X, W);
    return 0;
}
