// Sustained rate of the two exact-f32 matrix instructions of gfx950 (and of the fp16 / bf16 32x32x16 instructions of the piece products) in bare register loops (no memory traffic
// inside the loop), on random operands, with the in-kernel clock: what the matrix pipe delivers when nothing
// else competes for power.  The conv kernels are priced against this ceiling.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate
//   usage: tools/mfma_rate [seconds per variant]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: v_mfma_f32_32x32x2_f32, 4 independent accumulators (64 regs); MODE 1: v_mfma_f32_16x16x4_f32, 8 accumulators (32 regs);
// MODE 2 / 3: v_mfma_f32_32x32x16_f16 / _bf16 (the piece products of the 3x3 convolutions), 4 accumulators, operands = random values rounded to the type
template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ stamps) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = in[(tid * 16 + i) & 0xFFFFF]; b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF]; }
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sum = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + t) & 7], b[j], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) sum += acc[t][r];
    } else if constexpr (MODE >= 2) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        f16x8 ha[4], hb[2];
        bf16x8 ba[4], bb[2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int k = 0; k < 8; k++) { const float v = in[(tid * 64 + 8 * i + k) & 0xFFFFF]; ha[i][k] = (_Float16)v; ba[i][k] = (__bf16)v; }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int k = 0; k < 8; k++) { const float v = in[(tid * 64 + 32 + 8 * i + k) & 0xFFFFF]; hb[i][k] = (_Float16)v; bb[i][k] = (__bf16)v; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if constexpr (MODE == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[(j + t) & 3], hb[j & 1], acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[(j + t) & 3], bb[j & 1], acc[t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) sum += acc[t][r];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[t][r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int t = 0; t < 8; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + t) & 7], b[j], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) sum += acc[t][r];
    }
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[tid] = sum;
}

template <int MODE>
void run(const char* name, int blocks, double seconds, const float* in, float* out, unsigned long long* stamps) {
    // per wave and iteration: MODE 0: 32 MFMAs x 4096 FLOP; MODE 1: 64 MFMAs x 2048 FLOP
    const double flop_iter = (MODE >= 2 ? 8.0 : 1.0) * 131072.0 * 4 * blocks;         // MODE 2 / 3: 32 MFMAs x 32768 FLOP
    const int iters = MODE >= 2 ? 10000 : 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
    CK(hipDeviceSynchronize());
    double first = 0, last = 0, clk_last = 0;
    double elapsed = 0;
    int n = 0;
    std::vector<unsigned long long> h(2 * blocks);
    while (elapsed < seconds) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        elapsed += ms * 1e-3;
        const double tf = flop_iter * iters / (ms * 1e-3) / 1e12;
        if (n == 0) first = tf;
        last = tf;
        n++;
    }
    CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int i = 0; i < blocks; i++) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);   // MHz (memrealtime = 100 MHz)
    std::sort(clk.begin(), clk.end());
    clk_last = clk[clk.size() / 2];
    printf("%-34s blocks %4d  first launch %6.1f TFLOP/s  after %.1f s %6.1f TFLOP/s  in-kernel clock %.0f MHz (median WG)\n",
           name, blocks, first, elapsed, last, clk_last);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    float *in, *out;
    unsigned long long* stamps;
    CK(hipMalloc(&in, (1 << 20) * sizeof(float)));
    CK(hipMalloc(&out, 4096 * 256 * sizeof(float)));
    CK(hipMalloc(&stamps, 2 * 4096 * sizeof(unsigned long long)));
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    run<0>("mfma_f32_32x32x2  1 wave/SIMD", 256, seconds, in, out, stamps);
    run<0>("mfma_f32_32x32x2  2 waves/SIMD", 512, seconds, in, out, stamps);
    run<1>("mfma_f32_16x16x4  1 wave/SIMD", 256, seconds, in, out, stamps);
    run<1>("mfma_f32_16x16x4  2 waves/SIMD", 512, seconds, in, out, stamps);
    run<2>("mfma_f32_32x32x16_f16  1 wave/SIMD", 256, seconds, in, out, stamps);
    run<2>("mfma_f32_32x32x16_f16  2 waves/SIMD", 512, seconds, in, out, stamps);
    run<3>("mfma_f32_32x32x16_bf16 1 wave/SIMD", 256, seconds, in, out, stamps);
    run<3>("mfma_f32_32x32x16_bf16 2 waves/SIMD", 512, seconds, in, out, stamps);
    return 0;
}
