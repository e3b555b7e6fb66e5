// Which bf16 matrix-instruction shape should the piece kernels use?  The step of conv_fwd_planes_kernel (csrc/conv2d_mfma.hip) without its loads:
// a 128x128 tile per 8-wave workgroup, two workgroups per CU (72 KiB of LDS each), per 16-deep step a barrier, the fragment reads of the wave's
// 64x32 sub-tile from a [piece][row][2 x 16 B] image in LDS, the six piece products from an exact zero (largest first), and the fold of the
// step's sum into the running fp32 sum -- in two forms:
//   MODE 0  v_mfma_f32_32x32x16_bf16 (what the kernels use): 9 ds_read_b128, 12 MFMAs of 32 cycles, 32 adds per wave and step
//   MODE 1  v_mfma_f32_16x16x32_bf16 with PAIRED pieces: the instruction's K = 32 is [16 channels of piece i | 16 channels of piece j], so
//           [a0|a1].[b0|b0] = a0 b0 + a1 b0,  [a0|a1].[b1|b1] = a0 b1 + a1 b1,  [a0|a2].[b2|b0] = a0 b2 + a2 b0   (all six products, largest
//           first): 14 ds_read_b128 (8 for the four 16-row A tiles, 6 for the two 16-column B tiles), 24 MFMAs of 16 cycles, 32 adds
// MI355X_MICROARCH.md measures the 16x16x32 form at 1.12-1.15x the FLOP/s of the 32x32x16 form in bare loops on random data (it holds a higher
// clock under the power limit); this probe asks whether that survives the piece kernel's own mix (more LDS reads per step in MODE 1).
// Operands are random bf16 (LDS filled once); the data never changes, so this measures rate and clock, not arithmetic.
//   build: hipcc -O3 --offload-arch=gfx950 tools/piece_shape_probe.hip -o tools/piece_shape_probe
//   usage: tools/piece_shape_probe [seconds per variant]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int P_IMG = 3 * 128 * 32;        // [3 pieces][128 rows][32 B]
constexpr int P_STAGE = 2 * P_IMG;         // A + B
constexpr int NSTAGE = 3;

// address of the 16-byte half `half` of row r in piece q of an image (the kernel's swizzle)
__device__ __forceinline__ int img_off(int q, int r, int half) { return q * 4096 + (2 * r + (half ^ ((r >> 3) & 1))) * 16; }

template <int MODE>
__global__ __launch_bounds__(512, 4) void step_loop(const unsigned* __restrict__ seed, float* __restrict__ out, int steps,
                                                     unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * P_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // fill LDS with random bf16 bit patterns of moderate exponent (sign random, exponent 120..127, mantissa random)
    for (int i = tid; i < NSTAGE * P_STAGE / 4; i += 512) {
        unsigned x = seed[(blockIdx.x * 7919 + i) & 0xFFFFF];
        unsigned lo = (x & 0x807Fu) | ((120u + ((x >> 8) & 7u)) << 7), hi = ((x >> 16) & 0x807Fu) | ((120u + ((x >> 24) & 7u)) << 7);
        if constexpr (MODE >= 2) {      // fp16 patterns: sign and mantissa random, exponent 12..15
            lo = (x & 0x83FFu) | ((12u + ((x >> 10) & 3u)) << 10); hi = ((x >> 16) & 0x83FFu) | ((12u + ((x >> 26) & 3u)) << 10);
        }
        reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
    }
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sum = 0.f;
    int st = 0;
    if constexpr (MODE == 0) {
        const int l31 = lane & 31, h = lane >> 5;
        int fa[2];
        for (int tm = 0; tm < 2; tm++) fa[tm] = img_off(0, wm * 64 + tm * 32 + l31, h);
        const int fb = P_IMG + img_off(0, wn * 32 + l31, h);
        f32x16 acc[2];
        for (int tm = 0; tm < 2; tm++) for (int r = 0; r < 16; r++) acc[tm][r] = 0.f;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < steps; c++) {
            __syncthreads();
            const unsigned char* S = smem + st * P_STAGE;
            bf16x8 af[2][3], bfr[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                bfr[q] = *reinterpret_cast<const bf16x8*>(S + fb + q * 4096);
#pragma unroll
                for (int tm = 0; tm < 2; tm++) af[tm][q] = *reinterpret_cast<const bf16x8*>(S + fa[tm] + q * 4096);
            }
            f32x16 t[2];
#pragma unroll
            for (int tm = 0; tm < 2; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[0], zero, 0, 0, 0);
#pragma unroll
            for (int o = 1; o < 3; o++)
#pragma unroll
                for (int i = 0; i <= o; i++)
#pragma unroll
                    for (int tm = 0; tm < 2; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][i], bfr[o - i], t[tm], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < 2; tm++) acc[tm] += t[tm];
            st = (st + 1 == NSTAGE) ? 0 : st + 1;
        }
        for (int tm = 0; tm < 2; tm++) for (int r = 0; r < 16; r++) sum += acc[tm][r];
    } else if constexpr (MODE == 2 || MODE == 3) {
        // TWO fp16 pieces per operand (a = a0 + 2^-11 a1', exact to 2^-24 inside fp16's range after a per-tensor power-of-two scale): three products per
        // step, the main one from an exact zero and folded by the vector ALU (as above), the two cross products chained in their own accumulator
        // (they carry 2^-11 of the result: the instruction's own rounding is far below fp32's there).  LDS image [2 pieces][128 rows][32 B].
        // MODE 3: two 16-deep sub-steps per barrier.
        constexpr int SUB = MODE == 3 ? 2 : 1;
        constexpr int H_IMG = 2 * 128 * 32, H_STAGE = 2 * H_IMG;
        const int l31 = lane & 31, h = lane >> 5;
        int fa[2];
        for (int tm = 0; tm < 2; tm++) fa[tm] = img_off(0, wm * 64 + tm * 32 + l31, h);
        const int fb = H_IMG + img_off(0, wn * 32 + l31, h);
        f32x16 acc[2], u[2];
        for (int tm = 0; tm < 2; tm++) for (int r = 0; r < 16; r++) { acc[tm][r] = 0.f; u[tm][r] = 0.f; }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < steps; c += SUB) {
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < SUB; sub++) {
                const unsigned char* S = smem + st * H_STAGE;
                f16x8 a0[2], a1[2], b0, b1;
                b0 = *reinterpret_cast<const f16x8*>(S + fb);
                b1 = *reinterpret_cast<const f16x8*>(S + fb + 4096);
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    a0[tm] = *reinterpret_cast<const f16x8*>(S + fa[tm]);
                    a1[tm] = *reinterpret_cast<const f16x8*>(S + fa[tm] + 4096);
                }
                f32x16 t[2];
#pragma unroll
                for (int tm = 0; tm < 2; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b0, zero, 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < 2; tm++) u[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b1, u[tm], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < 2; tm++) u[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[tm], b0, u[tm], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < 2; tm++) acc[tm] += t[tm];
                st = (st + 1 == 4) ? 0 : st + 1;
            }
        }
        for (int tm = 0; tm < 2; tm++) for (int r = 0; r < 16; r++) sum += acc[tm][r] + u[tm][r] * (1.0f / 2048.0f);
    } else {
        // 16x16x32: lane l holds row / column l % 16 and the 8 k values of k-group l / 16 (groups 0, 1 = first 16 channels, 2, 3 = second 16)
        const int l15 = lane & 15, g = lane >> 4, half = g & 1, second = g >> 1;
        // A': [a0|a1] and [a0|a2]; B': [b0|b0], [b1|b1], [b2|b0]
        int fa01[4], fa02[4], fb00[2], fb11[2], fb20[2];
        for (int rt = 0; rt < 4; rt++) {
            const int r = wm * 64 + rt * 16 + l15;
            fa01[rt] = img_off(second ? 1 : 0, r, half);
            fa02[rt] = img_off(second ? 2 : 0, r, half);
        }
        for (int ct = 0; ct < 2; ct++) {
            const int r = wn * 32 + ct * 16 + l15;
            fb00[ct] = P_IMG + img_off(0, r, half);
            fb11[ct] = P_IMG + img_off(1, r, half);
            fb20[ct] = P_IMG + img_off(second ? 0 : 2, r, half);
        }
        f32x4 acc[4][2];
        for (int rt = 0; rt < 4; rt++) for (int ct = 0; ct < 2; ct++) for (int r = 0; r < 4; r++) acc[rt][ct][r] = 0.f;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < steps; c++) {
            __syncthreads();
            const unsigned char* S = smem + st * P_STAGE;
            bf16x8 a01[4], a02[4], b00[2], b11[2], b20[2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                b00[ct] = *reinterpret_cast<const bf16x8*>(S + fb00[ct]);
                b11[ct] = *reinterpret_cast<const bf16x8*>(S + fb11[ct]);
                b20[ct] = *reinterpret_cast<const bf16x8*>(S + fb20[ct]);
            }
#pragma unroll
            for (int rt = 0; rt < 4; rt++) {
                a01[rt] = *reinterpret_cast<const bf16x8*>(S + fa01[rt]);
                a02[rt] = *reinterpret_cast<const bf16x8*>(S + fa02[rt]);
            }
            f32x4 t[4][2];
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) t[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01[rt], b00[ct], zero, 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) t[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01[rt], b11[ct], t[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) t[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a02[rt], b20[ct], t[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) acc[rt][ct] += t[rt][ct];
            st = (st + 1 == NSTAGE) ? 0 : st + 1;
        }
        for (int rt = 0; rt < 4; rt++) for (int ct = 0; ct < 2; ct++) for (int r = 0; r < 4; r++) sum += acc[rt][ct][r];
    }
    if (tid == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * 512 + tid] = sum;
}

template <int MODE>
static void run(const char* name, const unsigned* seed, float* out, unsigned long long* stamps, double seconds) {
    const int blocks = 512;                 // two workgroups per CU on 256 CUs
    const int steps = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best_ms = 1e30, clk = 0;
    double t_total = 0;
    int reps = 0;
    while (t_total < seconds * 1e3 || reps < 3) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(step_loop<MODE>, dim3(blocks), dim3(512), 0, 0, seed, out, steps, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t_total += ms; reps++;
        if (t_total > seconds * 500) {      // second half: sustained clocks
            best_ms = std::min(best_ms, (double)ms);
            std::vector<unsigned long long> h(2 * blocks);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> r;
            for (int b = 0; b < blocks; b++) if (h[2 * b + 1]) r.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100e6);
            std::sort(r.begin(), r.end());
            clk = r[r.size() / 2];
        }
    }
    // one workgroup step = 128 x 128 x 16 fp32 products = 2 * 128 * 128 * 16 fp32-equivalent FLOP
    const double flop = 2.0 * 128 * 128 * 16 * (double)steps * blocks;
    printf("%-58s %8.3f ms  %7.1f fp32-equivalent TFLOP/s  (bf16 rate x 6 = %6.1f TFLOP/s)  in-kernel clock %.2f GHz  %.0f cycles per step and workgroup pair\n",
           name, best_ms, flop / best_ms / 1e9, 6 * flop / best_ms / 1e9, clk / 1e9, clk * best_ms * 1e-3 / steps);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    unsigned* seed; float* out; unsigned long long* stamps;
    std::vector<unsigned> h(1 << 20);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CK(hipMalloc(&seed, h.size() * 4)); CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 512 * 512 * 4)); CK(hipMalloc(&stamps, 1024 * 8));
    for (int round = 0; round < 2; round++) {
        run<0>("32x32x16: 9 reads, 12 MFMAs, 32 adds per wave-step", seed, out, stamps, seconds);
        run<1>("16x16x32 paired pieces: 14 reads, 24 MFMAs, 32 adds", seed, out, stamps, seconds);
        run<2>("fp16 x 2 pieces, 32x32x16: 6 reads, 6 MFMAs, 32 adds", seed, out, stamps, seconds);
        run<3>("fp16 x 2 pieces, two sub-steps per barrier", seed, out, stamps, seconds);
    }
    return 0;
}
