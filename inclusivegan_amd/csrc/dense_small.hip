// Small-batch dense layers on gfx950: y[M,N] = f(x)[M,K] . w  for M <= 64 rows, with the surrounding
// per-element arithmetic of the StyleGAN2 style path folded in (networks_stylegan2.py:41-46,99-107,117:
// the mapping network and, per synthesis layer, the style affine  s = A(w_lat) + b + 1  and the
// demodulation coefficients  d = rsqrt(s^2 . sum_taps(w^2) + 1e-8)  -- 38 small matmuls per generator
// pass at batch 3..24), and their data / weight gradients.
//
// These are not MFMA work: 12 MFLOP against a 1 MB weight matrix.  On the implicit-GEMM path a call
// cost a 32x128-tile kernel cut 8 ways along K plus a reduce kernel, and every surrounding element-wise
// step (bias, +1, square, +eps, rsqrt, the chain-rule factors of the backward) was a launch of its
// own: about 40 launches per modulated layer and direction.  Here one launch streams the weight
// matrix once and applies a prologue to x and an epilogue to y:
//   * a workgroup owns 4 output channels; its 256 lanes are 64 reduction groups x 4 channels, every
//     group takes 4 consecutive k out of each 256 (so the x operand is one ds_read_b128 per row,
//     broadcast to the 4 channel lanes, and the weights are 16 B pieces);
//   * x (up to 32 rows x 512 floats per super-tile, prologue applied) is staged in LDS once per
//     workgroup, rows beyond M zero;
//   * the 64 group partials meet in LDS and are added in fixed order: bit-reproducible, no atomics.
// The weight-gradient kernel is an M-term outer product per element: one thread per 4 output channels.
#include "igan_common.h"
#include <cstdlib>

namespace igan { bool dense_small_fits(long long M, long long K, long long N, long long ldx); }

namespace {

constexpr int DS_COLS = 4;                   // output channels per workgroup
constexpr int DS_GROUPS = 64;                // reduction groups per workgroup
// reduction super-tile staged in LDS at once (floats): 512 for up to 32 rows, 256 for 48 / 64 rows (64 KB of LDS either way)
__host__ __device__ constexpr int ds_ks(int MB) { return MB > 32 ? 256 : 512; }

__device__ __forceinline__ float pro_apply(int pro, float v, float v2, float ps) {
    if (pro == IGAN_DENSE_PRO_SQUARE) return v * v;
    if (pro == IGAN_DENSE_PRO_DEMOD_GRAD) return ps * v * v2 * v2 * v2;    // -1/2 c^2 dd d^3 with ps = -c^2/2
    return v;
}

// The kernel is a latency chain, not a throughput problem (1 MB of weights over >= 128 workgroups): all of a
// super-tile's global loads -- the lane's weights and its share of x -- are issued before anything waits, so
// the chain is  [one round of loads] -> [LDS image of x] -> [FMAs] -> [group reduce].
// Up to IGAN_DENSE_MAX_GROUPS independent problems per launch (blockIdx.y): the 18 style affines / 12 demodulations
// of one generator pass are one launch instead of 30.
struct DenseGroups { igan_dense_params g[IGAN_DENSE_MAX_GROUPS]; };
struct WgradGroups { igan_dense_wgrad_params g[IGAN_DENSE_MAX_GROUPS]; };
struct TapsGroups { igan_taps_params g[IGAN_DENSE_MAX_GROUPS]; };

template <int MB, bool WT>
__global__ __launch_bounds__(256) void dense_small_kernel(DenseGroups G) {
    const igan_dense_params a = G.g[blockIdx.y];
    if ((int)blockIdx.x * DS_COLS >= a.N) return;      // groups differ in width (uniform per workgroup)
    constexpr int DS_KS = ds_ks(MB);
    constexpr int DS_NIT = DS_KS / (4 * DS_GROUPS);   // float4 k-groups per lane per super-tile
    constexpr int RSTR = MB * DS_COLS + 4;   // group stride in the partial-sum image (bank-spread)
    constexpr int XV = MB * (DS_KS / 4) / 256;   // float4 of x per lane per super-tile
    __shared__ __attribute__((aligned(16))) float xs[MB * DS_KS];
    __shared__ float red[DS_GROUPS * RSTR];
    const int M = a.M, K = a.K, N = a.N;
    const int tid = threadIdx.x, c = tid & (DS_COLS - 1), g = tid / DS_COLS;
    const int j = blockIdx.x * DS_COLS + c;
    const bool jok = j < N;
    const float* __restrict__ w = a.w;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // Round 5 (profiles/r05_replay_mismatch.txt: the multi-process mismatch of round 4).  With several PROCESSES on one GPU this kernel returned different
    // results for identical inputs whenever a workgroup of another process's fp16 tile kernel (conv_fwd_planes_kernel<2>) shared its CU; never alone, never
    // beside the bf16 / fp32 tiles, the fp16 weight gradient or itself (tools/coresidency_probe.py: up to 75 % of the calls, errors of 0.1).  What was hit were
    // this kernel's LOADS, in two forms, and both are gone:
    //   * written as `ok ? *p : zero` a guarded load was compiled as a select between two ADDRESSES -- the global one and that of a copy of `zero` in scratch --
    //     and a FLAT load: every instantiation carried 32 bytes of scratch per lane and a flat path into it.  Every guarded load now goes through a buffer
    //     descriptor (an out-of-range offset returns 0): no scratch, no flat instruction.  That alone made the forward form (!WT) reproducible;
    //   * the transposed form's weights, one 16-byte load per lane whose wave request falls into four 256-byte runs 2 KiB apart, still came back wrong; as four
    //     4-byte loads (below) they do not -- 0 of 1200 calls against 700-900 of 1000 (same probe).
    // The aggressor needs nothing but its main loop (barrier, LDS reads, matrix and vector instructions: no LDS-DMA, no store, no atomics: the probe's bisect),
    // so this is an interaction below the programming model, between co-resident waves of different processes; it cannot occur with one process per GPU, which
    // is how the engine is deployed -- the eight-ranks-on-one-GPU runs are a test vehicle.  tests/test_gpu_dist.py::test_eight_rank_replay_stress guards it.
    constexpr unsigned OOB = 0x7FFFFFF0u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(((unsigned)(M - 1) * (unsigned)a.ldx + (unsigned)K) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2 ? a.x2 : a.x), 0, a.x2 ? (int)((unsigned)M * (unsigned)K * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)((unsigned)N * (unsigned)K * 4u), 0x00020000);
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto ld1 = [](__amdgpu_buffer_rsrc_t r, unsigned off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); };
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; m++) acc[m] = 0.f;

    for (int k0 = 0; k0 < K; k0 += DS_KS) {
        // weights of this lane: k = k0 + (it*GROUPS + g)*4 .. +3   (K % 4 == 0: in range together)
        float4 wv[DS_NIT];
#pragma unroll
        for (int it = 0; it < DS_NIT; it++) {
            const int k = k0 + (it * DS_GROUPS + g) * 4;
            const bool wok = jok && k < K;
            if constexpr (WT) {      // Round 5: four 4-byte loads, not one 16-byte load -- see the note at the top of the kernel (the 16-byte form returned wrong values under co-residency; its build switch is gone)
                const unsigned o = wok ? ((unsigned)j * (unsigned)K + (unsigned)k) * 4u : OOB;
                wv[it].x = ld1(rw, o); wv[it].y = ld1(rw, o + 4u); wv[it].z = ld1(rw, o + 8u); wv[it].w = ld1(rw, o + 12u);
            } else {
                const unsigned o = wok ? ((unsigned)k * (unsigned)N + (unsigned)j) * 4u : OOB;      // OOB + 3 N * 4 stays out of range (N * K * 4 < 2^31)
                wv[it].x = ld1(rw, o); wv[it].y = ld1(rw, o + (unsigned)N * 4u);
                wv[it].z = ld1(rw, o + (unsigned)N * 8u); wv[it].w = ld1(rw, o + (unsigned)N * 12u);
            }
        }
        // x super-tile [MB][DS_KS], prologue applied, rows >= M and columns >= K zero
        float4 xv[XV], xu[XV];
#pragma unroll
        for (int q = 0; q < XV; q++) {
            const int v = tid + 256 * q;
            const int m = v / (DS_KS / 4), kv = v - m * (DS_KS / 4);
            const int k = k0 + 4 * kv;
            const bool ok = (m < M) && (k < K);
            xv[q] = ld4(rx, ok ? ((unsigned)m * (unsigned)a.ldx + (unsigned)k) * 4u : OOB);
            xu[q] = ld4(rx2, (ok && a.prologue == IGAN_DENSE_PRO_DEMOD_GRAD) ? ((unsigned)m * (unsigned)K + (unsigned)k) * 4u : OOB);
        }
        if (k0 > 0) __syncthreads();         // the previous super-tile's readers are done with xs
#pragma unroll
        for (int q = 0; q < XV; q++) {
            const int v = tid + 256 * q;
            float4 t = xv[q];
            if (a.prologue != IGAN_DENSE_PRO_NONE) {
                t.x = pro_apply(a.prologue, t.x, xu[q].x, a.pro_scale); t.y = pro_apply(a.prologue, t.y, xu[q].y, a.pro_scale);
                t.z = pro_apply(a.prologue, t.z, xu[q].z, a.pro_scale); t.w = pro_apply(a.prologue, t.w, xu[q].w, a.pro_scale);
            }
            *reinterpret_cast<float4*>(xs + 4 * v) = t;      // v enumerates [m][kv] row-major: offset m*DS_KS + 4*kv
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < DS_NIT; it++) {
            const int i = (it * DS_GROUPS + g) * 4;
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const float4 t = *reinterpret_cast<const float4*>(xs + m * DS_KS + i);
                acc[m] = fmaf(t.x, wv[it].x, fmaf(t.y, wv[it].y, fmaf(t.z, wv[it].z, fmaf(t.w, wv[it].w, acc[m]))));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; m++) red[g * RSTR + m * DS_COLS + c] = acc[m];
    __syncthreads();
    float* ysum = xs;      // [MB][DS_COLS] epilogue values for the column sums (xs is free: all lanes passed the barrier)
    if (tid < MB * DS_COLS) {
        const int o = tid;
        const int m = o / DS_COLS, cc = o - m * DS_COLS;
        float s = 0.f;
#pragma unroll 16
        for (int gg = 0; gg < DS_GROUPS; gg++) s += red[gg * RSTR + o];
        const int jj = blockIdx.x * DS_COLS + cc;
        float v = 0.f;
        if (m < M && jj < N) {
            v = s * a.alpha;
            if (a.epilogue == IGAN_DENSE_EPI_BIAS) v += a.bias_scale * a.bias[jj] + a.add_const;
            else if (a.epilogue == IGAN_DENSE_EPI_RSQRT) v = rsqrtf(v + a.eps);
            else if (a.epilogue == IGAN_DENSE_EPI_STYLE_GRAD) {
                v = 2.0f * a.e2[(size_t)m * N + jj] * v;
                if (a.e1) v += a.e1[(size_t)m * N + jj];
            }
            a.y[(size_t)m * a.ldy + jj] = v;
        }
        if (a.colsum) ysum[o] = v;
    }
    if (a.colsum) {        // colsum[j] = bias_scale * sum_m y[m][j]  (the bias gradient of the style affine)
        __syncthreads();
        if (tid < DS_COLS) {
            const int jj = blockIdx.x * DS_COLS + tid;
            float s = 0.f;
            for (int m = 0; m < MB; m++) s += ysum[m * DS_COLS + tid];
            if (jj < N) a.colsum[jj] = a.bias_scale * s;
        }
    }
}

// dw[k][n] = alpha * sum_m fa(a[m][k]) * fb(b[m][n]); rows in batches of 8 whose loads are all in flight
// together (a serial loop over m is M dependent L2 round trips).
__global__ __launch_bounds__(256) void dense_small_wgrad_kernel(WgradGroups G) {
    const igan_dense_wgrad_params p = G.g[blockIdx.y];
    const int nv = p.N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.K * nv) return;
    const int k = idx / nv, n4 = idx - k * nv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m0 = 0; m0 < p.M; m0 += 8) {
        float xv[8];
        float4 d[8], u[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int m = min(m0 + q, p.M - 1);      // clamped: always a valid address, masked below
            xv[q] = p.a[(size_t)m * p.lda + k];
            d[q] = *reinterpret_cast<const float4*>(p.b + (size_t)m * p.N + 4 * n4);
            if (p.pro_b == IGAN_DENSE_PRO_DEMOD_GRAD) u[q] = *reinterpret_cast<const float4*>(p.b2 + (size_t)m * p.N + 4 * n4);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            float x = (m0 + q < p.M) ? xv[q] : 0.f;
            if (p.pro_a == IGAN_DENSE_PRO_SQUARE) x *= x;
            float4 t = d[q];
            if (p.pro_b == IGAN_DENSE_PRO_DEMOD_GRAD) {
                t.x = p.pro_scale * t.x * u[q].x * u[q].x * u[q].x; t.y = p.pro_scale * t.y * u[q].y * u[q].y * u[q].y;
                t.z = p.pro_scale * t.z * u[q].z * u[q].z * u[q].z; t.w = p.pro_scale * t.w * u[q].w * u[q].w * u[q].w;
            }
            acc.x = fmaf(x, t.x, acc.x); acc.y = fmaf(x, t.y, acc.y);
            acc.z = fmaf(x, t.z, acc.z); acc.w = fmaf(x, t.w, acc.w);
        }
    }
    *reinterpret_cast<float4*>(p.dw + (size_t)k * p.N + 4 * n4) =
        make_float4(acc.x * p.alpha, acc.y * p.alpha, acc.z * p.alpha, acc.w * p.alpha);
}

// out[i] = sum_t w[t][i]^2
__global__ __launch_bounds__(256) void sumsq_taps_kernel(TapsGroups G) {
    const igan_taps_params q = G.g[blockIdx.y];
    const float* __restrict__ w = q.w;
    float* __restrict__ out = q.out;
    const int taps = q.taps, n4 = q.n >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < taps; t++) {
        const float4 v = reinterpret_cast<const float4*>(w)[(size_t)t * n4 + i];
        s.x = fmaf(v.x, v.x, s.x); s.y = fmaf(v.y, v.y, s.y); s.z = fmaf(v.z, v.z, s.z); s.w = fmaf(v.w, v.w, s.w);
    }
    reinterpret_cast<float4*>(out)[i] = s;
}

// out[t][i] = scale * w[t][i] * v[i]
__global__ __launch_bounds__(256) void bcast_mul_taps_kernel(TapsGroups G) {
    const igan_taps_params q = G.g[blockIdx.y];
    const float* __restrict__ w = q.w;
    const float* __restrict__ v = q.v;
    float* __restrict__ out = q.out;
    const int taps = q.taps, n4 = q.n >> 2;
    const float scale = q.scale;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 m = reinterpret_cast<const float4*>(v)[i];
    for (int t = 0; t < taps; t++) {
        const float4 x = reinterpret_cast<const float4*>(w)[(size_t)t * n4 + i];
        reinterpret_cast<float4*>(out)[(size_t)t * n4 + i] = make_float4(scale * x.x * m.x, scale * x.y * m.y, scale * x.z * m.z, scale * x.w * m.w);
    }
}

template <int MB>
void launch_dense(hipStream_t stream, const DenseGroups& G, int count, int maxN, bool wt) {
    const dim3 grid(igan::ceil_div(maxN, DS_COLS), count);
    if (wt) hipLaunchKernelGGL((dense_small_kernel<MB, true>), grid, dim3(256), 0, stream, G);
    else hipLaunchKernelGGL((dense_small_kernel<MB, false>), grid, dim3(256), 0, stream, G);
}

int dense_check(const igan_dense_params* p) {
    IGAN_REQUIRE(p->x && p->w && p->y, "dense_small: null buffer");
    IGAN_REQUIRE(p->M >= 1 && p->M <= IGAN_DENSE_MAX_ROWS, "dense_small: 1 <= M <= %d rows (use igan_conv2d for larger batches)", IGAN_DENSE_MAX_ROWS);
    IGAN_REQUIRE(p->K >= 4 && p->K % 4 == 0 && p->N >= 1, "dense_small: K must be a positive multiple of 4, N positive");
    IGAN_REQUIRE(p->ldx >= p->K && p->ldx % 4 == 0 && p->ldy >= p->N, "dense_small: bad row strides");
    IGAN_REQUIRE(igan::dense_small_fits(p->M, p->K, p->N, p->ldx), "dense_small: x, x2 and w must each stay below 2 GiB (32-bit buffer offsets; use igan_conv2d for larger operands)");
    IGAN_REQUIRE((((uintptr_t)p->x) & 15) == 0 && (!p->w_transposed || (((uintptr_t)p->w) & 15) == 0), "dense_small: x (and a transposed w) must be 16-byte aligned");
    IGAN_REQUIRE(p->prologue >= IGAN_DENSE_PRO_NONE && p->prologue <= IGAN_DENSE_PRO_DEMOD_GRAD, "dense_small: unknown prologue");
    IGAN_REQUIRE(p->epilogue >= IGAN_DENSE_EPI_SCALE && p->epilogue <= IGAN_DENSE_EPI_STYLE_GRAD, "dense_small: unknown epilogue");
    IGAN_REQUIRE(p->prologue != IGAN_DENSE_PRO_DEMOD_GRAD || (p->x2 && (((uintptr_t)p->x2) & 15) == 0), "dense_small: demod-gradient prologue needs an aligned x2");
    IGAN_REQUIRE(p->epilogue != IGAN_DENSE_EPI_BIAS || p->bias, "dense_small: bias epilogue needs a bias");
    IGAN_REQUIRE(p->epilogue != IGAN_DENSE_EPI_STYLE_GRAD || p->e2, "dense_small: style-gradient epilogue needs e2");
    return IGAN_OK;
}

int wgrad_check(const igan_dense_wgrad_params* p) {
    IGAN_REQUIRE(p->a && p->b && p->dw, "dense_small_wgrad: null buffer");
    IGAN_REQUIRE(p->M >= 1 && p->M <= IGAN_DENSE_MAX_ROWS, "dense_small_wgrad: 1 <= M <= %d rows", IGAN_DENSE_MAX_ROWS);
    IGAN_REQUIRE(p->K >= 1 && p->N >= 4 && p->N % 4 == 0 && p->lda >= p->K, "dense_small_wgrad: N must be a positive multiple of 4");
    IGAN_REQUIRE(((((uintptr_t)p->b) | ((uintptr_t)p->dw)) & 15) == 0, "dense_small_wgrad: b and dw must be 16-byte aligned");
    IGAN_REQUIRE(p->pro_a == IGAN_DENSE_PRO_NONE || p->pro_a == IGAN_DENSE_PRO_SQUARE, "dense_small_wgrad: unknown prologue for a");
    IGAN_REQUIRE(p->pro_b == IGAN_DENSE_PRO_NONE || (p->pro_b == IGAN_DENSE_PRO_DEMOD_GRAD && p->b2 && (((uintptr_t)p->b2) & 15) == 0), "dense_small_wgrad: bad prologue for b");
    IGAN_REQUIRE((long long)p->K * p->N <= INT32_MAX, "dense_small_wgrad: too large");
    return IGAN_OK;
}

int taps_check(const igan_taps_params* q, bool need_v, const char* who) {
    IGAN_REQUIRE(q->w && q->out && (!need_v || q->v) && q->taps >= 1 && q->n >= 4 && q->n % 4 == 0, "%s: n must be a positive multiple of 4", who);
    IGAN_REQUIRE(((((uintptr_t)q->w) | ((uintptr_t)q->v) | ((uintptr_t)q->out)) & 15) == 0, "%s: buffers must be 16-byte aligned", who);
    return IGAN_OK;
}

}  // namespace

namespace igan {

// The kernel reads x, x2 and w through buffer descriptors with 32-bit byte offsets and marks a guarded-off lane by the offset 0x7FFFFFF0 (which must lie
// OUTSIDE every descriptor's range, also after the + 3 N * 4 of the forward form's four row loads): every operand must stay below that many bytes.
// (DCI's re-rank reaches this path with N = 8192 candidates x an unprojected dim of 196 608 = 6 GiB; such a call takes the MFMA tiles.)
bool dense_small_fits(long long M, long long K, long long N, long long ldx) {
    const long long lim = 0x7FFFFFF0LL;
    return N * K * 4 < lim && ((M - 1) * ldx + K) * 4 < lim && M * K * 4 < lim && lim + 3 * N * 4 < 0xFFFFFFFFLL;
}

bool dense_small_ok(int M, int K, int N, const void* x, const void* w, bool wt) {
    // w[k][n] is streamed in 16 B pieces per k row: fine while the matrix is small and L2-resident, 4x read
    // amplification from HBM for a long reduction axis (D's 8192 -> 512 layer): that one goes to the MFMA tiles.
    static const int maxk = getenv("IGAN_DENSE_MAXK") ? atoi(getenv("IGAN_DENSE_MAXK")) : 2048;
    if (!wt && K > maxk) return false;
    if (!dense_small_fits(M, K, N, K)) return false;
    return M >= 1 && M <= IGAN_DENSE_MAX_ROWS && (K % 4) == 0 && (((uintptr_t)x) & 15) == 0 && (!wt || (((uintptr_t)w) & 15) == 0);
}

int dense_small_rows(int M) { return M <= 8 ? 8 : (M <= 16 ? 16 : (M <= 24 ? 24 : (M <= 32 ? 32 : (M <= 48 ? 48 : 64)))); }

// all groups of one launch share the weight layout (w_transposed) and the row-count bucket
void dense_small_launch_groups(hipStream_t stream, const igan_dense_params* groups, int count) {
    DenseGroups G;
    int maxM = 1, maxN = 1;
    for (int i = 0; i < count; i++) { G.g[i] = groups[i]; maxM = std::max(maxM, groups[i].M); maxN = std::max(maxN, groups[i].N); }
    const int mb = dense_small_rows(maxM);
    const bool wt = groups[0].w_transposed != 0;
    if (mb == 8) launch_dense<8>(stream, G, count, maxN, wt);
    else if (mb == 16) launch_dense<16>(stream, G, count, maxN, wt);
    else if (mb == 24) launch_dense<24>(stream, G, count, maxN, wt);
    else if (mb == 32) launch_dense<32>(stream, G, count, maxN, wt);
    else if (mb == 48) launch_dense<48>(stream, G, count, maxN, wt);
    else launch_dense<64>(stream, G, count, maxN, wt);
}

void dense_small_launch(hipStream_t stream, const igan_dense_params& a) { dense_small_launch_groups(stream, &a, 1); }

void dense_small(hipStream_t stream, const float* x, const float* w, float* y, int M, int K, int N, bool wt, float alpha) {
    igan_dense_params a = {};
    a.x = x; a.ldx = K; a.w = w; a.y = y; a.ldy = N;
    a.M = M; a.K = K; a.N = N; a.w_transposed = wt ? 1 : 0;
    a.prologue = IGAN_DENSE_PRO_NONE; a.epilogue = IGAN_DENSE_EPI_SCALE; a.alpha = alpha;
    dense_small_launch(stream, a);
}

bool dense_small_wgrad_ok(int M, int N, const void* dy, const void* dw) {
    return M >= 1 && M <= IGAN_DENSE_MAX_ROWS && (N % 4) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dw)) & 15) == 0;
}

void dense_small_wgrad_groups(hipStream_t stream, const igan_dense_wgrad_params* groups, int count) {
    WgradGroups G;
    int maxTotal = 1;
    for (int i = 0; i < count; i++) { G.g[i] = groups[i]; maxTotal = std::max(maxTotal, groups[i].K * (groups[i].N >> 2)); }
    hipLaunchKernelGGL(dense_small_wgrad_kernel, dim3(ceil_div(maxTotal, 256), count), dim3(256), 0, stream, G);
}

void dense_small_wgrad(hipStream_t stream, const float* x, const float* dy, float* dw, int M, int K, int N, float alpha) {
    igan_dense_wgrad_params p = {};
    p.a = x; p.lda = K; p.b = dy; p.dw = dw; p.M = M; p.K = K; p.N = N; p.alpha = alpha;
    dense_small_wgrad_groups(stream, &p, 1);
}

}  // namespace igan

extern "C" int igan_dense_small(igan_stream_t stream_, const igan_dense_params* p) {
    return igan_dense_small_grouped(stream_, p, 1);
}

extern "C" int igan_dense_small_grouped(igan_stream_t stream_, const igan_dense_params* groups, int count) {
    using namespace igan;
    IGAN_REQUIRE(groups && count >= 1 && count <= IGAN_DENSE_MAX_GROUPS, "dense_small: 1 <= count <= %d groups", IGAN_DENSE_MAX_GROUPS);
    for (int i = 0; i < count; i++) {
        if (int rc = dense_check(&groups[i])) return rc;
        IGAN_REQUIRE((groups[i].w_transposed != 0) == (groups[0].w_transposed != 0), "dense_small: the groups of one launch share w_transposed");
    }
    dense_small_launch_groups((hipStream_t)stream_, groups, count);
    IGAN_LAUNCH_CHECK("dense_small launch");
    return IGAN_OK;
}

extern "C" int igan_dense_small_wgrad(igan_stream_t stream_, const igan_dense_wgrad_params* p) {
    return igan_dense_small_wgrad_grouped(stream_, p, 1);
}

extern "C" int igan_dense_small_wgrad_grouped(igan_stream_t stream_, const igan_dense_wgrad_params* groups, int count) {
    using namespace igan;
    IGAN_REQUIRE(groups && count >= 1 && count <= IGAN_DENSE_MAX_GROUPS, "dense_small_wgrad: 1 <= count <= %d groups", IGAN_DENSE_MAX_GROUPS);
    for (int i = 0; i < count; i++)
        if (int rc = wgrad_check(&groups[i])) return rc;
    dense_small_wgrad_groups((hipStream_t)stream_, groups, count);
    IGAN_LAUNCH_CHECK("dense_small_wgrad launch");
    return IGAN_OK;
}

static int taps_launch(igan_stream_t stream_, const igan_taps_params* groups, int count, bool mul, const char* who) {
    using namespace igan;
    IGAN_REQUIRE(groups && count >= 1 && count <= IGAN_DENSE_MAX_GROUPS, "%s: 1 <= count <= %d groups", who, IGAN_DENSE_MAX_GROUPS);
    TapsGroups G;
    int maxn4 = 1;
    for (int i = 0; i < count; i++) {
        if (int rc = taps_check(&groups[i], mul, who)) return rc;
        G.g[i] = groups[i];
        maxn4 = std::max(maxn4, groups[i].n / 4);
    }
    const dim3 grid(ceil_div(maxn4, 256), count);
    if (mul) hipLaunchKernelGGL(bcast_mul_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream_, G);
    else hipLaunchKernelGGL(sumsq_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream_, G);
    IGAN_LAUNCH_CHECK(who);
    return IGAN_OK;
}

extern "C" int igan_sumsq_taps(igan_stream_t stream_, const float* w, float* out, int taps, int n) {
    igan_taps_params q = {w, nullptr, out, taps, n, 1.0f};
    return taps_launch(stream_, &q, 1, false, "sumsq_taps");
}

extern "C" int igan_bcast_mul_taps(igan_stream_t stream_, const float* w, const float* v, float* out, int taps, int n, float scale) {
    igan_taps_params q = {w, v, out, taps, n, scale};
    return taps_launch(stream_, &q, 1, true, "bcast_mul_taps");
}

extern "C" int igan_sumsq_taps_grouped(igan_stream_t stream_, const igan_taps_params* groups, int count) {
    return taps_launch(stream_, groups, count, false, "sumsq_taps");
}

extern "C" int igan_bcast_mul_taps_grouped(igan_stream_t stream_, const igan_taps_params* groups, int count) {
    return taps_launch(stream_, groups, count, true, "bcast_mul_taps");
}
