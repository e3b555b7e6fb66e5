// Exact streaming 1-nearest-neighbour for the IMLE assignment step on gfx950.
//
// Behavioural contract: the reference builds a Prioritized-DCI index over the
// generated candidates and asks for ONE neighbour per real image
// (training/training_loop.py:367-368,398; dci_code/src/dci.c:108-337,788-828),
// with Euclidean distances (dci_code/src/util.c:62-69).  DCI is approximate and
// seeded from time(NULL) (dci.c:77,860); what the training loop consumes is the
// arg-min candidate per real and its distance.
// MI355X design: no index at all.  |q - c|^2 = |q|^2 + |c|^2 - 2 q.c, so a batch
// of candidates is one [nq x dim] x [dim x nc] product on the exact-fp32 MFMA
// (the same implicit-GEMM kernel as conv2d, called as a 1x1 conv with the
// candidate matrix as a transposed weight -- both operands are read in their
// natural row-major [rows][dim] layout), followed by a per-query wavefront
// min-reduction that folds the batch into a running packed (distance,index)
// minimum.  Candidate batches can therefore be generated, consumed and discarded:
// the reference's 118 GB fp64 host array (training_loop.py:358) never exists.
#include "igan_common.h"

namespace {

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* a, float* out, int rows, int dim) {
    // one wavefront per row, fp64 accumulation
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* p = a + (size_t)wave * dim;
    double s = 0.0;
    for (int i = lane; i < dim; i += 64) {
        const double v = (double)p[i];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[wave] = (float)s;
}

// Relative half-width of the interval the fp32 cancellation form |q|^2 + |c|^2 - 2 q.c is trusted to: the dot product is an
// fp32 FMA chain of `dim` terms (error ~ 0.3 * 2^-24 * sqrt(dim) * |q.c| statistically; measured 3.5e-7 * sum|a b| at dim 4096),
// the norms are fp64 sums rounded to fp32.  2^-22 * sqrt(dim) (5.3e-5 at dim 49 152) leaves an order of magnitude of margin;
// a wider interval only costs more exact evaluations, never a wrong answer.
__device__ __forceinline__ double nn1_tol(int dim) { return fmax(2.384185791015625e-07 * sqrt((double)dim), 1e-6); }

// Squared distance of two rows as a direct difference with fp64 accumulation (compute_dist, dci_code/src/util.c:62-69,
// before its sqrt), by one whole wavefront; the result is in every lane.  Fixed summation order: deterministic.
__device__ __forceinline__ double wave_sqdist(const float* __restrict__ a, const float* __restrict__ b, int dim, int lane) {
    double s0 = 0.0, s1 = 0.0;
    int i = lane * 4;
    if ((dim & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0)) {
        for (; i + 256 < dim; i += 512) {      // two 16 B loads per operand in flight
            const float4 x0 = *reinterpret_cast<const float4*>(a + i), y0 = *reinterpret_cast<const float4*>(b + i);
            const float4 x1 = *reinterpret_cast<const float4*>(a + i + 256), y1 = *reinterpret_cast<const float4*>(b + i + 256);
            double d;
            d = (double)x0.x - (double)y0.x; s0 += d * d; d = (double)x0.y - (double)y0.y; s0 += d * d;
            d = (double)x0.z - (double)y0.z; s0 += d * d; d = (double)x0.w - (double)y0.w; s0 += d * d;
            d = (double)x1.x - (double)y1.x; s1 += d * d; d = (double)x1.y - (double)y1.y; s1 += d * d;
            d = (double)x1.z - (double)y1.z; s1 += d * d; d = (double)x1.w - (double)y1.w; s1 += d * d;
        }
        for (; i < dim; i += 256) {
            const float4 x0 = *reinterpret_cast<const float4*>(a + i), y0 = *reinterpret_cast<const float4*>(b + i);
            double d;
            d = (double)x0.x - (double)y0.x; s0 += d * d; d = (double)x0.y - (double)y0.y; s0 += d * d;
            d = (double)x0.z - (double)y0.z; s0 += d * d; d = (double)x0.w - (double)y0.w; s0 += d * d;
        }
    } else {
        for (int j = lane; j < dim; j += 64) {
            const double d = (double)a[j] - (double)b[j];
            s0 += d * d;
        }
    }
    double s = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// One wavefront per query.  The cancellation form only SCREENS: a candidate stays a contender while the interval
// [a - t, a + t] around its approximate squared distance a reaches below the best upper bound known (the running exact
// minimum, or the smallest a + t of this batch); every contender is then measured exactly (fp64 direct difference) and the
// choice is made on (exact distance, index) -- lexicographic, ties to the lower index, exactly what an fp64 brute-force
// search returns (oracle/nn.py; the reference's dci_query(num_neighbours=1) approximates it).  A non-finite dot product or
// distance compares false everywhere: such a candidate can never win.
__global__ __launch_bounds__(256) void nn1_fold_kernel(const float* __restrict__ dots, const float* __restrict__ qnorm,
                                                       const float* __restrict__ cnorm, const float* __restrict__ query,
                                                       const float* __restrict__ cand, double* __restrict__ best_d2,
                                                       int* __restrict__ best_idx, int nq, int nc, int dim, int idx_base) {
    const int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    const double qn = (double)qnorm[q];
    const double tol = nn1_tol(dim);
    double bd = best_d2[q];
    int bi = best_idx[q];
    // pass 1: smallest upper bound of this batch
    double u = bd;
    for (int c = lane; c < nc; c += 64) {
        const double cn = (double)cnorm[c];
        const double a = qn + cn - 2.0 * (double)dots[(size_t)q * nc + c];
        const double hi = a + tol * (qn + cn);
        u = (hi < u) ? hi : u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(u, off, 64);
        u = (o < u) ? o : u;
    }
    // pass 2: exact evaluation of the contenders, in index order
    const float* qrow = query + (size_t)q * dim;
    for (int c0 = 0; c0 < nc; c0 += 64) {
        const int c = c0 + lane;
        bool contender = false;
        if (c < nc) {
            const double cn = (double)cnorm[c];
            const double a = qn + cn - 2.0 * (double)dots[(size_t)q * nc + c];
            contender = (a - tol * (qn + cn)) <= u;
        }
        unsigned long long mask = __ballot(contender);
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const double e = wave_sqdist(qrow, cand + (size_t)(c0 + j) * dim, dim, lane);
            const int gi = idx_base + c0 + j;
            if (e < bd || (e == bd && gi < bi)) { bd = e; bi = gi; }
            u = (e < u) ? e : u;
        }
    }
    if (lane == 0) { best_d2[q] = bd; best_idx[q] = bi; }
}

}  // namespace

extern "C" int igan_row_sqnorm(igan_stream_t stream_, const float* a, float* out, int rows, int dim) {
    using namespace igan;
    IGAN_REQUIRE(a && out, "row_sqnorm: null buffer");
    IGAN_REQUIRE(rows >= 1 && dim >= 1, "row_sqnorm: sizes must be positive");
    const int grid = ceil_div(rows, 4);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, a, out, rows, dim);
    IGAN_LAUNCH_CHECK("row_sqnorm launch");
    return IGAN_OK;
}

extern "C" int igan_nn1_update(igan_stream_t stream_, const float* query, const float* qnorm,
                               const float* cand, const float* cnorm, double* best_d2, int* best_idx,
                               float* dots, int nq, int nc, int dim, int idx_base) {
    using namespace igan;
    IGAN_REQUIRE(query && qnorm && cand && cnorm && best_d2 && best_idx && dots, "nn1_update: null buffer");
    IGAN_REQUIRE(nq >= 1 && nc >= 1 && dim >= 1, "nn1_update: sizes must be positive");
    IGAN_REQUIRE(idx_base >= 0 && (long long)idx_base + nc <= INT32_MAX, "nn1_update: candidate index overflows int32");
    igan_conv2d_params p{};          // every optional field (epilogue, noise) zero
    p.x = query; p.w = cand; p.y = dots;
    p.in_scale = nullptr; p.out_scale = nullptr;
    p.workspace = nullptr; p.workspace_floats = 0;
    p.N = nq; p.H = 1; p.W = 1; p.Cin = dim;
    p.OH = 1; p.OW = 1; p.Cout = nc;
    p.KH = 1; p.KW = 1; p.stride = 1; p.up = 1; p.pad_y = 0; p.pad_x = 0;
    p.w_transposed = 1;  // cand is [nc][dim] == forward-layout [1][1][Cout][Cin]
    p.splits = 1;
    p.sliced_tiles = 0;
    p.alpha = 1.0f;
    p.bias = nullptr; p.act = 0; p.act_alpha = 0.0f; p.act_gain = 1.0f;
    if (int rc = igan_conv2d(stream_, &p)) return rc;
    const int grid = ceil_div(nq, 4);
    hipLaunchKernelGGL(nn1_fold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, dots, qnorm, cnorm, query, cand,
                       best_d2, best_idx, nq, nc, dim, idx_base);
    IGAN_LAUNCH_CHECK("nn1_fold launch");
    return IGAN_OK;
}
