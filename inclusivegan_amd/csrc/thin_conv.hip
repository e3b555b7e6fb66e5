// Thin-channel convolutions on gfx950: layers with 3 (at most 4) channels on one side -- ToRGB (Cin -> 3, modulated),
// FromRGB (3 -> Cout), VGG conv1_1 (3 -> 64, 3x3) and their data / weight gradients (networks_stylegan2.py:358-361,
// 433-436; LPIPS first layer).  On the MFMA tiles such a layer pads the thin side to 32 channels: > 90 % of the
// matrix work is multiplication by zero and the layer runs at 1-8 TFLOP/s.  They are streaming problems -- one
// pass over the wide tensor -- so each gets a direct kernel:
//   thin_out:  y[pix][co<4]  = alpha * out_scale * sum_taps sum_ci x[pix+tap][ci] * in_scale[n][ci] * W(tap,ci,co)
//              a group of Cin/4 lanes (at most 64) owns a pixel, 16 B per lane per tap, weights in registers,
//              butterfly reduction over the group
//   thin_in:   y[pix][co]    = alpha * out_scale * sum_taps sum_ci<4 x[pix+tap][ci] * W(tap,ci,co)
//              a lane owns 4 output channels of a pixel, its taps*4 weight vectors in registers, x broadcast
//   thin_wgrad (1x1):  dw = alpha * sum_pix wide[pix][c] * scale[n][c] * thin[pix][t]
//              one sample's pixel range per workgroup, 16 accumulators per lane, fixed-order two-level reduce
// Dispatched from igan_conv2d / igan_conv2d_wgrad (stride 1, up 1); everything else stays on the MFMA kernels.
#include "igan_common.h"
#include <cstdlib>

namespace {

struct ThinArgs {
    const float* x;
    const float* w;
    float* y;
    const float* in_scale;
    const float* out_scale;
    int N, H, W, Cin, OH, OW, Cout;
    int KW, pad_y, pad_x;
    int wt;
    float alpha;
};

// Sum NV per-lane values over a group of `width` lanes (power of two >= NV) with NV - 1 + log2(width / NV) shuffles
// instead of NV * log2(width): at offset 1, 2, 4, ... each lane keeps half of its values and sends the other half
// to its partner (reduce-scatter), then the single remaining value is butterflied over the higher offsets.
// On return lane l holds (in val[0]) the group total of value index  sum_k bit_k(l) * NV >> (k+1).
// Round 6: the stages are a template recursion, so that every index into val[] is a compile-time constant.  Written as a loop over (cnt, off) the compiler
// kept val[] dynamically indexed -- 450 compare / select pairs per call, 6 000 issue cycles per 4 KB of input: thin_out_kernel was bound by THAT (ToRGB at
// 128x128: 1.7 TB/s), not by memory.
template <int CNT, int OFF, int NV>
__device__ __forceinline__ void reduce_scatter_stage(float (&val)[NV], int lane_in_group) {
    if constexpr (CNT > 1) {
        const bool hi = (lane_in_group & OFF) != 0;
#pragma unroll
        for (int i = 0; i < CNT / 2; i++) {
            const float send = hi ? val[i] : val[i + CNT / 2];
            const float keep = hi ? val[i + CNT / 2] : val[i];
            val[i] = keep + __shfl_xor(send, OFF, 64);
        }
        reduce_scatter_stage<CNT / 2, OFF * 2, NV>(val, lane_in_group);
    }
}
template <int NV>
__device__ __forceinline__ void group_reduce_scatter(float (&val)[NV], int lane_in_group, int width) {
    reduce_scatter_stage<NV, 1, NV>(val, lane_in_group);
    for (int off = NV; off < width; off <<= 1) val[0] += __shfl_xor(val[0], off, 64);
}
template <int NV>
__device__ __forceinline__ int reduce_scatter_index(int lane_in_group) {
    int idx = 0, off = 1;
#pragma unroll
    for (int cnt = NV; cnt > 1; cnt >>= 1, off <<= 1) idx += (lane_in_group & off) ? (cnt >> 1) : 0;
    return idx;
}

// ---- thin_out: Cout <= 4, Cin % 4 == 0 ------------------------------------------------------------
// TAPS = KH*KW (1 or 9).  Lanes: gw = min(Cin/4, 64) per pixel; Cin > 256 loops over channel blocks.
template <int TAPS, int CB>   // CB = channel blocks of 256 (Cin = 512 -> 2)
__global__ __launch_bounds__(256) void thin_out_kernel(ThinArgs a, int gw) {
    const int lane = threadIdx.x & 63;
    const int lc = lane % gw, sub = lane / gw, ppw = 64 / gw;
    const int gwave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int npix = a.N * a.OH * a.OW;
    // weights of this lane: W(tap, ci = 4*(lc + gw*cb) + e, co) for cb < CB, e < 4, co < 4 (zero-padded).  Round 6: all of them are fetched by buffer loads whose
    // out-of-range offsets return 0 (no branch per element) and are in flight together -- written as `if (ok) v = w[...]` the compiler made every element a branch
    // with its own wait, a chain of 16 ... 144 dependent round trips per wave in front of a kernel that streams 200 MB (ToRGB at 128x128: 118 us, 1.7 TB/s).
    constexpr int KWc = (TAPS == 9) ? 3 : 1, KHc = TAPS / KWc;      // thin_conv_kind admits 1x1 and 3x3 only
    float wr[CB][TAPS][4][4];
    {
        constexpr unsigned WOOB = 0x7FFFFFF0u;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((unsigned)TAPS * (unsigned)a.Cin * (unsigned)a.Cout * 4u), 0x00020000);
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int t = 0; t < TAPS; t++)
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int co = 0; co < 4; co++) {
                        const int ci = 4 * (lc + gw * cb) + e;
                        const int ky = t / KWc, kx = t - ky * KWc;
                        const unsigned idx = a.wt ? (unsigned)(((KHc - 1 - ky) * KWc + (KWc - 1 - kx)) * a.Cout + co) * (unsigned)a.Cin + (unsigned)ci
                                                  : ((unsigned)t * (unsigned)a.Cin + (unsigned)ci) * (unsigned)a.Cout + (unsigned)co;
                        const bool ok = co < a.Cout && ci < a.Cin;
                        wr[cb][t][e][co] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, ok ? idx * 4u : WOOB, 0, 0));
                    }
    }
    // U pixel steps per iteration with all their loads in flight together (one load per iteration is a latency chain)
    constexpr int U = (TAPS == 1) ? 4 : 1;
    for (int p0 = gwave * ppw * U; p0 < npix; p0 += nwaves * ppw * U) {      // wave-uniform trip count
        float4 v[U][CB][TAPS];
        float4 sc[U][CB];
        int pn[U];
        bool okp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = p0 + u * ppw + sub;
            okp[u] = p < npix;
            const int pp = okp[u] ? p : 0;
            const int n = pp / (a.OH * a.OW);
            int oy = 0, ox = 0;
            if constexpr (TAPS != 1) {
                const int r = pp - n * (a.OH * a.OW);
                oy = r / a.OW; ox = r - oy * a.OW;
            }
            pn[u] = n;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const int ci = 4 * (lc + gw * cb);
                sc[u][cb] = a.in_scale ? *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + ci) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
                for (int t = 0; t < TAPS; t++) {
                    const int ky = t / KWc, kx = t - ky * KWc;
                    const int iy = oy + ky - a.pad_y, ix = ox + kx - a.pad_x;
                    // 1x1 (pad 0, same size: thin_conv_kind): the input pixel is the output pixel
                    const bool in = (TAPS == 1) ? okp[u] : (okp[u] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W);
                    const size_t off = !in ? (size_t)ci : (TAPS == 1) ? (size_t)pp * a.Cin + ci
                                                                      : ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + ci;   // always a valid address
                    const float4 q = *reinterpret_cast<const float4*>(a.x + off);
                    v[u][cb][t] = in ? q : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        float acc[U * 4];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int co = 0; co < 4; co++) acc[u * 4 + co] = 0.f;
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int t = 0; t < TAPS; t++) {
                    const float4 q = make_float4(v[u][cb][t].x * sc[u][cb].x, v[u][cb][t].y * sc[u][cb].y,
                                                 v[u][cb][t].z * sc[u][cb].z, v[u][cb][t].w * sc[u][cb].w);
#pragma unroll
                    for (int co = 0; co < 4; co++)
                        acc[u * 4 + co] = fmaf(q.x, wr[cb][t][0][co], fmaf(q.y, wr[cb][t][1][co], fmaf(q.z, wr[cb][t][2][co], fmaf(q.w, wr[cb][t][3][co], acc[u * 4 + co]))));
                }
        }
        // the group's U*4 sums land one per lane: lane lc (< U*4 distinct indices) owns (pixel step, channel) = idx
        group_reduce_scatter<U * 4>(acc, lc, gw);
        if (lc < U * 4) {
            const int idx = reduce_scatter_index<U * 4>(lc);
            const int u = idx >> 2, co = idx & 3;
            const int p = p0 + u * ppw + sub;
            // okp / pn are per-lane arrays indexed by a lane-dependent u: select without dynamic indexing
            bool ok = false;
            int n = 0;
#pragma unroll
            for (int q = 0; q < U; q++) { ok = (q == u) ? okp[q] : ok; n = (q == u) ? pn[q] : n; }
            if (ok && co < a.Cout) {
                float o = acc[0] * a.alpha;
                if (a.out_scale) o *= a.out_scale[n * a.Cout + co];
                a.y[(size_t)p * a.Cout + co] = o;
            }
        }
    }
}

// ---- thin_in: Cin <= 4, Cout % 4 == 0 -------------------------------------------------------------
// Lanes: cg = min(Cout/4, 64) per pixel; Cout > 256 loops over channel blocks (<= 2).
template <int TAPS>
__global__ __launch_bounds__(256) void thin_in_kernel(ThinArgs a, int cg, int cblocks) {
    const int lane = threadIdx.x & 63;
    const int lc = lane % cg, sub = lane / cg, ppw = 64 / cg;
    const int gwave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int npix = a.N * a.OH * a.OW;
    for (int cb = 0; cb < cblocks; cb++) {
        const int co = 4 * (lc + cg * cb);
        // weights: W(tap, ci < 4, co..co+3), zero-padded in ci
        float4 wr[TAPS][4];
#pragma unroll
        for (int t = 0; t < TAPS; t++)
#pragma unroll
            for (int ci = 0; ci < 4; ci++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < a.Cin && co < a.Cout) {
                    const int ky = t / a.KW, kx = t - ky * a.KW;
                    const int KH = TAPS / a.KW;
                    if (a.wt) {
                        const float* b = a.w + ((size_t)((KH - 1 - ky) * a.KW + (a.KW - 1 - kx)) * a.Cout + co) * a.Cin + ci;
                        v = make_float4(b[0], b[a.Cin], b[2 * a.Cin], b[3 * a.Cin]);
                    } else {
                        v = *reinterpret_cast<const float4*>(a.w + ((size_t)t * a.Cin + ci) * a.Cout + co);
                    }
                }
                wr[t][ci] = v;
            }
        for (int p0 = gwave * ppw; p0 < npix; p0 += nwaves * ppw) {
            const int p = p0 + sub;
            if (p >= npix) continue;
            const int n = p / (a.OH * a.OW);
            const int r = p - n * (a.OH * a.OW);
            const int oy = r / a.OW, ox = r - oy * a.OW;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < TAPS; t++) {
                const int ky = t / a.KW, kx = t - ky * a.KW;
                const int iy = oy + ky - a.pad_y, ix = ox + kx - a.pad_x;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                    const float* xp = a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin;
#pragma unroll
                    for (int ci = 0; ci < 4; ci++) {
                        if (ci < a.Cin) {
                            const float xv = xp[ci];
                            acc.x = fmaf(xv, wr[t][ci].x, acc.x); acc.y = fmaf(xv, wr[t][ci].y, acc.y);
                            acc.z = fmaf(xv, wr[t][ci].z, acc.z); acc.w = fmaf(xv, wr[t][ci].w, acc.w);
                        }
                    }
                }
            }
            if (co < a.Cout) {
                float4 v = make_float4(acc.x * a.alpha, acc.y * a.alpha, acc.z * a.alpha, acc.w * a.alpha);
                if (a.out_scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(a.out_scale + (size_t)n * a.Cout + co);
                    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
                }
                *reinterpret_cast<float4*>(a.y + (size_t)p * a.Cout + co) = v;
            }
        }
    }
}

// ---- thin_in, 3x3 (VGG conv1_1: 3 -> 64): sliding window along x ----------------------------------
// A lane owns 4 output channels and walks a run of RUN output pixels of one row; its 3 x 3 x CIN input window lives in registers
// and moves one column per pixel (9 loads per pixel instead of 27, the three column buffers rotate through an unroll by three
// instead of being copied), the 27 weight vectors stay in registers.  The cg lanes of a pixel group read the same addresses
// (one broadcast request); their float4 outputs are one contiguous row segment.
template <int CIN>
__global__ __launch_bounds__(256) void thin_in3x3_kernel(ThinArgs a, int cg, int run, int runs_per_row) {
    const int lane = threadIdx.x & 63;
    const int lc = lane % cg, sub = lane / cg, ppw = 64 / cg;
    const int gwave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int co = 4 * lc;
    const long long total_runs = (long long)a.N * a.OH * runs_per_row;
    float4 wr[9][CIN];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < a.Cout) {
                const int ky = t / 3, kx = t - ky * 3;
                if (a.wt) {
                    const float* b = a.w + ((size_t)((2 - ky) * 3 + (2 - kx)) * a.Cout + co) * CIN + ci;
                    v = make_float4(b[0], b[CIN], b[2 * CIN], b[3 * CIN]);
                } else {
                    v = *reinterpret_cast<const float4*>(a.w + ((size_t)t * CIN + ci) * a.Cout + co);
                }
            }
            wr[t][ci] = v;
        }
    for (long long rr = (long long)gwave * ppw + sub; rr < total_runs; rr += (long long)nwaves * ppw) {
        const int xr = (int)(rr % runs_per_row);
        const long long row = rr / runs_per_row;
        const int oy = (int)(row % a.OH);
        const int n = (int)(row / a.OH);
        const int x0 = xr * run, x1 = min(x0 + run, a.OW);
        const float* rowp[3];
        bool rowok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const int iy = oy + ky - a.pad_y;
            rowok[ky] = (unsigned)iy < (unsigned)a.H;
            rowp[ky] = a.x + ((size_t)(n * a.H + (rowok[ky] ? iy : 0)) * a.W) * CIN;
        }
        float c0[3][CIN], c1[3][CIN], c2[3][CIN];
        auto load_col = [&](float (&c)[3][CIN], int ix) {
            const bool ok = (unsigned)ix < (unsigned)a.W;
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) c[ky][ci] = (ok && rowok[ky]) ? rowp[ky][(size_t)ix * CIN + ci] : 0.f;
        };
        auto emit = [&](const float (&k0)[3][CIN], const float (&k1)[3][CIN], const float (&k2)[3][CIN], int ox) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) {
                    const float v0 = k0[ky][ci], v1 = k1[ky][ci], v2 = k2[ky][ci];
                    const float4 w0 = wr[ky * 3 + 0][ci], w1 = wr[ky * 3 + 1][ci], w2 = wr[ky * 3 + 2][ci];
                    acc.x = fmaf(v0, w0.x, acc.x); acc.y = fmaf(v0, w0.y, acc.y); acc.z = fmaf(v0, w0.z, acc.z); acc.w = fmaf(v0, w0.w, acc.w);
                    acc.x = fmaf(v1, w1.x, acc.x); acc.y = fmaf(v1, w1.y, acc.y); acc.z = fmaf(v1, w1.z, acc.z); acc.w = fmaf(v1, w1.w, acc.w);
                    acc.x = fmaf(v2, w2.x, acc.x); acc.y = fmaf(v2, w2.y, acc.y); acc.z = fmaf(v2, w2.z, acc.z); acc.w = fmaf(v2, w2.w, acc.w);
                }
            if (co < a.Cout) {
                float4 v = make_float4(acc.x * a.alpha, acc.y * a.alpha, acc.z * a.alpha, acc.w * a.alpha);
                if (a.out_scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(a.out_scale + (size_t)n * a.Cout + co);
                    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
                }
                *reinterpret_cast<float4*>(a.y + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.Cout + co) = v;
            }
        };
        load_col(c0, x0 - a.pad_x);
        load_col(c1, x0 + 1 - a.pad_x);
        for (int ox = x0; ox < x1; ox += 3) {          // three pixels per trip: the column buffers rotate, nothing is copied
            load_col(c2, ox + 2 - a.pad_x);
            emit(c0, c1, c2, ox);
            if (ox + 1 < x1) {
                load_col(c0, ox + 3 - a.pad_x);
                emit(c1, c2, c0, ox + 1);
            }
            if (ox + 2 < x1) {
                load_col(c1, ox + 4 - a.pad_x);
                emit(c2, c0, c1, ox + 2);
            }
        }
    }
}

// ---- thin_wgrad (1x1, stride 1): partial[block][c][t] = scale[n][c] * sum_{pixels of the block} wide[pix][c] * thin[pix][t]
struct ThinWgArgs {
    const float* wide;      // [N, HW, C]
    const float* thin;      // [N, HW, T], T <= 4
    const float* scale;     // [N, C] or NULL
    float* partial;         // [N * bps][C][4]
    int N, HW, C, T, bps;   // bps = blocks per sample
};

__global__ __launch_bounds__(256) void thin_wgrad_kernel(ThinWgArgs a) {
    __shared__ float red[256 * 17];
    __shared__ float ths[4 * 1024];               // the block's pixels of the thin operand, [pixel][4] (zero-padded)
    const int tid = threadIdx.x;
    const int cg = min(a.C >> 2, 256);            // lanes per pixel (C <= 1024)
    const int lc = tid % cg, sub = tid / cg, ppb = 256 / cg;
    const int n = blockIdx.x / a.bps, b = blockIdx.x - n * a.bps;
    const int per = (a.HW + a.bps - 1) / a.bps;
    const int p0 = b * per, p1 = min(p0 + per, a.HW);
    float acc[4][4];
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[e][t] = 0.f;
    // pixels in slabs of 1024: the thin operand of a slab goes through LDS (coalesced once, then broadcast reads)
    for (int s0 = p0; s0 < p1; s0 += 1024) {
      const int s1 = min(s0 + 1024, p1);
      __syncthreads();
      for (int i = tid; i < (s1 - s0) * 4; i += 256) {
          const int pl = i >> 2, t = i & 3;
          ths[i] = (t < a.T) ? a.thin[((size_t)n * a.HW + s0 + pl) * a.T + t] : 0.f;
      }
      __syncthreads();
      // 4 pixel steps per iteration, all loads in flight together (clamped addresses, masked values)
      for (int pb = s0 + sub; pb < s1; pb += 4 * ppb) {
        float4 v[4];
        float th[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = pb + u * ppb;
            const bool ok = p < s1;
            const size_t pix = (size_t)n * a.HW + (ok ? p : s0);
            v[u] = *reinterpret_cast<const float4*>(a.wide + pix * a.C + 4 * lc);
            const float4 tq = *reinterpret_cast<const float4*>(ths + 4 * ((ok ? p : s0) - s0));
            th[u][0] = ok ? tq.x : 0.f; th[u][1] = ok ? tq.y : 0.f; th[u][2] = ok ? tq.z : 0.f; th[u][3] = ok ? tq.w : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[0][t] = fmaf(v[u].x, th[u][t], acc[0][t]); acc[1][t] = fmaf(v[u].y, th[u][t], acc[1][t]);
                acc[2][t] = fmaf(v[u].z, th[u][t], acc[2][t]); acc[3][t] = fmaf(v[u].w, th[u][t], acc[3][t]);
            }
      }
    }
    // fixed-order reduce over the ppb pixel sub-lanes of each channel group
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int t = 0; t < 4; t++) red[tid * 17 + e * 4 + t] = acc[e][t];
    __syncthreads();
    if (tid < cg) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + (size_t)n * a.C + 4 * tid);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < ppb; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) s[t] += red[(q * cg + tid) * 17 + e * 4 + t];
            *reinterpret_cast<float4*>(a.partial + ((size_t)blockIdx.x * a.C + 4 * tid + e) * 4) =
                make_float4(s[0] * scv[e], s[1] * scv[e], s[2] * scv[e], s[3] * scv[e]);
        }
    }
}

// dw = alpha * sum_blocks partial[block][c][t]; thin_out: dw[c][t] ([Cin][Cout]); thin_in: dw[t][c] ([Cin][Cout]).
// One wave per output: lane l adds blocks l, l+64, ... (independent loads in flight), then a fixed-order butterfly.
__global__ __launch_bounds__(256) void thin_wgrad_final_kernel(const float* partial, float* dw, int blocks, int C, int T, int thin_is_out, float alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= C * T) return;
    const int c = i / T, t = i - c * T;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += partial[((size_t)b * C + c) * 4 + t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) dw[thin_is_out ? c * T + t : t * C + c] = s * alpha;
}

}  // namespace

namespace igan {

// 0 = not a thin layer, 1 = thin output (Cout <= 4), 2 = thin input (Cin <= 4)
// IGAN_THIN=0 routes the thin layers back to the MFMA tiles (A/B runs)
static bool thin_enabled() {
    static const bool on = !(getenv("IGAN_THIN") && atoi(getenv("IGAN_THIN")) == 0);
    return on;
}

int thin_conv_kind(const igan_conv2d_params* p) {
    if (!thin_enabled() || p->stride != 1 || p->up != 1 || p->act != 0) return 0;
    const int taps = p->KH * p->KW;
    if (!(taps == 1 || (p->KH == 3 && p->KW == 3))) return 0;
    if (taps == 1 && (p->pad_y != 0 || p->pad_x != 0 || p->OH != p->H || p->OW != p->W)) return 0;
    if ((((uintptr_t)p->x | (uintptr_t)p->w | (uintptr_t)p->y | (uintptr_t)p->in_scale | (uintptr_t)p->out_scale) & 15) != 0) return 0;
    if (p->Cout <= 4 && p->Cin % 4 == 0 && p->Cin >= 64 && p->Cin <= (taps == 1 ? 512 : 256)) {     // >= 16 lanes per pixel
        const int cv = p->Cin / 4;
        if ((cv & (cv - 1)) == 0) return 1;          // power-of-two lane groups
    }
    if (p->Cin <= 4 && p->Cout % 4 == 0 && p->Cout >= 16 && p->Cout <= 512 && !p->in_scale) {
        const int cv = p->Cout / 4;
        if ((cv & (cv - 1)) == 0) return 2;
    }
    return 0;
}

void thin_conv(hipStream_t stream, const igan_conv2d_params* p, int kind) {
    ThinArgs a;
    a.x = p->x; a.w = p->w; a.y = p->y; a.in_scale = p->in_scale; a.out_scale = p->out_scale;
    a.N = p->N; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.OH = p->OH; a.OW = p->OW; a.Cout = p->Cout;
    a.KW = p->KW; a.pad_y = p->pad_y; a.pad_x = p->pad_x; a.wt = p->w_transposed; a.alpha = p->alpha;
    const long long npix = (long long)p->N * p->OH * p->OW;
    const int taps = p->KH * p->KW;
    const int cv = (kind == 1 ? p->Cin : p->Cout) / 4;
    const int gw = std::min(cv, 64), cblocks = ceil_div(cv, 64);
    const int ppw = 64 / gw;
    const int grid = (int)std::min<long long>(ceil_div_ll(npix, 4LL * ppw * 8), 256 * 6);    // >= 8 pixel steps per wave, <= 6 workgroups per CU
    if (kind == 1) {
        if (taps == 1 && cblocks == 2) hipLaunchKernelGGL((thin_out_kernel<1, 2>), dim3(std::max(grid, 1)), dim3(256), 0, stream, a, gw);
        else if (taps == 1) hipLaunchKernelGGL((thin_out_kernel<1, 1>), dim3(std::max(grid, 1)), dim3(256), 0, stream, a, gw);
        else hipLaunchKernelGGL((thin_out_kernel<9, 1>), dim3(std::max(grid, 1)), dim3(256), 0, stream, a, gw);
    } else {
        static const bool slide = !(getenv("IGAN_THIN_SLIDE") && atoi(getenv("IGAN_THIN_SLIDE")) == 0);      // A/B switch
        if (taps == 1) hipLaunchKernelGGL((thin_in_kernel<1>), dim3(std::max(grid, 1)), dim3(256), 0, stream, a, gw, cblocks);
        else if (slide && p->Cin == 3 && cblocks == 1 && p->OW >= 16) {      // 3x3 from 3 channels: sliding window along x
            const int run = 32, runs_per_row = ceil_div(p->OW, run);
            const long long runs = (long long)p->N * p->OH * runs_per_row;
            const int g = (int)std::min<long long>(ceil_div_ll(runs, 4LL * ppw), 256 * 8);
            hipLaunchKernelGGL((thin_in3x3_kernel<3>), dim3(std::max(g, 1)), dim3(256), 0, stream, a, gw, run, runs_per_row);
        } else hipLaunchKernelGGL((thin_in_kernel<9>), dim3(std::max(grid, 1)), dim3(256), 0, stream, a, gw, cblocks);
    }
}

int thin_wgrad_kind(const igan_conv2d_wgrad_params* p) {
    if (!thin_enabled()) return 0;
    if (p->stride != 1 || p->up != 1 || p->KH != 1 || p->KW != 1 || p->pad_y != 0 || p->pad_x != 0) return 0;
    if (p->OH != p->H || p->OW != p->W) return 0;
    if ((((uintptr_t)p->x | (uintptr_t)p->dy | (uintptr_t)p->in_scale | (uintptr_t)p->out_scale) & 15) != 0) return 0;
    if (p->Cout <= 4 && p->Cin % 4 == 0 && p->Cin >= 16 && p->Cin <= 1024 && !p->out_scale && 256 % (p->Cin / 4) == 0) return 1;
    if (p->Cin <= 4 && p->Cout % 4 == 0 && p->Cout >= 16 && p->Cout <= 1024 && !p->in_scale && 256 % (p->Cout / 4) == 0) return 2;
    return 0;
}

// blocks per sample: enough workgroups to fill the machine, at least 64 pixels each
int thin_wgrad_bps(const igan_conv2d_wgrad_params* p) {
    const int hw = p->H * p->W;
    return std::max(1, std::min(ceil_div(1536, p->N), hw / 64 > 0 ? hw / 64 : 1));
}

size_t thin_wgrad_workspace(const igan_conv2d_wgrad_params* p, int kind) {
    const int C = (kind == 1) ? p->Cin : p->Cout;
    // never less than what the MFMA path would need for the same `splits` (it is the fallback for a misaligned workspace)
    return std::max((size_t)p->N * thin_wgrad_bps(p) * C * 4, (size_t)2 * p->Cin * p->Cout);
}

void thin_wgrad(hipStream_t stream, const igan_conv2d_wgrad_params* p, int kind) {
    ThinWgArgs a;
    const bool thin_out = (kind == 1);
    a.wide = thin_out ? p->x : p->dy;
    a.thin = thin_out ? p->dy : p->x;
    a.scale = thin_out ? p->in_scale : p->out_scale;
    a.partial = p->workspace;
    a.N = p->N; a.HW = p->H * p->W;
    a.C = thin_out ? p->Cin : p->Cout;
    a.T = thin_out ? p->Cout : p->Cin;
    a.bps = thin_wgrad_bps(p);
    const int blocks = p->N * a.bps;
    hipLaunchKernelGGL(thin_wgrad_kernel, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(thin_wgrad_final_kernel, dim3(ceil_div(a.C * a.T, 4)), dim3(256), 0, stream,
                       (const float*)p->workspace, p->dw, blocks, a.C, a.T, thin_out ? 1 : 0, p->alpha);
}

}  // namespace igan
