// conv2d / conv2d_transpose / matmul and their data- and weight-gradients for
// gfx950, as ONE implicit-GEMM family on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32.
//
// Behavioural contract (what the reference gets from TensorFlow + cuDNN):
//   tf.nn.conv2d SAME stride 1      training/networks_stylegan2.py:60,120
//   tf.nn.conv2d VALID stride 2     dnnlib/tflib/ops/upfirdn_2d.py:332
//   tf.nn.conv2d_transpose stride 2 dnnlib/tflib/ops/upfirdn_2d.py:286-291
//   tf.matmul (dense_layer)         training/networks_stylegan2.py:41-46
// and, through tf.gradients, their input and filter gradients (first and second
// order -- the data gradient of one geometry is the forward op of the mirrored
// geometry, see include/igan_hip.h).  fp32 in, fp32 accumulate, like the reference
// (dtype='float32', networks_stylegan2.py:264,323,422).
//
// A second, switchable form of the large 3x3 layers (IGAN_CONV_PLANES=1: conv_fwd_planes_kernel, conv_wgrad_planes_kernel, with
// to_planes_kernel / filter_planes_kernel) runs the same products on the bf16 matrix instruction from three bf16 pieces per fp32
// operand.  It is a labelled VARIANT, never selected unless the environment asks for it.
//
// MI355X design (none of it is in the reference, which calls cuDNN):
//  * activations are channel-minor [N,H,W,C]: an A-tile row (one output pixel, 32
//    input channels of one tap) is a contiguous 128 B segment -> coalesced 16 B
//    loads, no im2col buffer ever exists in HBM;
//  * conv2d_transpose is NOT run as a zero-stuffed convolution: output pixels are
//    partitioned into up*up parity classes, each class being a dense conv with
//    the sub-lattice of taps that hits real samples (2.25 instead of 9 taps for
//    3x3, up 2), one grid.z slice per class;
//  * block tile BM x BN x 32, 4 wavefronts (one per SIMD), each owning TM x TN
//    tiles of 32x32 accumulators (16 VGPRs each); the two 32-lane halves of a wave
//    take k = 0..15 and 16..31 of the chunk (any bijection of k is legal for a
//    reduction) so a lane's 16 A values are 64 contiguous bytes in LDS
//    (4 x ds_read_b128, conflict-free with the 36-float row pitch);
//  * register-staged double buffering: the global loads of chunk c+1 are issued
//    before the 16*TM*TN MFMAs of chunk c and written to the other LDS buffer
//    after them -- one barrier per chunk;
//  * StyleGAN2 modulation / demodulation ride along as per-(sample,channel)
//    scales on the A-operand load and in the epilogue (non-fused modconv form,
//    networks_stylegan2.py:112,126), so x*s and y*d never round-trip HBM;
//  * the grid is a 1-D list of tiles dealt round-robin to the 256 CUs (two resident per CU).  A layer
//    of T tiles leaves T mod 256 tiles for a last, partly empty round (all of them, for the small
//    4x4 .. 16x16 layers): only those tail tiles are cut along the reduction axis so that the tail
//    occupies every CU.  Their partial tiles go to a caller-owned tile-compact workspace and a fix-up
//    kernel adds them in fixed order (bit-reproducible, no float atomics) -- the fix-up traffic is that
//    of the tail tiles only, not of the whole output.
#include "igan_common.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;        // reduction chunk (floats)
#ifndef IGAN_SPLIT
#define IGAN_SPLIT 8          // k steps (of 16) issued before the staged chunk is written to LDS: the lead time of the global loads
#endif
constexpr int LDK = BK + 4;   // row pitch of a [rows][k] LDS image (144 B: 16 B aligned, conflict-free b128 reads)

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;          // workspace of the sliced tail tiles (splits > 1), else unused
    float* out;        // final output
    const float* in_scale;
    const float* out_scale;
    int N, H, W, Cin;
    int OH, OW, Cout;
    int KH, KW;
    int stride, up_shift;   // up == 1 << up_shift
    int pad_y, pad_x;
    int splits;             // reduction slices of each TAIL tile (1 = no workspace)
    int full_tiles;         // leading tiles computed whole; tiles >= full_tiles are sliced
    int nx, ny;             // m tiles (largest class) and n tiles; tile id = (cls * ny + nt) * nx + mt
    int cpt;                // chunks per tap = ceil(Cin / 32)
    int Mtot;               // N*OH*OW
    int vecA, vecB, vecS;   // 16 B paths usable for x rows / w rows / in_scale rows
    int vecY;               // 8 B output stores usable (Cout even, y / out_scale 8 B aligned)
    float alpha;            // output multiplier (applied here when splits == 1, else by the reduce kernel)
    int xcd_remap;          // XCD-aware block order (remap_xcd)
    int b_scale;            // LDS-DMA kernel: one-sample tiles apply the modulation to the B fragments (A/B switch IGAN_CONV_BSCALE)
    int prio;               // LDS-DMA kernel: raised issue priority during the tile prologue (A/B switch IGAN_CONV_PROLOGUE_PRIO)
    int walk;               // walking address computation usable (16 B paths, Cin % 32 == 0)
    int stagger;            // start delay (64-cycle quanta) of the workgroup in the upper LDS slot (0 = none)
    unsigned long long* diag;   // diagnostic build-in: 4 time stamps (100 MHz ticks) per workgroup, or nullptr
    const float* bias;      // fused epilogue (act != 0): y = act(y + noise[n, pixel] * strength + bias[co]) * act_gain
    int act;                // 0 none, 1 linear, 2 relu, 3 lrelu
    float act_alpha, act_gain;
    const float* noise;     // [N or 1, OH, OW] or nullptr; strength = *noise_strength (device scalar)
    const float* noise_strength;
    int noise_bcast;        // noise has one sample, shared by the batch
    int diag_mode;              // IGAN_DIAGNOSTIC builds only (tools/coresidency_probe.py): bit mask of parts of conv_fwd_planes_kernel<2> to leave out; 0 in the product
    const unsigned short* xp;   // bf16-piece form (conv_fwd_planes_kernel): x * in_scale as [pixel][Cin/16][3 pieces][16] bf16
    const unsigned short* wp;   //   and the filter as [tap][n][Cin/16][3][16] (the orientation is resolved when it is written)
};

// noise[n, pixel] * strength of output pixel `pix` (linear index over [N, OH, OW]); 0 without noise or for padding rows
__device__ __forceinline__ float noise_term(const ConvArgs& a, int pix) {
    if (a.noise == nullptr || pix < 0) return 0.0f;
    return a.noise[a.noise_bcast ? pix % (a.OH * a.OW) : pix] * a.noise_strength[0];
}

__device__ __forceinline__ float epi_act(int act, float v, float alpha) {
    if (act == 2) return v > 0.f ? v : 0.f;
    if (act == 3) return v > 0.f ? v : v * alpha;
    return v;
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
// scale operand: the loaded values when the scale exists, 1.0 otherwise (select, not branch)
__device__ __forceinline__ float4 f4sel(bool has, float4 s) { return make_float4(has ? s.x : 1.f, has ? s.y : 1.f, has ? s.z : 1.f, has ? s.w : 1.f); }

// Predicated operand loads through buffer descriptors: an out-of-range offset makes the hardware
// return 0 for that lane, so padding taps, ragged tile edges and channel tails need no branch and no
// select -- all of a chunk's loads issue back to back (a per-load exec-mask branch would serialise
// them behind s_waitcnt).  Offsets are 32-bit bytes, hence the 2 GiB-per-operand limit in the ABI.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7FFFFFF0u;   // >= any num_records we create

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}

// Address of 4 consecutive floats at element offset `off`, of which `left` (may be <= 0) are valid.
// VEC: off % 4 == 0, base 16 B aligned, all four in or all four out -> one byte offset (or OOB).
// !VEC: byte offset of element 0 (never OOB) plus the valid count, resolved per element at load time.
struct LoadAddr {
    unsigned off;
    int left;
};
template <bool VEC>
__device__ __forceinline__ LoadAddr make_addr(int off, int left) {
    LoadAddr a;
    if constexpr (VEC) { a.off = (left > 0) ? (unsigned)off * 4u : OOB; a.left = 0; }
    else { a.off = (unsigned)off * 4u; a.left = left; }
    return a;
}
// Same, from a running BYTE offset of element 0 (the weight-gradient loaders walk pixels incrementally).
template <bool VEC>
__device__ __forceinline__ LoadAddr make_addr_b(unsigned byte_off, int left) {
    LoadAddr a;
    if constexpr (VEC) { a.off = (left > 0) ? byte_off : OOB; a.left = 0; }
    else { a.off = byte_off; a.left = left; }
    return a;
}
template <bool VEC>
__device__ __forceinline__ float4 load4(__amdgpu_buffer_rsrc_t r, LoadAddr a) {
    if constexpr (VEC) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, a.off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    } else {
        float4 f;
        f.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (a.left > 0) ? a.off + 0u : OOB, 0, 0));
        f.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (a.left > 1) ? a.off + 4u : OOB, 0, 0));
        f.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (a.left > 2) ? a.off + 8u : OOB, 0, 0));
        f.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (a.left > 3) ? a.off + 12u : OOB, 0, 0));
        return f;
    }
}

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8, each with its own
// 4 MiB L2), so with the plain order neighbouring tiles -- which read overlapping input rows (the 3x3 halo) or the same
// pixel slice under different taps -- land on eight different L2s and every one of them fetches the shared rows again.
// remap_xcd() returns the logical index of block b such that XCD x works through ONE contiguous range of logical indices
// (bijective for any grid size): blocks that are resident together on an XCD are neighbours in tile order and share their
// halo through that XCD's L2.  Placement is a speed matter only; no result depends on it.
__device__ __forceinline__ int remap_xcd(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// One chunk of MFMAs for this wave.  A image: [m][k] (A_KMAJOR = false, pitch LDK)
// or [k][m] (A_KMAJOR = true, pitch LDA).  B image: [k][n] (B_KMAJOR = true, pitch
// LDB) or [n][k] (B_KMAJOR = false, pitch LDK).
// Row -> accumulator-tile assignment inside the wave's span (any bijection is legal, the epilogue
// undoes it through tile_row()):
//   [m][k] image: tile t holds rows  t*32 + l31          (a lane's 16 k values: 4 x ds_read_b128)
//   [k][m] image: tile t holds rows  T*l31 + t           (a lane's T tiles are T CONSECUTIVE floats
//                 of one k row: one ds_read_b64 for T = 2 instead of two half-rate ds_read_b32)
template <int T, bool KMAJOR>
__device__ __forceinline__ int tile_row(int t, int l) { return KMAJOR ? T * l + t : t * 32 + l; }

template <int T, bool KMAJOR, int LD>
__device__ __forceinline__ void load_frag(const float* __restrict__ S, int r0, int l31, int h, float (&f)[T][16]) {
    if constexpr (!KMAJOR) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float4* p = reinterpret_cast<const float4*>(S + (r0 + t * 32 + l31) * LDK + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = p[q];
                f[t][4 * q + 0] = v.x; f[t][4 * q + 1] = v.y; f[t][4 * q + 2] = v.z; f[t][4 * q + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float* p = S + (16 * h + j) * LD + r0 + T * l31;
            if constexpr (T == 2) {
                const float2 v = *reinterpret_cast<const float2*>(p);
                f[0][j] = v.x; f[1][j] = v.y;
            } else if constexpr (T == 4) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                f[0][j] = v.x; f[1][j] = v.y; f[2][j] = v.z; f[3][j] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < T; t++) f[t][j] = p[t];
            }
        }
    }
}

// MFMAs of k steps [J0, J1) of the chunk whose fragments are in af / bf.
// PRIO: raise the wave's issue priority while it is in a matrix cluster, so that another wave's address / staging VALU work
// cannot delay its next MFMA.  Measured (sustained, B = 6, 128x128 C128): weight-gradient kernel 269 -> 255 us (+5.6 %);
// forward kernel 244 -> 252 us (its staging stores then wait behind every other wave's matrix cluster); the 8-wave form of the
// weight-gradient kernel loses 20 %, the plain 4-wave form 1 % -- so only the 4-wave modulated weight-gradient kernel sets it.
template <int TM, int TN, int J0, int J1, bool PRIO = false>
__device__ __forceinline__ void mma_steps(const float (&af)[TM][16], const float (&bf)[TN][16], f32x16 (&acc)[TM][TN]) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = J0; j < J1; j++)
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j], acc[tm][tn], 0, 0, 0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
}

// ------------------------------------------------------------------------------
// Forward-type kernel (conv, transposed conv, dense, and all data gradients).
// grid = (m tiles of the largest class, n tiles, classes * splits)
// SC: an in_scale operand exists (host dispatch on the pointer) -- without it the scale loads,
// their addresses and the multiplies are not in the loop at all.
template <int BM, int BN, int WM, int WN, bool WT, bool VEC, bool SC>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 16) ? 8 : ((WM * WN == 8) ? 4 : 2)) void conv_fwd_kernel(ConvArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int LDB = WT ? LDK : BN + 4;
    constexpr int A_ELEMS = BM * LDK;
    constexpr int B_ELEMS = WT ? BN * LDK : BK * (BN + 4);
    constexpr int NT = WM * WN * 64;     // threads: 4 waves (one per SIMD) or 8 (two per SIMD)
    constexpr int AROWS = NT / 8;        // tile rows covered by one pass of the loaders (8 float4 per 32-k row)
    constexpr int AR = BM / AROWS;       // A rows per thread
    constexpr int BR = BN / AROWS;       // B float4 per thread (both layouts: BK * BN / 4 / NT)
    static_assert(TM >= 1 && TN >= 1 && (WM * WN == 4 || WM * WN == 8 || WM * WN == 16) && AR >= 1 && BR >= 1, "bad tile config");

    __shared__ __attribute__((aligned(16))) float As[2 * A_ELEMS];
    __shared__ __attribute__((aligned(16))) float Bs[2 * B_ELEMS];
    __shared__ int row_pix[BM];  // linear output pixel (n*OH+oy)*OW+ox, or -1
    __shared__ int row_n[BM];
    __shared__ float row_nz[BM];  // noise * strength of the row's pixel (fused epilogue)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int up = 1 << a.up_shift;
    auto stamp = [&](int k) {       // diagnostic only (a.diag == nullptr in every normal launch)
        if (a.diag != nullptr && (threadIdx.x >> 6) == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if ((threadIdx.x & 63) == 0) a.diag[(size_t)blockIdx.x * 4 + k] = t;
        }
    };
    stamp(0);
    if (a.stagger > 0) {
        // Two workgroups share a CU and run the same program on equal tiles: left alone they stay in phase, and their
        // barrier / staging bubbles coincide.  The one that did not get the bottom of the CU's LDS starts `stagger`
        // sleep quanta (64 cycles each) late, i.e. about half a chunk out of phase.
        const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6) & 0xFF;   // HW_REG_LDS_ALLOC.LDS_BASE
        if (lds_base != 0)
            for (int i = 0; i < a.stagger; i++) __builtin_amdgcn_s_sleep(1);
    }
    // block -> (tile, reduction slice): whole tiles first, then the sliced tail
    // XCD-aware order INSIDE groups of equal-cost tiles only: the whole tiles of one parity class (a transposed conv's classes
    // have 1 / 2 / 2 / 4 taps; the sliced tail tiles are short).  Remapping across groups would hand one XCD all the heavy
    // tiles and another all the light ones (measured: the up = 2 layers ran at half speed).
    int bid = blockIdx.x;
    if (a.xcd_remap && bid < a.full_tiles) {
        const int per_class = a.nx * a.ny;
        const int cls0 = bid / per_class;
        const int lo = cls0 * per_class;
        const int cnt = min(per_class, a.full_tiles - lo);
        bid = lo + remap_xcd(bid - lo, cnt);
    }
    const bool sliced = bid >= a.full_tiles;
    const int tail = bid - a.full_tiles;
    const int tile = sliced ? a.full_tiles + tail / a.splits : bid;
    const int split = sliced ? tail % a.splits : 0;
    const int nsplit = sliced ? a.splits : 1;
    const int mt = tile % a.nx, nt = (tile / a.nx) % a.ny, cls = tile / (a.nx * a.ny);
    const int py = cls >> a.up_shift, px = cls & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Mcls = a.N * QH * QW;
    const int m0 = mt * BM;
    if (m0 >= Mcls) return;  // uniform: smaller classes have fewer tiles
    const int n0 = nt * BN;

    // taps of this class: ky = ky0 + up*i  (all taps when up == 1)
    int ky0 = (a.pad_y - py * a.stride) & (up - 1);
    int kx0 = (a.pad_x - px * a.stride) & (up - 1);
    const int nky = (ky0 < a.KH) ? ((a.KH - ky0 + up - 1) >> a.up_shift) : 0;
    const int nkx = (kx0 < a.KW) ? ((a.KW - kx0 + up - 1) >> a.up_shift) : 0;
    const int chunks = nky * nkx * a.cpt;
    const int c_begin = (int)(((long long)split * chunks) / nsplit);
    const int c_end = (int)(((long long)(split + 1) * chunks) / nsplit);

    // ---- per-row bookkeeping ----
    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1, nn = 0;
        if (m < Mcls) {
            nn = m / (QH * QW);
            const int r = m - nn * (QH * QW);
            const int qy = r / QW, qx = r - qy * QW;
            pix = (nn * a.OH + (qy * up + py)) * a.OW + (qx * up + px);
        }
        row_pix[tid] = pix;
        row_n[tid] = nn;
        row_nz[tid] = noise_term(a, pix);
    }
    // loader rows (registers)
    const int kvec = tid & 7;
    const int arow0 = tid >> 3;
    int rn[AR], rby[AR], rbx[AR];
    bool rok[AR];
#pragma unroll
    for (int i = 0; i < AR; i++) {
        const int m = m0 + arow0 + AROWS * i;
        rok[i] = m < Mcls;
        const int mm = rok[i] ? m : 0;
        const int nn = mm / (QH * QW);
        const int r = mm - nn * (QH * QW);
        const int qy = r / QW, qx = r - qy * QW;
        rn[i] = nn;
        rby[i] = (qy * up + py) * a.stride - a.pad_y;
        rbx[i] = (qx * up + px) * a.stride - a.pad_x;
    }
    // B loader coordinates
    constexpr int NV = BN / 4;            // float4 per B row (normal layout)
    constexpr int KROWS = NT / NV;        // k rows per pass (normal layout)
    const int nvec = WT ? 0 : (tid % NV);
    const int krow0 = WT ? 0 : (tid / NV);
    const int brow0 = tid >> 3;           // transposed layout: n row

    float4 ra[AR], rsa[AR], rb[BR];

    // (tap, channel-chunk) of the next chunk to load, advanced incrementally (no per-chunk division)
    const int ld_t0 = (c_begin < c_end) ? c_begin / a.cpt : 0;
    int ld_cc = (c_begin < c_end) ? c_begin - ld_t0 * a.cpt : 0;
    int ld_ta = (c_begin < c_end) ? ld_t0 / nkx : 0;
    int ld_tb = (c_begin < c_end) ? ld_t0 - ld_ta * nkx : 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)a.N * a.H * a.W * a.Cin * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)a.KH * a.KW * a.Cin * a.Cout * 4u);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(SC ? a.in_scale : a.x, SC ? (unsigned)a.N * a.Cin * 4u : 0u);

    // Addresses of the NEXT prefetch are computed one stage ahead, inside the MFMA phase (pure VALU
    // that the scheduler tucks into MFMA shadows); the prefetch itself is then 3*AR.. buffer loads
    // issued back to back at the top of the iteration.
    LoadAddr aa[AR], as_[AR], ab[BR];
    // Walking form of the address computation (a.walk: 16 B paths and Cin % 32 == 0, i.e. every layer of the networks but the
    // RGB ends): inside one tap consecutive chunks differ by 32 input channels = +128 B on every activation / scale row and
    // a uniform step on the filter rows, and validity (padding, ragged rows) is a property of the tap.  So the full decode --
    // integer multiplies, compares, selects: ~90 vector instructions per chunk, which compete with the MFMAs for the SIMD's
    // issue slots -- runs once per TAP (uniform branch), and a chunk costs three adds per staged row.  Past the last chunk
    // the walk runs on into addresses that are either out of range (the buffer descriptor returns 0) or valid memory; what
    // is fetched there lands in the LDS buffer nobody reads.
    unsigned offA[AR], offS[AR], offB[BR];
    bool walk_fresh = true;
    const unsigned stepB = WT ? 128u : (unsigned)(BK * a.Cout) * 4u;
    auto decode_tap = [&]() {
        const int ky = ky0 + (ld_ta << a.up_shift), kx = kx0 + (ld_tb << a.up_shift);
        const int ci = ld_cc * BK + 4 * kvec;
#pragma unroll
        for (int i = 0; i < AR; i++) {
            const int vy = rby[i] + ky, vx = rbx[i] + kx;
            const int iy = vy >> a.up_shift, ix = vx >> a.up_shift;
            const bool ok = rok[i] & (vy >= 0) & (vx >= 0) & (iy < a.H) & (ix < a.W);
            offA[i] = ok ? (unsigned)(((rn[i] * a.H + iy) * a.W + ix) * a.Cin + ci) * 4u : OOB;
            if constexpr (SC) offS[i] = ok ? (unsigned)(rn[i] * a.Cin + ci) * 4u : OOB;
        }
        if constexpr (!WT) {
#pragma unroll
            for (int i = 0; i < BR; i++) {
                const int cik = ld_cc * BK + krow0 + KROWS * i;
                const int co = n0 + 4 * nvec;
                offB[i] = (co < a.Cout) ? (unsigned)(((ky * a.KW + kx) * a.Cin + cik) * a.Cout + co) * 4u : OOB;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BR; i++) {
                const int co = n0 + brow0 + AROWS * i;
                offB[i] = (co < a.Cout) ? (unsigned)((((a.KH - 1 - ky) * a.KW + (a.KW - 1 - kx)) * a.Cout + co) * a.Cin + ci) * 4u : OOB;
            }
        }
    };
    auto prep_chunk = [&](bool live) {
        if constexpr (VEC) {
            if (a.walk) {
                if (walk_fresh | (ld_cc == 0)) decode_tap();     // uniform
                walk_fresh = false;
#pragma unroll
                for (int i = 0; i < AR; i++) {
                    aa[i].off = offA[i]; aa[i].left = 0;
                    offA[i] += 128u;                             // OOB stays out of range: 0x7FFFFFF0 + cpt * 128 < 2^32
                    if constexpr (SC) { as_[i].off = offS[i]; as_[i].left = 0; offS[i] += 128u; }
                }
#pragma unroll
                for (int i = 0; i < BR; i++) { ab[i].off = offB[i]; ab[i].left = 0; offB[i] += stepB; }
                ++ld_cc;
                const int w1 = (ld_cc == a.cpt) ? 1 : 0;
                ld_cc = w1 ? 0 : ld_cc;
                ld_tb += w1;
                const int w2 = (ld_tb == nkx) ? 1 : 0;
                ld_tb = w2 ? 0 : ld_tb;
                ld_ta += w2;
                return;
            }
        }
        const int ci0 = ld_cc * BK;
        const int ky = ky0 + (ld_ta << a.up_shift), kx = kx0 + (ld_tb << a.up_shift);
        const int ci = ci0 + 4 * kvec;
#pragma unroll
        for (int i = 0; i < AR; i++) {
            const int vy = rby[i] + ky, vx = rbx[i] + kx;
            const int iy = vy >> a.up_shift, ix = vx >> a.up_shift;
            const bool ok = live & rok[i] & (vy >= 0) & (vx >= 0) & (iy < a.H) & (ix < a.W);
            const int left = ok ? (a.Cin - ci) : 0;
            aa[i] = make_addr<VEC>(((rn[i] * a.H + iy) * a.W + ix) * a.Cin + ci, left);
            if constexpr (SC) as_[i] = make_addr<VEC>(rn[i] * a.Cin + ci, left);
        }
        if constexpr (!WT) {
#pragma unroll
            for (int i = 0; i < BR; i++) {
                const int kr = krow0 + KROWS * i;
                const int cik = ci0 + kr;
                const int co = n0 + 4 * nvec;
                ab[i] = make_addr<VEC>(((ky * a.KW + kx) * a.Cin + cik) * a.Cout + co, (live & (cik < a.Cin)) ? (a.Cout - co) : 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < BR; i++) {
                const int co = n0 + brow0 + AROWS * i;
                ab[i] = make_addr<VEC>((((a.KH - 1 - ky) * a.KW + (a.KW - 1 - kx)) * a.Cout + co) * a.Cin + ci, (live & (co < a.Cout)) ? (a.Cin - ci) : 0);
            }
        }
        // branch-free advance (keeps the loop body one basic block)
        ++ld_cc;
        const int w1 = (ld_cc == a.cpt) ? 1 : 0;
        ld_cc = w1 ? 0 : ld_cc;
        ld_tb += w1;
        const int w2 = (ld_tb == nkx) ? 1 : 0;
        ld_tb = w2 ? 0 : ld_tb;
        ld_ta += w2;
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < AR; i++) {
            ra[i] = load4<VEC>(rx, aa[i]);
            if constexpr (SC) rsa[i] = load4<VEC>(rs, as_[i]);   // multiplied in at store time (after the MFMAs)
        }
#pragma unroll
        for (int i = 0; i < BR; i++) rb[i] = load4<VEC>(rw, ab[i]);
    };
    auto store_chunk = [&](int buf) {
        float* A = As + buf * A_ELEMS;
        float* B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int i = 0; i < AR; i++)
            *reinterpret_cast<float4*>(A + (arow0 + AROWS * i) * LDK + 4 * kvec) = SC ? f4mul(ra[i], rsa[i]) : ra[i];
        if constexpr (!WT) {
#pragma unroll
            for (int i = 0; i < BR; i++) *reinterpret_cast<float4*>(B + (krow0 + KROWS * i) * LDB + 4 * nvec) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < BR; i++) *reinterpret_cast<float4*>(B + (brow0 + AROWS * i) * LDK + 4 * kvec) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.0f;

    if (c_begin < c_end) {
        prep_chunk(true);
        issue_loads();
        store_chunk(0);
        prep_chunk(c_begin + 1 < c_end);
    }
    __syncthreads();
    stamp(1);
#ifdef IGAN_YOUNG_PRIO
    // The second-dispatched half of an 8-wave workgroup (waves 4-7: one per SIMD, beside waves 0-3) loses every issue
    // arbitration to its older partner, arrives late at each chunk's barrier and makes the older half wait there (measured per
    // chunk: older waves 1370 cycles at the barrier, younger ones 270).  A static priority for the younger half evens it out.
    if (WM * WN == 8 && wave >= 4) __builtin_amdgcn_s_setprio(IGAN_YOUNG_PRIO);
#endif
#ifdef IGAN_LOOP_STAMPS
    // Diagnostic variant (make variant DEFS=-DIGAN_LOOP_STAMPS): per wave, the shader cycles spent in the four phases of a
    // chunk, summed over the tile's chunks: [top .. first MFMA half issued] [.. staged chunk written to LDS] [.. second MFMA
    // half issued] [.. barrier passed].  Written to diag[4 * grid + (block * 8 + wave) * 4 + k] (tools/conv_phases.py --loop).
    unsigned long long ph[4] = {0, 0, 0, 0};
#define LOOP_STAMP(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define LOOP_STAMP(var)
#endif
    for (int c = c_begin; c < c_end; c++) {
        const int cur = (c - c_begin) & 1;
        LOOP_STAMP(ts0)
        // Prefetch of chunk c+1 (predicated off -- every lane out of range, no memory traffic -- on the
        // last iteration, instead of branched around: the loop body stays ONE basic block).
        issue_loads();
        __builtin_amdgcn_sched_barrier(0);   // the prefetch stays AHEAD of the MFMAs ...
        float af[TM][16], bf[TN][16];
        load_frag<TM, false, LDK>(As + cur * A_ELEMS, wm * (BM / WM), l31, h, af);
        load_frag<TN, !WT, LDB>(Bs + cur * B_ELEMS, wn * (BN / WN), l31, h, bf);
        mma_steps<TM, TN, 0, IGAN_SPLIT>(af, bf, acc);
        LOOP_STAMP(ts1)
        prep_chunk(c + 2 < c_end);           // addresses of chunk c+2: VALU only, free to interleave with the MFMAs
        // ... and its consumers (with their s_waitcnt vmcnt) stay BEHIND the first half of the MFMAs:
        // without this fence hipcc hoists the first scale-multiply + ds_write up to the first MFMA and
        // stalls there.  Nobody reads buffer cur^1 during this iteration (every wave passed the
        // barrier below after its last read of it), so its stores need no phase of their own: they
        // interleave with the second half of the MFMAs, whose issue slots they fill.
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(cur ^ 1);
        LOOP_STAMP(ts2)
        mma_steps<TM, TN, IGAN_SPLIT, 16>(af, bf, acc);
        LOOP_STAMP(ts3)
        __syncthreads();
#ifdef IGAN_LOOP_STAMPS
        const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
        ph[0] += ts1 - ts0; ph[1] += ts2 - ts1; ph[2] += ts3 - ts2; ph[3] += ts4 - ts3;
#endif
    }
#ifdef IGAN_LOOP_STAMPS
    if (a.diag != nullptr && lane == 0 && wave < 8)
        for (int k = 0; k < 4; k++) a.diag[(size_t)gridDim.x * 4 + ((size_t)blockIdx.x * 8 + wave) * 4 + k] = ph[k];
#endif

    stamp(2);
    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*h; tile -> column via tile_row ----
    if (sliced && nsplit > 1) {
        // partial tile, raw, into its [BM][BN] slot of the workspace (a.y): slot = tail tile * splits + slice
        float* wst = a.y + ((size_t)(tile - a.full_tiles) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm * (BM / WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* wr = wst + row * BN + wn * (BN / WN);
                if constexpr (!WT && TN == 2) {
                    *reinterpret_cast<float2*>(wr + 2 * l31) = make_float2(acc[tm][0][r], acc[tm][1][r]);
                } else {
#pragma unroll
                    for (int tn = 0; tn < TN; tn++) wr[tile_row<TN, !WT>(tn, l31)] = acc[tm][tn][r];
                }
            }
        return;
    }
    float* out = a.out;
    const bool scale = a.out_scale != nullptr;
    const float alpha = a.alpha;
    const int cbase = n0 + wn * (BN / WN);
    // Per-lane constants of the epilogue: the lane's output channels are fixed, so bias (and alpha) are loaded once; the
    // per-(sample, channel) output scale is loaded once too when the whole tile lies in one sample (every layer from 16x16
    // up: 128 rows <= H*W), otherwise per row.  (One dependent L2 round trip per row made the epilogue 7-20 us per tile.)
    int cos[TN];
    float mul[TN], bia[TN];
    const int n_first = row_n[0], n_last = row_n[min(BM, Mcls - m0) - 1];
    const bool one_sample = n_first == n_last;
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        cos[tn] = cbase + tile_row<TN, !WT>(tn, l31);
        const bool in = cos[tn] < a.Cout;
        mul[tn] = (scale && one_sample && in) ? a.out_scale[n_first * a.Cout + cos[tn]] : 1.0f;
        bia[tn] = (a.act && a.bias && in) ? a.bias[cos[tn]] : 0.0f;
    }
    const bool row_scale = scale && !one_sample;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = wm * (BM / WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pix = row_pix[row];
            if (pix < 0) continue;
            float v[TN];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                v[tn] = acc[tm][tn][r] * alpha;
                if (scale && one_sample) v[tn] *= mul[tn];       // same product order as the per-row form and the fix-up kernel
                if (row_scale && cos[tn] < a.Cout) v[tn] *= a.out_scale[row_n[row] * a.Cout + cos[tn]];
                if (a.act) v[tn] = epi_act(a.act, v[tn] + row_nz[row] + bia[tn], a.act_alpha) * a.act_gain;
            }
            if constexpr (!WT && TN == 2) {
                if (a.vecY) {   // the lane's two tiles are adjacent channels: one 8 B store
                    if (cos[0] < a.Cout) *reinterpret_cast<float2*>(out + (size_t)pix * a.Cout + cos[0]) = make_float2(v[0], v[1]);
                    continue;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
                if (cos[tn] < a.Cout) out[(size_t)pix * a.Cout + cos[tn]] = v[tn];
        }
    }
    stamp(3);
}

// ------------------------------------------------------------------------------
// m / d for 0 <= m < 2^24, d >= 1, with inv = 1.0f / d: the float quotient is within one of the true one (a handful of
// instructions instead of the ~30 of the generic 32-bit division; a tile's prologue has eight of them per lane).
__device__ __forceinline__ int div_small(int m, int d, float inv) {
    int q = (int)((float)m * inv);
    int r = m - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
    q += (r >= d) ? 1 : 0;          // the estimate is within one; the second step is margin
    return q;
}

// Forward-type kernel, LDS-DMA form (128x128x32 tile, 8 waves = 2 x 4 of 64x32).  Same tile list, same MFMA order per
// accumulator and therefore the same results bit for bit as conv_fwd_kernel<128,128,2,4>; what changes is how a chunk gets
// into LDS.  In the register-staged kernel every wave spends 600-1250 cycles per chunk between its two MFMA halves waiting
// for its global loads and writing them to LDS (tools/conv_phases.py --loop), a window in which it issues no MFMA.  Here the
// chunk is fetched by `buffer_load_dwordx4 ... lds`: the data goes from L2 straight into the LDS stage, needs no staging
// registers, no scale multiply at store time, no ds_write and no mid-chunk wait -- a wave's chunk is
//     [4 DMA instructions for chunk c+1] [fragment reads of chunk c] [32 MFMAs] [vmcnt(0) + barrier].
//  * An LDS-DMA instruction writes 64 lanes x 16 B to ONE contiguous KiB (wave-uniform base + lane * 16), so the stage images
//    are unpadded: A [128 rows][32 k] and (transposed weights) B [128 n][32 k] store the 16 B k-segment q of row r at slot
//    q ^ ((r >> 1) & 7) -- each lane simply fetches the segment that belongs in its slot -- which makes the fragment reads
//    (ds_read_b128, 16 lanes per LDS cycle) conflict-free; B [32 k][128 n] needs no swizzle (ds_read_b32 rows).
//  * Out-of-range lanes (padding taps, ragged rows) use an out-of-range buffer offset: the DMA writes zeros for them.
//  * The modulation scale is applied to the A fragments after the LDS read (the products x*s are the same single roundings as
//    in the staged kernel); the scale rows of the samples the tile touches sit in LDS (at most 2048 floats: the host checks),
//    and 16 B paths with Cin % 32 == 0 (`walk`) are required.
//  * PIECES = 2 / 3 (measure-only variants behind IGAN_CONV_BF16X3=1 / =6, never the default): the fp32 fragments are split in
//    registers into bf16 pieces, x = x0 + x1 (+ x2) with x0 = bf16(x), x1 = bf16(x - x0), ..., and a product runs as the three
//    (a1 b0 + a0 b1 + a0 b0) or six (all a_i b_j, i + j <= 2) v_mfma_f32_32x32x16_bf16 with fp32 accumulation instead of eight
//    v_mfma_f32_32x32x2_f32: 12 / 24 matrix instructions of 32 cycles per chunk and wave instead of 32 of 64.  Not fp32-exact
//    (two pieces: ~7x the rounding error of the fp32 FMA chain; three: about the chain's); the accumulator layout, and with it the
//    whole epilogue, is the same.  A lane's eight k values of a 16-deep step are quarters (2s, 2s+1) of its half.
template <bool WT, bool SC, int PIECES = 0>      // PIECES: 0 = fp32 MFMA (the product path), 2 / 3 = bf16 pieces per operand (measure-only)
__global__ __launch_bounds__(512, 4) void conv_fwd_dma_kernel(ConvArgs a) {
    constexpr bool SPLIT = PIECES != 0;
    constexpr int BM = 128, BN = 128, WN = 4, TM = 2, TN = 1;
    constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN;          // floats per stage (16 KiB each)
    constexpr int SMAX = 2048;                                   // scale rows of the tile's samples: (samples per tile) * Cin <= SMAX, checked by the host
    // one LDS object: [A0 A1 B0 B1 | scale row | row_pix row_n]
    __shared__ __attribute__((aligned(1024))) float smem[2 * A_STAGE + 2 * B_STAGE + SMAX + 3 * BM];
    float* As = smem;
    float* Bs = smem + 2 * A_STAGE;
    float* s_tab = smem + 2 * A_STAGE + 2 * B_STAGE;
    int* row_pix = reinterpret_cast<int*>(s_tab + SMAX);
    int* row_n = row_pix + BM;
    float* row_nz = reinterpret_cast<float*>(row_n + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int up = 1 << a.up_shift;
    auto stamp = [&](int k) {       // diagnostic only (a.diag == nullptr in every normal launch), as in conv_fwd_kernel
        if (a.diag != nullptr && (threadIdx.x >> 6) == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if ((threadIdx.x & 63) == 0) a.diag[(size_t)blockIdx.x * 4 + k] = t;
        }
    };
    stamp(0);
    // The prologue of a tile (a few hundred scalar / vector instructions) runs beside the partner workgroup's MFMA stream, and at
    // equal priority it takes 18 us instead of 3: let it win the issue arbitration until the first chunk is on its way.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int bid = blockIdx.x;
    if (a.xcd_remap && bid < a.full_tiles) {
        const int per_class = a.nx * a.ny;
        const int lo = (bid / per_class) * per_class;
        bid = lo + remap_xcd(bid - lo, min(per_class, a.full_tiles - lo));
    }
    const bool sliced = bid >= a.full_tiles;
    const int tail = bid - a.full_tiles;
    const int tile = sliced ? a.full_tiles + tail / a.splits : bid;
    const int split = sliced ? tail % a.splits : 0;
    const int nsplit = sliced ? a.splits : 1;
    const int mt = tile % a.nx, nt = (tile / a.nx) % a.ny, cls = tile / (a.nx * a.ny);
    const int py = cls >> a.up_shift, px = cls & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Mcls = a.N * QH * QW;
    const int m0 = mt * BM;
    if (m0 >= Mcls) return;
    const int n0 = nt * BN;
    const int ky0 = (a.pad_y - py * a.stride) & (up - 1);
    const int kx0 = (a.pad_x - px * a.stride) & (up - 1);
    const int nky = (ky0 < a.KH) ? ((a.KH - ky0 + up - 1) >> a.up_shift) : 0;
    const int nkx = (kx0 < a.KW) ? ((a.KW - kx0 + up - 1) >> a.up_shift) : 0;
    const int chunks = nky * nkx * a.cpt;
    const int c_begin = sliced ? (split * chunks) / nsplit : 0;           // split * chunks < 2^31: splits <= 256, chunks < 2^20
    const int c_end = sliced ? ((split + 1) * chunks) / nsplit : chunks;

    const float inv_hw = 1.0f / (float)(QH * QW), inv_w = 1.0f / (float)QW;     // Mcls < 2^24 (checked by the host: 2 GiB operands)
    const int n_lo = div_small(m0, QH * QW, inv_hw);                 // first sample of the tile
    // ---- DMA lane geometry.  A (and transposed B): wave w fills rows [8w, 8w+8) and [64 + 8w, ...): lane -> row = lane >> 3,
    // LDS slot = lane & 7, fetched k-segment = slot ^ ((row >> 1) & 7).
    int rn[2], rby[2], rbx[2];
    bool rok[2];
    int aseg[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = wave * 8 + (lane >> 3) + 64 * i;
        const int m = m0 + row;
        rok[i] = m < Mcls;
        const int mm = rok[i] ? m : 0;
        const int nn = div_small(mm, QH * QW, inv_hw);
        const int r = mm - nn * (QH * QW);
        const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
        rn[i] = nn;
        rby[i] = (qy * up + py) * a.stride - a.pad_y;
        rbx[i] = (qx * up + px) * a.stride - a.pad_x;
        aseg[i] = (lane & 7) ^ ((row >> 1) & 7);
    }
    int ld_t0 = (c_begin < c_end) ? c_begin / a.cpt : 0;
    int ld_cc = (c_begin < c_end) ? c_begin - ld_t0 * a.cpt : 0;
    int ld_ta = (c_begin < c_end) ? ld_t0 / nkx : 0;
    int ld_tb = (c_begin < c_end) ? ld_t0 - ld_ta * nkx : 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)a.N * a.H * a.W * a.Cin * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)a.KH * a.KW * a.Cin * a.Cout * 4u);
    unsigned offA[2], offB[2];
    bool fresh = true;
    const unsigned stepB = WT ? 128u : (unsigned)(BK * a.Cout) * 4u;
    auto decode_tap = [&]() {
        const int ky = ky0 + (ld_ta << a.up_shift), kx = kx0 + (ld_tb << a.up_shift);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int vy = rby[i] + ky, vx = rbx[i] + kx;
            const int iy = vy >> a.up_shift, ix = vx >> a.up_shift;
            const bool ok = rok[i] & (vy >= 0) & (vx >= 0) & (iy < a.H) & (ix < a.W);
            offA[i] = ok ? (unsigned)(((rn[i] * a.H + iy) * a.W + ix) * a.Cin + ld_cc * BK + 4 * aseg[i]) * 4u : OOB;
        }
        if constexpr (!WT) {
#pragma unroll
            for (int i = 0; i < 2; i++) {   // wave w fills k rows 2w, 2w+1 (+16 i): lane -> k row = lane >> 5, 4 columns at 4 * (lane & 31)
                const int cik = ld_cc * BK + wave * 2 + (lane >> 5) + 16 * i;
                const int co = n0 + 4 * (lane & 31);
                offB[i] = (co < a.Cout) ? (unsigned)(((ky * a.KW + kx) * a.Cin + cik) * a.Cout + co) * 4u : OOB;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++) {   // like A: n rows, swizzled k segments
                const int co = n0 + wave * 8 + (lane >> 3) + 64 * i;
                offB[i] = (co < a.Cout) ? (unsigned)((((a.KH - 1 - ky) * a.KW + (a.KW - 1 - kx)) * a.Cout + co) * a.Cin + ld_cc * BK + 4 * aseg[i]) * 4u : OOB;
            }
        }
    };
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_chunk = [&](int stage) {       // 4 wave instructions: 2 KiB of A, 2 KiB of B for this wave
        if (fresh | (ld_cc == 0)) decode_tap();
        fresh = false;
        float* A = As + stage * A_STAGE + wave * 8 * BK;
        float* B = Bs + stage * B_STAGE + (WT ? wave * 8 * BK : wave * 2 * BN);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)A, 16, offA[0], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 64 * BK), 16, offA[1], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)B, 16, offB[0], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(B + (WT ? 64 * BK : 16 * BN)), 16, offB[1], 0, 0, 0);
        offA[0] += 128u; offA[1] += 128u; offB[0] += stepB; offB[1] += stepB;
        ++ld_cc;
        const int w1 = (ld_cc == a.cpt) ? 1 : 0;
        ld_cc = w1 ? 0 : ld_cc;
        ld_tb += w1;
        const int w2 = (ld_tb == nkx) ? 1 : 0;
        ld_tb = w2 ? 0 : ld_tb;
        ld_ta += w2;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][0][r] = 0.0f;

    // Fragments in quarters of the chunk (k steps 4q .. 4q+3 of this lane's half: one 16 B read per A tile, four B values, one
    // 16 B read of the scale row), so that the reads of a chunk can straddle the barrier that precedes it:
    //     [DMA chunk c+1] [read Q2, Q3 of c] [MFMA Q0 Q1 Q2 of c] [vmcnt(0) + barrier] [read Q0, Q1 of c+1] [MFMA Q3 of c]
    // the 8 MFMAs of Q3 cover the LDS latency of the next chunk's first reads, and the DMA has 24 MFMA slots to land.
    float afq[4][TM][4], bfq[4][4];
    int srow[TM];        // offset of this lane's rows' scale rows in s_tab (rows of one tile can belong to different samples); set below
    // Where the modulation multiply happens.  A tile whose 128 rows belong to ONE sample (every layer from 16x16 up) can scale the
    // B fragments by s[k] instead of the A fragments: sum_k (x[m,k] s[k]) w[k,n] = sum_k x[m,k] (s[k] w[k,n]) -- 16 multiplies and
    // 4 scale reads per lane and chunk instead of 32 and 8 (the B fragment is half the size of the two A fragments).  Tiles that
    // straddle samples (8x8 and below) keep the row-wise A form.  Either way each product is one fp32 rounding before the MFMA.
    const bool b_scaled = SC && (n_lo == div_small(min(m0 + BM, Mcls) - 1, QH * QW, inv_hw));
    auto read_q = [&](auto bs_tag, int stage, int q, int ci_chunk) {
        constexpr bool BS = decltype(bs_tag)::value;
        const float* A = As + stage * A_STAGE;
        const float* B = Bs + stage * B_STAGE;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
            const int row = wm * 64 + tm * 32 + l31;
            float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (SC && !BS) sv = *reinterpret_cast<const float4*>(s_tab + srow[tm] + ci_chunk * BK + 4 * q);
            const float4 v = *reinterpret_cast<const float4*>(A + row * BK + (((4 * h + q) ^ ((row >> 1) & 7)) << 2));
            afq[q][tm][0] = (SC && !BS) ? v.x * sv.x : v.x; afq[q][tm][1] = (SC && !BS) ? v.y * sv.y : v.y;
            afq[q][tm][2] = (SC && !BS) ? v.z * sv.z : v.z; afq[q][tm][3] = (SC && !BS) ? v.w * sv.w : v.w;
        }
        if constexpr (!WT) {
#pragma unroll
            for (int j = 0; j < 4; j++) bfq[q][j] = B[(16 * h + 4 * q + j) * BN + wn * 32 + l31];
        } else {
            const int row = wn * 32 + l31;
            const float4 v = *reinterpret_cast<const float4*>(B + row * BK + (((4 * h + q) ^ ((row >> 1) & 7)) << 2));
            bfq[q][0] = v.x; bfq[q][1] = v.y; bfq[q][2] = v.z; bfq[q][3] = v.w;
        }
        if constexpr (SC && BS) {
            const float4 sv = *reinterpret_cast<const float4*>(s_tab + 16 * h + ci_chunk * BK + 4 * q);
            bfq[q][0] *= sv.x; bfq[q][1] *= sv.y; bfq[q][2] *= sv.z; bfq[q][3] *= sv.w;
        }
    };
    auto mma_q = [&](int q) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
                acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afq[q][tm][j], bfq[q][j], acc[tm][0], 0, 0, 0);
    };
    auto next_ci = [&](int ci) { return (ci + 1 == a.cpt) ? 0 : ci + 1; };
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    constexpr int NP = SPLIT ? PIECES : 1;
    auto split8 = [&](const float (&lo4)[4], const float (&hi4)[4], bf16x8 (&pc)[NP]) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float r = j < 4 ? lo4[j] : hi4[j - 4];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const __bf16 b = (__bf16)r;              // v_cvt_pk_bf16_f32: round to nearest even
                pc[p][j] = b;
                if (p + 1 < NP) r -= (float)b;
            }
        }
    };
    auto mma_split_step = [&](int s) {       // k step s of the chunk: quarters 2s, 2s+1
        bf16x8 b[NP];
        split8(bfq[2 * s], bfq[2 * s + 1], b);
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
            bf16x8 av[NP];
            split8(afq[2 * s][tm], afq[2 * s + 1][tm], av);
            // all products a_i b_j with i + j < PIECES, smallest first
#ifdef IGAN_SPLIT_PROMOTE      // experiment: the step's products are summed from zero and added to the running sum by the vector ALU (RN)
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = 0.0f;
#pragma unroll
            for (int oo = 0; oo < NP; oo++) {
                const int o = (IGAN_SPLIT_PROMOTE == 2) ? oo : NP - 1 - oo;       // 2: largest products first (the addend is never the small term)
#pragma unroll
                for (int i = 0; i <= o; i++)
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], b[o - i], t, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][0][r] += t[r];
#else
#pragma unroll
            for (int o = NP - 1; o >= 0; o--)
#pragma unroll
                for (int i = 0; i <= o; i++)
                    acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], b[o - i], acc[tm][0], 0, 0, 0);
#endif
        }
    };

#ifdef IGAN_PROLOGUE_STAMPS      // diagnostic build: slot 1 = tables computed / first DMA about to be issued, slot 2 = first barrier passed
    stamp(1);
    unsigned long long pw[4];
    pw[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (c_begin < c_end) dma_chunk(0);
    // Everything the first chunk's DMA does not need is computed while it is in flight: the epilogue's row tables and the scale
    // rows (the prologue of a tile runs beside the partner workgroup's MFMA stream and is several times slower than on an idle CU).
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
        const int m = min(m0 + wm * 64 + tm * 32 + l31, Mcls - 1);
        srow[tm] = SC ? (div_small(m, QH * QW, inv_hw) - n_lo) * a.Cin + 16 * h : 0;
    }
    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1, nn = 0;
        if (m < Mcls) {
            nn = div_small(m, QH * QW, inv_hw);
            const int r = m - nn * (QH * QW);
            const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
            pix = (nn * a.OH + (qy * up + py)) * a.OW + (qx * up + px);
        }
        row_pix[tid] = pix;
        row_n[tid] = nn;
        row_nz[tid] = noise_term(a, pix);
    }
    if constexpr (SC) {   // scale rows of the samples this tile touches (consecutive; their number is bounded by the host)
        const int n_hi = div_small(min(m0 + BM, Mcls) - 1, QH * QW, inv_hw);
        const int cnt = (n_hi - n_lo + 1) * a.Cin;
        for (int i = tid; i < cnt; i += 512) s_tab[i] = a.in_scale[n_lo * a.Cin + i];
    }

#ifdef IGAN_PROLOGUE_STAMPS
    pw[1] = __builtin_amdgcn_s_memrealtime();            // tables done, loads issued
    __builtin_amdgcn_s_waitcnt(0);
    pw[2] = __builtin_amdgcn_s_memrealtime();            // this wave's loads landed
#endif
    __syncthreads();                         // vmcnt(0): chunk c_begin has landed; the scale row and row tables are visible
    if (a.prio) __builtin_amdgcn_s_setprio(0);
#ifdef IGAN_PROLOGUE_STAMPS
    pw[3] = __builtin_amdgcn_s_memrealtime();            // barrier passed
    if (a.diag != nullptr && lane == 0 && wave < 8)
        for (int k = 0; k < 4; k++) a.diag[(size_t)gridDim.x * 4 + ((size_t)blockIdx.x * 8 + wave) * 4 + k] = pw[k];
    stamp(2);
#else
    stamp(1);
#endif
    auto main_loop = [&](auto bs_tag) {
        int sc_ci = c_begin - (c_begin / a.cpt) * a.cpt;   // channel chunk (within the tap) of the chunk being computed
        read_q(bs_tag, 0, 0, sc_ci);
        read_q(bs_tag, 0, 1, sc_ci);
        for (int c = c_begin; c < c_end; c++) {
            const int cur = (c - c_begin) & 1;
            dma_chunk(cur ^ 1);                  // chunk c+1 (past the end: out-of-range or harmless, lands in the idle stage)
            __builtin_amdgcn_sched_barrier(0);
            read_q(bs_tag, cur, 2, sc_ci);
            read_q(bs_tag, cur, 3, sc_ci);
            if constexpr (SPLIT) {
                mma_split_step(0);
            } else {
                mma_q(0);
                mma_q(1);
                mma_q(2);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                     // vmcnt(0) lgkmcnt(0) + barrier: chunk c+1 landed, everyone done reading stage `cur`
            __builtin_amdgcn_sched_barrier(0);
            sc_ci = next_ci(sc_ci);
            if constexpr (SPLIT) {
                mma_split_step(1);               // quarters 2, 3 (read before the barrier) while the next chunk's first reads land
                read_q(bs_tag, cur ^ 1, 0, sc_ci);
                read_q(bs_tag, cur ^ 1, 1, sc_ci);
            } else {
                read_q(bs_tag, cur ^ 1, 0, sc_ci);
                read_q(bs_tag, cur ^ 1, 1, sc_ci);
                mma_q(3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (SC && b_scaled && a.b_scale) main_loop(std::true_type{});
    else main_loop(std::false_type{});

#ifndef IGAN_PROLOGUE_STAMPS
    stamp(2);
#endif
    // ---- epilogue (as conv_fwd_kernel) ----
    if (sliced && nsplit > 1) {
        float* wst = a.y + ((size_t)(tile - a.full_tiles) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                wst[row * BN + wn * 32 + l31] = acc[tm][0][r];
            }
        return;
    }
    float* out = a.out;
    const bool scale = a.out_scale != nullptr;
    const float alpha = a.alpha;
    const int co = n0 + wn * 32 + l31;
    const bool in = co < a.Cout;
    const int n_first = row_n[0], n_last = row_n[min(BM, Mcls - m0) - 1];
    const bool one_sample = n_first == n_last;
    const float mul = (scale && one_sample && in) ? a.out_scale[n_first * a.Cout + co] : 1.0f;
    const float bia = (a.act && a.bias && in) ? a.bias[co] : 0.0f;
    const bool row_scale = scale && !one_sample;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pix = row_pix[row];
            if (pix < 0 || !in) continue;
            float v = acc[tm][0][r] * alpha;
            if (scale && one_sample) v *= mul;
            if (row_scale) v *= a.out_scale[row_n[row] * a.Cout + co];
            if (a.act) v = epi_act(a.act, v + row_nz[row] + bia, a.act_alpha) * a.act_gain;
            out[(size_t)pix * a.Cout + co] = v;
        }
    }
    stamp(3);
}

// Fix-up of the sliced tail tiles: y[tile] = alpha * out_scale * sum_slices ws[tile][slice]  (fixed order).
// grid = (tail tiles, row groups of RP rows); the partial tiles are dense [BM][BN] images, rows map to
// output pixels exactly as in the main kernel.
__global__ __launch_bounds__(256) void conv_fixup_kernel(ConvArgs a, int BM, int BN, int RP) {
    __shared__ int row_pix[128];
    __shared__ int row_n[128];
    __shared__ float row_nz[128];
    const int tid = threadIdx.x;
    const int up = 1 << a.up_shift;
    const int tile = a.full_tiles + blockIdx.x;
    const int mt = tile % a.nx, nt = (tile / a.nx) % a.ny, cls = tile / (a.nx * a.ny);
    const int py = cls >> a.up_shift, px = cls & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Mcls = a.N * QH * QW;
    const int r0 = blockIdx.y * RP;
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 + r0 >= Mcls) return;
    if (tid < RP) {
        const int m = m0 + r0 + tid;
        int pix = -1, nn = 0;
        if (r0 + tid < BM && m < Mcls) {      // the last row group may overhang the tile
            nn = m / (QH * QW);
            const int r = m - nn * (QH * QW);
            const int qy = r / QW, qx = r - qy * QW;
            pix = (nn * a.OH + (qy * up + py)) * a.OW + (qx * up + px);
        }
        row_pix[tid] = pix;
        row_n[tid] = nn;
        row_nz[tid] = noise_term(a, pix);
    }
    __syncthreads();
    const float* wst = a.y + (size_t)blockIdx.x * a.splits * (BM * BN);
    const int bn4 = BN >> 2;
    for (int i = tid; i < RP * bn4; i += 256) {
        const int rl = i / bn4, c4 = i - rl * bn4;
        const int pix = row_pix[rl];
        if (pix < 0) continue;
        const int row = r0 + rl;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 2 <= a.splits; k += 2) {     // two slices per step: their loads are in flight together (fixed order)
            const float4 v = *reinterpret_cast<const float4*>(wst + (size_t)k * (BM * BN) + row * BN + 4 * c4);
            const float4 u = *reinterpret_cast<const float4*>(wst + (size_t)(k + 1) * (BM * BN) + row * BN + 4 * c4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
        }
        if (k < a.splits) {
            const float4 v = *reinterpret_cast<const float4*>(wst + (size_t)k * (BM * BN) + row * BN + 4 * c4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
        const float v4[4] = {s.x, s.y, s.z, s.w};
        const int nn = row_n[rl];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int co = n0 + 4 * c4 + e;
            if (co < a.Cout) {
                float v = v4[e] * a.alpha;
                if (a.out_scale) v *= a.out_scale[nn * a.Cout + co];
                if (a.act) v = epi_act(a.act, v + row_nz[rl] + (a.bias ? a.bias[co] : 0.f), a.act_alpha) * a.act_gain;
                a.out[(size_t)pix * a.Cout + co] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------
// bf16-piece form of the forward-type kernel (IGAN_CONV_PLANES=1; profiles/r03_bf16_split_rounding.txt).
//
// Arithmetic.  Every fp32 operand is written ONCE as three bf16 pieces, v = v0 + v1 + v2 exactly (v0 = bf16(v), v1 = bf16(v - v0),
// v2 = v - v0 - v1; round to nearest), and a product runs as the six piece products a_i b_j, i + j <= 2, on
// v_mfma_f32_32x32x16_bf16 (12 matrix instructions of 32 cycles per 16-deep step and 64x32 wave tile instead of 16 of 64 on the
// fp32 instruction).  What the bf16 instruction does to its addend decides the order: it floors a small addend that sits next
// to a larger product (tools/mfma_round_probe.hip), so every 16-deep step starts from an exact zero, takes the LARGEST products
// first (the addend is then never the small term) and its sum t is added to the running fp32 sum by the vector ALU (round to
// nearest).  The dropped terms (a1 b2 + a2 b1 + a2 b2) are below 2^-24 of the product.  Measured against fp64 this is a third
// of the fp32 instruction's L2 error and the whole GPU parity suite passes at the fp32 path's tolerances.
//
// Data.  to_planes_kernel writes x * in_scale (the modulation: one fp32 rounding, as everywhere else) as
// [pixel][Cin/16][piece][16] bf16 (96 B per pixel and 16 channels); filter_planes_kernel writes the filter as
// [Cin/16][tap][n][piece][16] for either orientation, so the tile kernel has ONE form: both LDS images are [piece][row][2 x 16 B]
// (the two 8-deep halves of the row, the half of row r stored at position h ^ ((r >> 3) & 1): conflict-free ds_read_b128), a
// stage is 16 deep (12 KiB + 12 KiB), three stages, filled by `buffer_load_dwordx4 ... lds` two stages ahead.  A wave's step:
//     [vmcnt: chunk c landed] [barrier] [9 fragment reads] [6 MFMAs] [3 DMA instructions for chunk c+2] [6 MFMAs] [32 adds].
// The reduction runs 16-channel slice outermost, taps inside (the taps of a slice re-read the same shifted rows: L2 hits).
// Tile list, slicing, fix-up and the whole epilogue are those of conv_fwd_dma_kernel (same accumulator layout).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// The piece kernels prefetch two chunks ahead, so their last two iterations issue LDS-DMA instructions for chunks past the end of the reduction (out-of-range
// addresses: zeros, or a harmless slice) that nobody waits for.  Every wave waits for its own LDS-DMA before it leaves the main loop, so that no wave ends -- and
// no workgroup gives its LDS back -- with a write to that LDS still in flight.  (Round 5 looked at this while chasing the multi-process mismatch of round 4; it
// was NOT its cause -- the mismatch was unchanged with the drain, see profiles/r05_replay_mismatch.txt and dense_small.hip -- but an in-flight write into LDS that
// may already belong to another workgroup is not something to leave to the hardware's goodwill; the instructions return within a few hundred cycles.)
__device__ __forceinline__ void drain_lds_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int PK = 16;                          // reduction depth of a stage
constexpr int P_IMG = 3 * 128 * 32;             // bytes of one operand image: [3 pieces][128 rows][32 B]
constexpr int P_STAGE = 2 * P_IMG;              // A + B
constexpr int P_NSTAGE = 3;

__device__ __forceinline__ void split3(float v, unsigned short (&o)[3]) {
    float r = v;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const __bf16 b = (__bf16)r;             // v_cvt_pk_bf16_f32: round to nearest even
        o[q] = __builtin_bit_cast(unsigned short, b);
        r -= (float)b;
    }
}

// x [P][C] fp32 (times scale[p / HW][C] when given) -> [P][C/16][3][16] bf16.  One thread per (pixel, 16 channels): 64 B in, 96 B out.
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, const float* __restrict__ scale, unsigned short* __restrict__ out,
                                                         int total, int cpp, int C, int HW) {
    const int idx = min((int)(blockIdx.x * 256 + threadIdx.x), total - 1);      // the last block's spare threads repeat its last unit (not stored)
    const int p = idx / cpp, c = idx - p * cpp;
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)p * C + 16 * c);
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; i++) { const float4 f = src[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    if (scale != nullptr) {
        const float4* sc = reinterpret_cast<const float4*>(scale + (size_t)(p / HW) * C + 16 * c);
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = sc[i]; v[4 * i] *= f.x; v[4 * i + 1] *= f.y; v[4 * i + 2] *= f.z; v[4 * i + 3] *= f.w; }
    }
    unsigned short pc[3][16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned short o[3];
        split3(v[i], o);
        pc[0][i] = o[0]; pc[1][i] = o[1]; pc[2][i] = o[2];
    }
    // the block's 256 units are 24 KiB of contiguous output: staged through LDS so that every store instruction of a wave
    // writes one contiguous KiB (whole 128 B lines) instead of 64 pieces 96 B apart
    __shared__ uint4 stage[256 * 6];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            uint4 u;
            u.x = pc[q][8 * hh + 0] | ((unsigned)pc[q][8 * hh + 1] << 16); u.y = pc[q][8 * hh + 2] | ((unsigned)pc[q][8 * hh + 3] << 16);
            u.z = pc[q][8 * hh + 4] | ((unsigned)pc[q][8 * hh + 5] << 16); u.w = pc[q][8 * hh + 6] | ((unsigned)pc[q][8 * hh + 7] << 16);
            stage[threadIdx.x * 6 + 2 * q + hh] = u;
        }
    __syncthreads();
    const int units = min(256, total - (int)blockIdx.x * 256);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * 256 * 48);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int j = k * 256 + threadIdx.x;
        if (j < units * 6) dst[j] = stage[j];
    }
}

// filter -> [K/16][tap][n][3][16] bf16 (one 16-deep slice of one tap's 128-row tile is 12 KiB contiguous), element (tap (ky, kx), n, k):
//   !WT: w[ky][kx][k][n] (HWIO; n fastest across threads: coalesced reads)     WT: w[KH-1-ky][KW-1-kx][n][k] (k contiguous)
template <bool WT>
__global__ __launch_bounds__(256) void filter_planes_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int taps, int KW_, int Nn, int K) {
    const int cpk = K >> 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= taps * Nn * cpk) return;
    int tap, n, c;
    if constexpr (WT) { c = idx % cpk; const int r = idx / cpk; n = r % Nn; tap = r / Nn; }
    else { n = idx % Nn; const int r = idx / Nn; c = r % cpk; tap = r / cpk; }
    float v[16];
    if constexpr (WT) {
        const float4* src = reinterpret_cast<const float4*>(w + ((size_t)(taps - 1 - tap) * Nn + n) * K + 16 * c);   // both axes flipped = reversed tap index
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = src[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = w[((size_t)tap * K + 16 * c + i) * Nn + n];
    }
    unsigned short pc[3][16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned short o[3];
        split3(v[i], o);
        pc[0][i] = o[0]; pc[1][i] = o[1]; pc[2][i] = o[2];
    }
    uint4* dst = reinterpret_cast<uint4*>(out + (((size_t)c * taps + tap) * Nn + n) * 48);
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            uint4 u;
            u.x = pc[q][8 * hh + 0] | ((unsigned)pc[q][8 * hh + 1] << 16); u.y = pc[q][8 * hh + 2] | ((unsigned)pc[q][8 * hh + 3] << 16);
            u.z = pc[q][8 * hh + 4] | ((unsigned)pc[q][8 * hh + 5] << 16); u.w = pc[q][8 * hh + 6] | ((unsigned)pc[q][8 * hh + 7] << 16);
            dst[2 * q + hh] = u;
        }
}


// ------------------------------------------------------------------------------
// TWO-PIECE fp16 form of the piece kernels (IGAN_CONV_PLANES=2 or unset: the DEFAULT form; tools/piece_shape_probe.hip).
//
// Arithmetic.  fp16 carries 11 significand bits, so TWO pieces (11 + 11 + the sign of the second) hold 23 of an fp32 value's 24 significand bits always and
// all 24 in three cases of four (tests/test_fp16_pairs_arithmetic.py: exact for 74.5 % of random values, at most ONE unit in the last place = 2^-23 |v| for
// the rest, rms 4.4e-8 -- the size of one more fp32 rounding of the operand): with a power-of-two scale S (exact: an exponent shift) that brings the largest
// magnitude of the element's SCALE GROUP into [2^14, 2^15),  p0 = fp16(v S),  p1 = fp16((v S - p0) 2^11)  (|p1| <= |p0|: the second piece is stored 2^11 up
// so that it lives in fp16's normal range too), and  v S = p0 + 2^-11 p1  to 2^-23 |v S| for every element within 2^26 of its group's largest.
//
// Scale groups (round 5: NO per-tensor window any more).  A scale may vary along any axis of an operand that is NOT summed over inside one matrix
// instruction chain, because there it factors out of the chain exactly:
//   forward / data gradient   A = x in_scale: one scale per PIXEL (its whole channel vector; the "row image").  A 3x3 output row reads nine different
//                             pixels, so the scale does not factor out of the whole reduction -- it factors out of every 16-deep STEP (one tap, 16
//                             channels of one pixel per row), and the step's sum is multiplied by 1 / S_pixel when the vector ALU folds it into the
//                             running fp32 sum (one v_fma instead of one v_add).      B = the filter: one scale per OUTPUT CHANNEL (column n; the
//                             epilogue multiplies by 1 / S_n).
//   weight gradient           both operands are summed over pixels: one scale per CHANNEL of x in_scale and of dy out_scale (the "column image");
//                             the epilogue multiplies row ci by 1 / S_ci and column co by 1 / S_co.
// What is left is the window INSIDE one group, along the summed axis: an element more than 2^26 below the largest of its own pixel's channel vector
// (forward), of its own filter column, or of its own channel's pixels (weight gradient) keeps one bit less per binade -- next to a term 2^26 larger in
// the same sum, i.e. beyond what an fp32 sum of those terms resolves.  igan_debug_f16_window counts such elements.
// A product is three matrix instructions instead of six:
//     a b = (Sa Sb)^-1 [ p0a p0b + 2^-11 (p0a p1b + p1a p0b) ]        (dropped: 2^-22 p1a p1b -- at most 2^-22 |a b|, 2^-24.6 |a b| rms: one fp32 rounding of the product)
// Forward: per 16-deep step the main term and the two cross terms each start from an exact zero in the matrix pipe; the vector ALU forms
// (main + 2^-11 cross) / S_pixel and adds it to the running sum (round to nearest).  Weight gradient: main term folded per step, the cross terms chained
// in a second accumulator of the matrix pipe over the whole reduction (the scales are constant along it).
// Image layout [pixel][C/16][2][16] fp16 = 4 B per element.  Behind a row image: 1 / S per pixel (P floats).  Behind a column image: 1 / S per channel
// (C floats), S per channel (C floats), the column-maximum partials (H_COLBLOCKS x C floats).  Behind a filter image: 1 / S per output channel, the
// column-maximum partials.  No host involvement, no atomics in the data path (max is order-independent: bit-reproducible).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H_COLBLOCKS = 1024;    // partial rows of a column-maximum pass (as many blocks stream the tensor)
constexpr int H_KSK = 2;             // a filter's column maxima: partial rows per tap (HWIO orientation)
// Diagnostic (igan_debug_f16_window): how many non-zero elements were imaged BELOW the window in which the two pieces hold the value to 2^-23
// (|v S| < 2^-12, i.e. more than 2^26 below the largest magnitude of the element's scale group), and how many elements were imaged in all.
__device__ unsigned long long g_f16_below_window = 0ull, g_f16_imaged = 0ull;            // row images (forward / data gradient): the scale group is a pixel's channel vector
__device__ unsigned long long g_f16_below_window_cols = 0ull, g_f16_imaged_cols = 0ull;  // column images (weight gradient): the scale group is a channel's pixels -- the SUMMED axis

__host__ __device__ inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
// bytes of a row image of [P][C] (image + 1 / S per pixel), of a column image (image + 1 / S, S per channel + partials) and of a filter image
inline size_t rows_part_bytes(size_t P_, size_t C) { return align256(4 * P_ * C + 4 * P_); }
inline size_t cols_part_bytes(size_t P_, size_t C) { return align256(4 * P_ * C + 8 * C + 4 * (size_t)H_COLBLOCKS * C); }
inline size_t filter_part_bytes(size_t taps, size_t Nn, size_t K) { return align256(4 * taps * Nn * K + 4 * Nn + 4 * taps * H_KSK * Nn); }

// S = 2^(14 - floor(log2 amax)) as a float (amax = 0, subnormal or below 2^-112: the largest shift that keeps 1 / S normal; inf / nan: 2^-113)
__device__ __forceinline__ float scale_from_amax(float amax) {
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    e = max(15, min(e, 254));
    return __uint_as_float((unsigned)(268 - e) << 23);
}
__device__ __forceinline__ float inv_scale_from_amax(float amax) {
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    e = max(15, min(e, 254));
    return __uint_as_float((unsigned)(e - 14) << 23);
}
__device__ __forceinline__ void count_window(int below, unsigned long long* counter) {      // per wave: one atomic when anything is to be counted (diagnostic)
#ifdef IGAN_NO_WINDOW_COUNT
    return;
#endif
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o);
    if ((threadIdx.x & 63) == 0 && below != 0) atomicAdd(counter, (unsigned long long)below);
}

__device__ __forceinline__ void split2(float vs, unsigned short (&o)[2]) {        // vs = v * S
    const _Float16 p0 = (_Float16)vs;                   // round to nearest even
    const _Float16 p1 = (_Float16)((vs - (float)p0) * 2048.0f);
    o[0] = __builtin_bit_cast(unsigned short, p0);
    o[1] = __builtin_bit_cast(unsigned short, p1);
}

// the 16 values of one (pixel, 16-channel slice) unit of x [P][C] (times scale[p / HW][C] when given)
__device__ __forceinline__ void load_unit16(const float* __restrict__ x, const float* __restrict__ scale, int p, int c, int C, int HW, float (&v)[16]) {
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)p * C + 16 * c);
#pragma unroll
    for (int i = 0; i < 4; i++) { const float4 f = src[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    if (scale != nullptr) {
        const float4* sc = reinterpret_cast<const float4*>(scale + (size_t)(p / HW) * C + 16 * c);
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = sc[i]; v[4 * i] *= f.x; v[4 * i + 1] *= f.y; v[4 * i + 2] *= f.z; v[4 * i + 3] *= f.w; }
    }
}
// the block's 256 units (16 KiB of contiguous image) go through LDS so that every store instruction of a wave writes one contiguous KiB
__device__ __forceinline__ void store_units(const unsigned short (&pc)[2][16], uint4* stage, unsigned short* __restrict__ out, int total, int chunk) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            uint4 u;
            u.x = pc[q][8 * hh + 0] | ((unsigned)pc[q][8 * hh + 1] << 16); u.y = pc[q][8 * hh + 2] | ((unsigned)pc[q][8 * hh + 3] << 16);
            u.z = pc[q][8 * hh + 4] | ((unsigned)pc[q][8 * hh + 5] << 16); u.w = pc[q][8 * hh + 6] | ((unsigned)pc[q][8 * hh + 7] << 16);
            stage[threadIdx.x * 4 + 2 * q + hh] = u;
        }
    __syncthreads();
    const int units = min(256, total - chunk * 256);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)chunk * 256 * 32);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = k * 256 + threadIdx.x;
        if (j < units * 4) dst[j] = stage[j];
    }
}

// ROW image: x [P][C] fp32 (times scale[p / HW][C] when given) -> [P][C/16][2][16] fp16, one scale per pixel, 1 / S to rowinv[P].  One thread per
// (pixel, 16 channels); the cpp = C / 16 threads of a pixel (a power of two <= 64: they sit in one wave) share the pixel's maximum by shuffles.  ONE pass.
// A block takes chunks of 256 units blockIdx.x, blockIdx.x + gridDim.x, ...: its threads keep their 16-channel slice, so that -- when `colmax` is given -- the
// per-CHANNEL maxima of the block's share fall out of the same pass: colmax = [rows (as a float), 0, 0, 0][gridDim.x rows of C maxima], what the weight gradient's
// column image of the same tensor needs (cols_finalize_kernel) instead of a pass of its own over the tensor.
__global__ __launch_bounds__(256) void rows_f16_kernel(const float* __restrict__ x, const float* __restrict__ scale, unsigned short* __restrict__ out,
                                                        float* __restrict__ rowinv, float* __restrict__ colmax, int total, int cpp, int lcpp, int C, int HW) {
    __shared__ uint4 stage[256 * 4];
    const int c = threadIdx.x & (cpp - 1);
    float cm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cm[i] = 0.0f;
    int below = 0;
    const int nchunks = (total + 255) >> 8;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const bool mine = chunk * 256 + (int)threadIdx.x < total;
        const int idx = min(chunk * 256 + (int)threadIdx.x, total - 1);      // the last chunk's spare threads repeat its last unit (not stored)
        const int p = idx >> lcpp;
        float v[16];
        load_unit16(x, scale, p, c, C, HW, v);
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) { const float av = fabsf(v[i]); m = fmaxf(m, av); if (mine) cm[i] = fmaxf(cm[i], av); }
        for (int o = 1; o < cpp; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float S = scale_from_amax(m);
        if (mine && c == 0) rowinv[p] = inv_scale_from_amax(m);
        unsigned short pc[2][16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            unsigned short o[2];
            const float vs = v[i] * S;
            split2(vs, o);
            pc[0][i] = o[0]; pc[1][i] = o[1];
            below += (mine && vs != 0.0f && fabsf(vs) < 0x1p-12f) ? 1 : 0;
        }
        if (chunk != (int)blockIdx.x) __syncthreads();      // the previous chunk's copy out of `stage` is done
        store_units(pc, stage, out, total, chunk);
    }
    count_window(below, &g_f16_below_window);
#ifndef IGAN_NO_WINDOW_COUNT
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_f16_imaged, (unsigned long long)total * 16ull);
#endif
    if (colmax != nullptr) {        // wave-uniform
        __syncthreads();
        float* sm = reinterpret_cast<float*>(stage);            // [256 threads][16]
#pragma unroll
        for (int i = 0; i < 16; i++) sm[threadIdx.x * 16 + i] = cm[i];
        __syncthreads();
        const int groups = 256 >> lcpp;                          // threads that hold the same slice
        for (int ch = threadIdx.x; ch < C; ch += 256) {
            const int cc = ch >> 4, i = ch & 15;
            float m = 0.0f;
            for (int j = 0; j < groups; j++) m = fmaxf(m, sm[(j * cpp + cc) * 16 + i]);
            colmax[4 + (size_t)blockIdx.x * C + ch] = m;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) colmax[0] = (float)gridDim.x;
    }
}

// Column maxima of |x scale| over a [P][C] tensor: partial[block][C].  A thread owns one channel quad (C4 = C / 4, a power of two <= 256) and
// every (256 / C4)-th pixel of the block's share; two loads in flight per lane.
__global__ __launch_bounds__(256) void cols_amax_kernel(const float* __restrict__ x, const float* __restrict__ scale, float* __restrict__ partial,
                                                         int P_, int C4, int lC4, int HW, float* __restrict__ header) {
    if (header != nullptr && blockIdx.x == 0 && threadIdx.x == 0) header[0] = (float)gridDim.x;      // the partial rows' count travels with a caller-held buffer
    __shared__ float4 red[256];
    const int q = threadIdx.x & (C4 - 1), pl = threadIdx.x >> lC4, ppi = 256 >> lC4;
    const int step = gridDim.x * ppi;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    auto take = [&](int p) {
        float4 v = reinterpret_cast<const float4*>(x)[(size_t)p * C4 + q];
        if (scale != nullptr) {
            const float4 f = reinterpret_cast<const float4*>(scale)[(size_t)(p / HW) * C4 + q];
            v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
        }
        m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
    };
    int p = blockIdx.x * ppi + pl;
    for (; p + step < P_; p += 2 * step) { take(p); take(p + step); }
    if (p < P_) take(p);
    red[threadIdx.x] = m;
    __syncthreads();
    if (pl == 0) {
        for (int j = 1; j < ppi; j++) {
            const float4 o = red[j * C4 + q];
            m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
        }
        reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * C4 + q] = m;
    }
}
// partial[rows][C] -> inv[C] = 1 / S, sc[C] = S.  grid = C / 8 blocks of 32 row lanes x 8 channels (a thread takes every 32nd row: the
// partials are a latency chain otherwise -- four row lanes over 1024 rows took 44 us).
__global__ __launch_bounds__(256) void cols_finalize_kernel(const float* __restrict__ partial, float* __restrict__ inv, float* __restrict__ sc, int rows, int C, const float* __restrict__ header) {
    __shared__ float red[256];
    if (header != nullptr) rows = (int)header[0];       // maxima handed over by the call that wrote the tensor's row image (wave-uniform)
    const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    if (c < C) {
        int r = rl;
        for (; r + 96 < rows; r += 128) {       // four loads in flight per lane
            m0 = fmaxf(m0, partial[(size_t)r * C + c]); m1 = fmaxf(m1, partial[(size_t)(r + 32) * C + c]);
            m2 = fmaxf(m2, partial[(size_t)(r + 64) * C + c]); m3 = fmaxf(m3, partial[(size_t)(r + 96) * C + c]);
        }
        for (; r < rows; r += 32) m0 = fmaxf(m0, partial[(size_t)r * C + c]);
    }
    red[threadIdx.x] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    if (rl == 0 && c < C) {
        float m = red[threadIdx.x];
        for (int j = 1; j < 32; j++) m = fmaxf(m, red[8 * j + threadIdx.x]);
        inv[c] = inv_scale_from_amax(m);
        sc[c] = scale_from_amax(m);
    }
}
// COLUMN image: x [P][C] fp32 (times scale[p / HW][C]) -> [P][C/16][2][16] fp16 with the per-channel scales sc[C]
__global__ __launch_bounds__(256) void cols_f16_kernel(const float* __restrict__ x, const float* __restrict__ scale, unsigned short* __restrict__ out,
                                                        const float* __restrict__ sc, int total, int cpp, int C, int HW) {
    __shared__ uint4 stage[256 * 4];
    const bool mine = (int)(blockIdx.x * 256 + threadIdx.x) < total;
    const int idx = min((int)(blockIdx.x * 256 + threadIdx.x), total - 1);
    const int p = idx / cpp, c = idx - p * cpp;
    float v[16], S[16];
    load_unit16(x, scale, p, c, C, HW, v);
#pragma unroll
    for (int i = 0; i < 4; i++) { const float4 f = reinterpret_cast<const float4*>(sc + 16 * c)[i]; S[4 * i] = f.x; S[4 * i + 1] = f.y; S[4 * i + 2] = f.z; S[4 * i + 3] = f.w; }
    unsigned short pc[2][16];
    int below = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned short o[2];
        const float vs = v[i] * S[i];
        split2(vs, o);
        pc[0][i] = o[0]; pc[1][i] = o[1];
        below += (mine && vs != 0.0f && fabsf(vs) < 0x1p-12f) ? 1 : 0;
    }
    count_window(below, &g_f16_below_window_cols);
#ifndef IGAN_NO_WINDOW_COUNT
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_f16_imaged_cols, (unsigned long long)total * 16ull);
#endif
    store_units(pc, stage, out, total, (int)blockIdx.x);
}

// Column maxima of a filter: the largest |W(tap, k, n)| over k per (tap [, k quarter], n).  The image kernel below takes the maximum over the partial rows.
//   !WT: w[tap][k][n] (HWIO; n contiguous): grid (taps * H_KSK, ceil(Nn / 64)), 4 row lanes x 64 columns, partial[(tap * H_KSK + j) * Nn + n]
//    WT: w[tap][n][k] (k contiguous): one wave per (tap, n) row, partial[tap * Nn + n]
template <bool WT>
__global__ __launch_bounds__(256) void filter_amax_kernel(const float* __restrict__ w, float* __restrict__ partial, int taps, int Nn, int K) {
    if constexpr (WT) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= taps * Nn) return;
        const float4* src = reinterpret_cast<const float4*>(w + (size_t)row * K);
        float m = 0.0f;
        for (int i = lane; i < (K >> 2); i += 64) { const float4 f = src[i]; m = fmaxf(m, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w)))); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) partial[row] = m;
    } else {
        __shared__ float red[256];
        const int tap = blockIdx.x / H_KSK, j = blockIdx.x - tap * H_KSK;
        const int n = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
        const int k0 = (int)(((long long)j * K) / H_KSK), k1 = (int)(((long long)(j + 1) * K) / H_KSK);
        float m = 0.0f;
        if (n < Nn) {
            const float* src = w + (size_t)tap * K * Nn + n;
            int k = k0 + rl;
            for (; k + 28 < k1; k += 32) {          // eight rows in flight per thread (a row of 64 channels is one 256-byte request of the wave)
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = fabsf(src[(size_t)(k + 4 * i) * Nn]);
                m = fmaxf(m, fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))));
            }
            for (; k < k1; k += 4) m = fmaxf(m, fabsf(src[(size_t)k * Nn]));
        }
        red[threadIdx.x] = m;
        __syncthreads();
        if (rl == 0 && n < Nn) partial[(size_t)blockIdx.x * Nn + n] = fmaxf(fmaxf(m, red[64 + threadIdx.x]), fmaxf(red[128 + threadIdx.x], red[192 + threadIdx.x]));
    }
}

// filter -> [K/16][tap][n][2][16] fp16 with one scale per OUTPUT channel n (orientation as filter_planes_kernel); 1 / S_n to inv[Nn].
// partial: `prow` rows of Nn column maxima (filter_amax_kernel).
template <bool WT>
__global__ __launch_bounds__(256) void filter_planes_f16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, float* __restrict__ inv,
                                                                 const float* __restrict__ partial, int prow, int taps, int KW_, int Nn, int K) {
    const int cpk = K >> 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= taps * Nn * cpk) return;
    int tap, n, c;
    if constexpr (WT) { c = idx % cpk; const int r = idx / cpk; n = r % Nn; tap = r / Nn; }
    else { n = idx % Nn; const int r = idx / Nn; c = r % cpk; tap = r / cpk; }
    float amax = 0.0f;
    {   // independent loads, six in flight (a serial loop over the partial rows is `prow` dependent L2 round trips in front of every thread's work)
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
        int r = 0;
        for (; r + 6 <= prow; r += 6) {
            const float* q = partial + (size_t)r * Nn + n;
            const float v0 = q[0], v1 = q[Nn], v2 = q[2 * (size_t)Nn], v3 = q[3 * (size_t)Nn], v4 = q[4 * (size_t)Nn], v5 = q[5 * (size_t)Nn];
            m0 = fmaxf(m0, v0); m1 = fmaxf(m1, v1); m2 = fmaxf(m2, v2); m3 = fmaxf(m3, v3); m4 = fmaxf(m4, v4); m5 = fmaxf(m5, v5);
        }
        for (; r < prow; r++) m0 = fmaxf(m0, partial[(size_t)r * Nn + n]);
        amax = fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, m5));
    }
    const float S = scale_from_amax(amax);
    if (tap == 0 && c == 0) inv[n] = inv_scale_from_amax(amax);
    float v[16];
    if constexpr (WT) {
        const float4* src = reinterpret_cast<const float4*>(w + ((size_t)(taps - 1 - tap) * Nn + n) * K + 16 * c);
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = src[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = w[((size_t)tap * K + 16 * c + i) * Nn + n];
    }
    unsigned short pc[2][16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned short o[2];
        split2(v[i] * S, o);
        pc[0][i] = o[0]; pc[1][i] = o[1];
    }
    uint4* dst = reinterpret_cast<uint4*>(out + (((size_t)c * taps + tap) * Nn + n) * 32);
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            uint4 u;
            u.x = pc[q][8 * hh + 0] | ((unsigned)pc[q][8 * hh + 1] << 16); u.y = pc[q][8 * hh + 2] | ((unsigned)pc[q][8 * hh + 3] << 16);
            u.z = pc[q][8 * hh + 4] | ((unsigned)pc[q][8 * hh + 5] << 16); u.w = pc[q][8 * hh + 6] | ((unsigned)pc[q][8 * hh + 7] << 16);
            dst[2 * q + hh] = u;
        }
}

// TAPO (fp16 form only): the reduction runs TAP outermost, the 16-channel slices of a tap inside it.  A row's pixel -- and with it the row's scale --
// then stays the same for Cin / 16 consecutive steps, so the cross terms are chained in the matrix pipe over a tap and folded once per tap
// (32 vector instructions per step instead of 64).  !TAPO: slice outermost, taps inside (the taps of a slice re-read the same shifted rows back to
// back); every step folds its own cross terms.  Same products and the same per-step rounding of the main term either way; which one a layer takes is
// a measured choice (igan_conv2d: IGAN_F16_TAP_OUTER).
template <int NP, bool TAPO = false>
__global__ __launch_bounds__(512, 4) void conv_fwd_planes_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 128, WN = 4, TM = 2;
    static_assert(NP == 3 || NP == 2, "three bf16 pieces (six products) or two fp16 pieces (three products)");
    constexpr int IMG = NP * 128 * 32, STAGE = 2 * IMG;       // one operand's LDS image [NP pieces][128 rows][32 B]; a stage = A + B
    constexpr unsigned PB = NP * 32u;                          // bytes of one (pixel, 16-channel slice) in a piece image
    constexpr int TABS = (NP == 2) ? 9 * BM * 4 : 0;           // fp16 form: 1 / S of the input pixel each (tap, tile row) reads (at most 3x3 taps)
#ifndef IGAN_F16_LDS_PAD
#define IGAN_F16_LDS_PAD 0
#endif
    constexpr int LPAD = (NP == 2) ? IGAN_F16_LDS_PAD : 0;     // experiment: pad the fp16 tile's LDS footprint (co-residency with other kernels)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[P_NSTAGE * STAGE + 3 * BM * 4 + TABS + LPAD];
    int* row_pix = reinterpret_cast<int*>(smem + P_NSTAGE * STAGE);
    int* row_n = row_pix + BM;
    float* row_nz = reinterpret_cast<float*>(row_n + BM);
    float* tab_s = row_nz + BM;

#ifdef IGAN_DIAGNOSTIC
    const int dm = a.diag_mode;
    if (dm & 4) return;
#else
    constexpr int dm = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: kept in a scalar register
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int up = 1 << a.up_shift;
    auto stamp = [&](int k) {       // diagnostic only, as in conv_fwd_dma_kernel
        if (a.diag != nullptr && (threadIdx.x >> 6) == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if ((threadIdx.x & 63) == 0) a.diag[(size_t)blockIdx.x * 4 + k] = t;
        }
    };
    stamp(0);
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int bid = blockIdx.x;
    if (a.xcd_remap && bid < a.full_tiles) {
        const int per_class = a.nx * a.ny;
        const int lo = (bid / per_class) * per_class;
        const int cnt = min(per_class, a.full_tiles - lo);
        bid = lo + remap_xcd(bid - lo, cnt);
        // Round 4, STRIDE-2 convolutions only: inside an XCD's contiguous range the tiles run N-TILE FASTEST (when the class is whole).
        // The tile list is m-tile fastest, so an XCD's range is ~nx / 8 m-tiles of ONE filter column block: the ny column blocks of an
        // m-tile run on ny different XCDs and each fetches the m-tile's rows into its own L2.  With the column blocks adjacent in the
        // range they share one L2.  Measured (tools/planes_sched_ab.sh, profiles/r04_planes_experiments.txt): D 64 Conv1_down forward
        // 384 -> 359 us, the data gradient of G 64 Conv0_up 385 -> 359 us (at stride 2 the taps of a slice share few rows, the x image
        // is the traffic); stride-1 layers lose 0.5-1.3 % (their rows are re-read by the nine taps from L2 anyway, and the filter
        // slices of ny column blocks compete for it), so they keep the m-major order.  Placement only: digests equal.
        if (cnt == per_class && a.ny > 1 && a.stride > 1) {
            const int j = bid - lo;
            bid = lo + (j % a.ny) * a.nx + j / a.ny;
        }
    }
    const bool sliced = bid >= a.full_tiles;
    const int tail = bid - a.full_tiles;
    const int tile = sliced ? a.full_tiles + tail / a.splits : bid;
    const int split = sliced ? tail % a.splits : 0;
    const int nsplit = sliced ? a.splits : 1;
    const int mt = tile % a.nx, nt = (tile / a.nx) % a.ny, cls = tile / (a.nx * a.ny);
    const int py = cls >> a.up_shift, px = cls & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Mcls = a.N * QH * QW;
    const int m0 = mt * BM;
    if (m0 >= Mcls) return;
    const int n0 = nt * BN;
    const int ky0 = (a.pad_y - py * a.stride) & (up - 1);
    const int kx0 = (a.pad_x - px * a.stride) & (up - 1);
    const int nky = (ky0 < a.KH) ? ((a.KH - ky0 + up - 1) >> a.up_shift) : 0;
    const int nkx = (kx0 < a.KW) ? ((a.KW - kx0 + up - 1) >> a.up_shift) : 0;
    const int chunks = nky * nkx * a.cpt;                 // a.cpt = Cin / 16 here
    const int c_begin = sliced ? (int)(((long long)split * chunks) / nsplit) : 0;
    const int c_end = sliced ? (int)(((long long)(split + 1) * chunks) / nsplit) : chunks;

    const float inv_hw = 1.0f / (float)(QH * QW), inv_w = 1.0f / (float)QW;
    // ---- DMA lane geometry: every lane fills the same LDS position in each of its three instructions: row 32 (wave & 3) + lane / 2,
    // position lane & 1, i.e. half (lane & 1) ^ ((row >> 3) & 1).  Waves 0-3 fetch A pieces 0, 2 and B piece 1, waves 4-7 A piece 1
    // and B pieces 0, 2 (24 wave instructions per stage, three per wave).
    const int drow = 32 * (wave & 3) + (lane >> 1);
    const int dhalf = (lane & 1) ^ ((drow >> 3) & 1);
    const bool lowave = wave < 4;
    // reduction order: 16-channel slice outermost, taps inside it -- the taps of a slice re-read the same (shifted) 96 B row
    // pieces back to back, so eight of the nine fetches of a row piece hit the XCD's L2
    const int ntap = nky * nkx;
    int ld_cc, ld_t0;
    if constexpr (TAPO) {
        ld_t0 = (c_begin < c_end) ? c_begin / a.cpt : 0;
        ld_cc = (c_begin < c_end) ? c_begin - ld_t0 * a.cpt : 0;
    } else {
        ld_cc = (c_begin < c_end) ? c_begin / ntap : 0;
        ld_t0 = (c_begin < c_end) ? c_begin - ld_cc * ntap : 0;
    }
    int ld_ta = (c_begin < c_end) ? ld_t0 / nkx : 0;
    int ld_tb = (c_begin < c_end) ? ld_t0 - ld_ta * nkx : 0;
    const unsigned xbytes = (unsigned)a.N * a.H * a.W * a.Cin * (2u * NP), wbytes = (unsigned)a.KH * a.KW * a.Cin * a.Cout * (2u * NP);   // host: both < OOB
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.xp), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, (int)wbytes, 0x00020000);
    const unsigned pixA = (unsigned)a.Cin * (2u * NP);           // bytes per pixel
    // Per lane, once per tile: the byte offset of its A row under tap (0, 0) of the class and one validity bit per tap (tap (ta, tb)
    // reads input pixel (iy0 + ta, ix0 + tb): the class's taps are `up` apart and the parity makes the shift exact), and the
    // offset of its B row inside a (slice, tap) block.  A step then costs four vector instructions of address work: add the
    // wave-uniform (tap, slice) displacement, test the tap's bit, select the out-of-range marker.
    unsigned baseA, maskA = 0u, voffB;
    {
        const int m = m0 + drow;
        const bool rok = m < Mcls;
        const int mm = rok ? m : 0;
        const int nn = div_small(mm, QH * QW, inv_hw);
        const int r = mm - nn * (QH * QW);
        const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
        const int vy0 = (qy * up + py) * a.stride - a.pad_y + ky0, vx0 = (qx * up + px) * a.stride - a.pad_x + kx0;
        const int iy0 = vy0 >> a.up_shift, ix0 = vx0 >> a.up_shift;
        baseA = (unsigned)((nn * a.H + iy0) * a.W + ix0) * pixA + (unsigned)dhalf * 16u;     // modulo 2^32; exact for every valid tap
        for (int ta = 0; ta < nky; ta++)
            for (int tb = 0; tb < nkx; tb++) {
                const int iy = iy0 + ta, ix = ix0 + tb;
                const bool ok = rok & (iy >= 0) & (ix >= 0) & (iy < a.H) & (ix < a.W);
                maskA |= ok ? (1u << (ta * nkx + tb)) : 0u;
            }
        const int co = n0 + drow;
        voffB = (co < a.Cout) ? (unsigned)co * PB + (unsigned)dhalf * 16u : OOB;
    }
    unsigned offA = OOB, soffB = 0u;
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned char* dA = nullptr;
    bool dma_fresh = true;
    const unsigned sliceB = (unsigned)(a.KH * a.KW * a.Cout) * PB;          // bytes between two 16-channel slices of the filter image
    auto dma_prep = [&](int stage) {        // addresses of the next chunk, then one step forward in (slice, tap) order
        if (!TAPO || dma_fresh || ld_cc == 0) {     // (wave-uniform) TAPO: a full decode at the first chunk and at every tap start only
            const unsigned disp = (unsigned)(ld_ta * a.W + ld_tb) * pixA + (unsigned)ld_cc * PB;                      // scalar
            const unsigned bit = 1u << (ld_ta * nkx + ld_tb);                                                           // scalar
            offA = (maskA & bit) ? baseA + disp : OOB;
            const int ky = ky0 + (ld_ta << a.up_shift), kx = kx0 + (ld_tb << a.up_shift);
            soffB = __builtin_amdgcn_readfirstlane((unsigned)((ld_cc * (a.KH * a.KW) + ky * a.KW + kx) * a.Cout) * PB);      // scalar
            dma_fresh = false;
        } else {        // the next slice of the same tap: both operands one slice further (an out-of-range marker stays out of range: OOB + 64 Cin / 16 < 2^32)
            offA += PB;
            soffB = __builtin_amdgcn_readfirstlane(soffB + sliceB);       // wave-uniform: it is the instruction's scalar offset (left to itself the compiler keeps it in a vector register and wraps the DMA in a waterfall loop)
        }
        dA = smem + stage * STAGE + (wave & 3) * 1024;
        if constexpr (TAPO) {
            ++ld_cc;
            const int w1 = (ld_cc == a.cpt) ? 1 : 0;
            ld_cc = w1 ? 0 : ld_cc;
            ld_tb += w1;
            const int w2 = (ld_tb == nkx) ? 1 : 0;
            ld_tb = w2 ? 0 : ld_tb;
            ld_ta += w2;
        } else {
            ++ld_tb;
            const int w1 = (ld_tb == nkx) ? 1 : 0;
            ld_tb = w1 ? 0 : ld_tb;
            ld_ta += w1;
            const int w2 = (ld_ta == nky) ? 1 : 0;
            ld_ta = w2 ? 0 : ld_ta;
            ld_cc += w2;
        }
    };
    auto dma_piece = [&](int j) {           // one of this wave's three KiB of the 24 KiB stage; the pieces of a (pixel, slice) are 32 B apart
        unsigned char* A = dA;
        unsigned char* B = dA + IMG;
        // an out-of-range offset stays out of range with the piece offset added
        // (the piece displacement rides in the scalar offset: the instruction's immediate offset would also move the LDS address)
        if constexpr (NP == 2) {        // 16 wave instructions per stage, two per wave: waves 0-3 A piece 0 and B piece 1, waves 4-7 A piece 1 and B piece 0
            const unsigned sB = __builtin_amdgcn_readfirstlane(soffB);      // the instruction's SCALAR offset (wave-uniform by construction; said so, or the compiler wraps the DMA in a waterfall loop)
            if (lowave) {
                if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)A, 16, offA, 0, 0, 0);
                if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(B + 4096), 16, voffB, sB + 32u, 0, 0);
            } else {
                if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 4096), 16, offA, 32, 0, 0);
                if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)B, 16, voffB, sB, 0, 0);
            }
        } else
        if (lowave) {
            if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)A, 16, offA, 0, 0, 0);
            if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 2 * 4096), 16, offA, 64, 0, 0);
            if (j == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(B + 4096), 16, voffB, soffB + 32u, 0, 0);
        } else {
            if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 4096), 16, offA, 32, 0, 0);
            if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)B, 16, voffB, soffB, 0, 0);
            if (j == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(B + 2 * 4096), 16, voffB, soffB + 64u, 0, 0);
        }
    };
    auto dma_chunk = [&](int stage) { dma_prep(stage); dma_piece(0); dma_piece(1); if constexpr (NP == 3) dma_piece(2); };

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][r] = 0.0f;

    // NP == 2: 1 / S of the input pixel that tile row r reads under tap t (0 for padding taps and rows past the end: their A rows are zeros), fetched
    // BEFORE the first DMA instructions (vmcnt counts in order: the table's loads must not wait behind the chunks) and stored to LDS behind them.
    float tabv[3] = {0.0f, 0.0f, 0.0f};
    if constexpr (NP == 2) {
        const float* rowinv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.xp) + (size_t)a.N * a.H * a.W * a.Cin * 4);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = tid + 512 * j;
            const int t = i >> 7, r = i & 127;
            if (t < ntap) {
                const int ta = t / nkx, tb = t - ta * nkx;
                const int m = m0 + r;
                const bool rok = m < Mcls;
                const int mm = rok ? m : 0;
                const int nn = div_small(mm, QH * QW, inv_hw);
                const int rr = mm - nn * (QH * QW);
                const int qy = div_small(rr, QW, inv_w), qx = rr - qy * QW;
                const int iy = (((qy * up + py) * a.stride - a.pad_y + ky0) >> a.up_shift) + ta, ix = (((qx * up + px) * a.stride - a.pad_x + kx0) >> a.up_shift) + tb;
                if (rok & (iy >= 0) & (ix >= 0) & (iy < a.H) & (ix < a.W)) tabv[j] = rowinv[(nn * a.H + iy) * a.W + ix];
            }
        }
    }
    if ((c_begin < c_end) && !(dm & 32)) { dma_chunk(0); dma_chunk(1); }
    if constexpr (NP == 2) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (tid + 512 * j < 9 * BM && !(dm & 64)) tab_s[tid + 512 * j] = tabv[j];
    }
    // the epilogue's row tables, computed while the first chunks are in flight
    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1, nn = 0;
        if (m < Mcls) {
            nn = div_small(m, QH * QW, inv_hw);
            const int r = m - nn * (QH * QW);
            const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
            pix = (nn * a.OH + (qy * up + py)) * a.OW + (qx * up + px);
        }
        row_pix[tid] = pix;
        row_n[tid] = nn;
        row_nz[tid] = noise_term(a, pix);
    }
    // fragment addresses: row r of an image sits at [piece][2 r + (half ^ ((r >> 3) & 1))] x 16 B
    int fa[TM];
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
        const int r = wm * 64 + tm * 32 + l31;
        fa[tm] = (2 * r + (h ^ ((r >> 3) & 1))) * 16;
    }
    const int rb_ = wn * 32 + l31;
    const int fb = IMG + (2 * rb_ + (h ^ ((rb_ >> 3) & 1))) * 16;
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    stamp(1);

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int st = 0;
    if constexpr (NP == 2) {
        // The FILTER fragment is the matrix instruction's first operand, so the accumulators hold the tile transposed: a lane owns ONE pixel row
        // (l31) and its 16 registers are output channels (r & 3) + 8 (r >> 2) + 4 h of the wave's 32 -- the pixel's 1 / S is one value per lane
        // and tile, and the epilogue stores 16 B per register quad.
        const float* tab_row = tab_s + wm * 64 + l31;
        if constexpr (TAPO) {
            int ct = (c_begin < c_end) ? c_begin / a.cpt : 0;                       // tap of the chunk being consumed
            int cs = (c_begin < c_end) ? c_begin - ct * a.cpt : 0;                  // its slice
            bool cs_new = true;
            float sc0 = 0.0f, sc1 = 0.0f;
            f32x16 u[TM];       // the cross terms p0a p1b + p1a p0b of the current tap, chained in the matrix pipe (2^-11 of the tap's sum)
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int r = 0; r < 16; r++) u[tm][r] = 0.0f;
            for (int c = (dm & 8) ? c_end : c_begin; c < c_end; c++) {
                asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // as below
                const int nst = st >= 1 ? st - 1 : P_NSTAGE - 1;
                const unsigned char* S = smem + st * STAGE;
                f16x8 a0[TM], a1[TM];
                const f16x8 b0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fb));
                const f16x8 b1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fb + 4096));
#pragma unroll
                for (int tm = 0; tm < TM; tm++) {
                    a0[tm] = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fa[tm]));
                    a1[tm] = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fa[tm] + 4096));
                }
                if (cs_new) { sc0 = tab_row[ct * BM]; sc1 = tab_row[ct * BM + 32]; cs_new = false; }      // the rows' scales change with the tap only
                dma_prep(nst);
                f32x16 t;           // one set of registers for the main term of both tiles
                __builtin_amdgcn_sched_barrier(0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0[0], zero, 0, 0, 0);       // main term: from an exact zero
                u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0[0], u[0], 0, 0, 0);
                u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0[1], u[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(dm & 32)) { dma_piece(0); dma_piece(1); }            // chunk c + 2, issued inside the matrix cluster
#pragma unroll
                for (int r = 0; r < 16; r++) acc[0][r] = __builtin_fmaf(t[r], sc0, acc[0][r]);        // vector ALU: round to nearest
                __builtin_amdgcn_sched_barrier(0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0[1], zero, 0, 0, 0);
                u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1[0], u[0], 0, 0, 0);
                u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1[1], u[1], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) acc[1][r] = __builtin_fmaf(t[r], sc1, acc[1][r]);
                ++cs;
                if (cs == a.cpt || c + 1 == c_end) {        // the tap (or this slice of the reduction) ends: its cross terms join the sum under the tap's scale
                    const float f0 = sc0 * (1.0f / 2048.0f), f1 = sc1 * (1.0f / 2048.0f);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        acc[0][r] = __builtin_fmaf(u[0][r], f0, acc[0][r]); acc[1][r] = __builtin_fmaf(u[1][r], f1, acc[1][r]);
                        u[0][r] = 0.0f; u[1][r] = 0.0f;
                    }
                    cs = 0; ++ct; cs_new = true;
                }
                st = (st + 1 == P_NSTAGE) ? 0 : st + 1;
            }
        } else {
        int ct = (c_begin < c_end) ? c_begin - (c_begin / ntap) * ntap : 0;      // tap of the chunk being consumed (the reduction runs slice outermost, taps inside)
        for (int c = c_begin; c < c_end; c++) {
            // chunk c landed (two younger instructions: chunk c+1); stage st+2 is free.  lgkmcnt(0): the first barrier also publishes the scale table
            // (every wave's ds_write has completed before it arrives); later iterations have no LDS operation in flight at this point.
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nst = st >= 1 ? st - 1 : P_NSTAGE - 1;
            const unsigned char* S = smem + st * STAGE;
            f16x8 a0[TM], a1[TM];
            const f16x8 b0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fb));
            const f16x8 b1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fb + 4096));
#pragma unroll
            for (int tm = 0; tm < TM; tm++) {
                a0[tm] = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fa[tm]));
                a1[tm] = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(S + fa[tm] + 4096));
            }
            const float sc0 = tab_row[ct * BM], sc1 = tab_row[ct * BM + 32];
            ct = (ct + 1 == ntap) ? 0 : ct + 1;
            dma_prep(nst);
            // One set of 32 registers (main term t, cross terms v) serves both 32x32 tiles of the wave: every product chain of a step starts from an exact
            // zero, and (t + 2^-11 v) / S_pixel is added to the running sum by the vector ALU (round to nearest).
            f32x16 t, v;
            __builtin_amdgcn_sched_barrier(0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0[0], zero, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0[0], zero, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1[0], v, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma_piece(0); dma_piece(1);             // chunk c + 2, issued inside the matrix cluster
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][r] = __builtin_fmaf(__builtin_fmaf(v[r], 1.0f / 2048.0f, t[r]), sc0, acc[0][r]);
            __builtin_amdgcn_sched_barrier(0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0[1], zero, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0[1], zero, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1[1], v, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) acc[1][r] = __builtin_fmaf(__builtin_fmaf(v[r], 1.0f / 2048.0f, t[r]), sc1, acc[1][r]);
            st = (st + 1 == P_NSTAGE) ? 0 : st + 1;
        }
        }
        drain_lds_dma();
        stamp(2);
        if (dm & 16) return;
        // ---- epilogue of the transposed tile: every register quad is four consecutive output channels of the lane's pixel ----
        // 1 / S_n of the filter's columns (exact: a power of two), then as in conv_fwd_dma_kernel
        const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wp) + (size_t)a.KH * a.KW * a.Cin * a.Cout * 4);
        const int cq = n0 + wn * 32 + 4 * h;        // channel of register quad q: cq + 8 q
        float4 wi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) wi[q] = (cq + 8 * q < a.Cout) ? *reinterpret_cast<const float4*>(winv + cq + 8 * q) : f4zero();     // Cout % 4 == 0 (host)
        if (sliced && nsplit > 1) {
            float* wst = a.y + ((size_t)(tile - a.full_tiles) * a.splits + split) * (BM * BN);
#pragma unroll
            for (int tm = 0; tm < TM; tm++) {
                const int row = wm * 64 + tm * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<float4*>(wst + row * BN + wn * 32 + 8 * q + 4 * h) =
                        make_float4(acc[tm][4 * q] * wi[q].x, acc[tm][4 * q + 1] * wi[q].y, acc[tm][4 * q + 2] * wi[q].z, acc[tm][4 * q + 3] * wi[q].w);
            }
            return;
        }
        __syncthreads();        // the row tables
        const bool scale = a.out_scale != nullptr;
        const float alpha = a.alpha;
        float4 bia[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bia[q] = (a.act && a.bias && cq + 8 * q < a.Cout) ? *reinterpret_cast<const float4*>(a.bias + cq + 8 * q) : f4zero();
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
            const int row = wm * 64 + tm * 32 + l31;
            const int pix = row_pix[row];
            if (pix < 0) continue;
            const float* osc = a.out_scale + (size_t)row_n[row] * a.Cout;
            const float nz = row_nz[row];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = cq + 8 * q;
                if (co >= a.Cout) continue;
                float4 o = make_float4(acc[tm][4 * q] * wi[q].x, acc[tm][4 * q + 1] * wi[q].y, acc[tm][4 * q + 2] * wi[q].z, acc[tm][4 * q + 3] * wi[q].w);
                o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
                if (scale) { const float4 d = *reinterpret_cast<const float4*>(osc + co); o.x *= d.x; o.y *= d.y; o.z *= d.z; o.w *= d.w; }
                if (a.act) {
                    o.x = epi_act(a.act, o.x + nz + bia[q].x, a.act_alpha) * a.act_gain; o.y = epi_act(a.act, o.y + nz + bia[q].y, a.act_alpha) * a.act_gain;
                    o.z = epi_act(a.act, o.z + nz + bia[q].z, a.act_alpha) * a.act_gain; o.w = epi_act(a.act, o.w + nz + bia[q].w, a.act_alpha) * a.act_gain;
                }
                *reinterpret_cast<float4*>(a.out + (size_t)pix * a.Cout + co) = o;
            }
        }
        stamp(3);
        return;
    } else {
    for (int c = c_begin; c < c_end; c++) {
        // chunk c: issued two iterations ago (three instructions of this wave are younger: chunk c+1)
        asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        // everyone has chunk c in stage st, and has finished reading stage st + 2 (chunk c-1): it is refilled with chunk c+2
        const int nst = st >= 1 ? st - 1 : P_NSTAGE - 1;
        const unsigned char* S = smem + st * STAGE;
        bf16x8 af[TM][3], bfr[3];
        const unsigned char* S0 = S;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            bfr[q] = *reinterpret_cast<const bf16x8*>(S0 + fb + q * 4096);
#pragma unroll
            for (int tm = 0; tm < TM; tm++) af[tm][q] = *reinterpret_cast<const bf16x8*>(S0 + fa[tm] + q * 4096);
        }
        dma_prep(nst);
        f32x16 t[TM];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[0], zero, 0, 0, 0);
        int g = 0;
#pragma unroll
        for (int o = 1; o < 3; o++)
#pragma unroll
            for (int i = 0; i <= o; i++) {
                // the next-but-one chunk's three DMA instructions go out in the middle of the matrix cluster (their issue, ~60+ cycles
                // each, then overlaps the cluster instead of delaying its start: 753 -> 648 us on the 32x32 C512 layer; placed
                // after 0 / 1 / 2 / 3 / 4 product groups: 671 / 660 / 648 / 672 / 676)
                if (g == 2) { __builtin_amdgcn_sched_barrier(0); dma_piece(0); dma_piece(1); dma_piece(2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int tm = 0; tm < TM; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][i], bfr[o - i], t[tm], 0, 0, 0);
                ++g;
            }
#pragma unroll
        for (int tm = 0; tm < TM; tm++) acc[tm] += t[tm];
        st = (st + 1 == P_NSTAGE) ? 0 : st + 1;
    }
    }
    drain_lds_dma();
    stamp(2);
    // ---- epilogue (as conv_fwd_dma_kernel) ----
    if (sliced && nsplit > 1) {
        float* wst = a.y + ((size_t)(tile - a.full_tiles) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                wst[row * BN + wn * 32 + l31] = acc[tm][r];
            }
        return;
    }
    __syncthreads();        // the row tables (written before the loop; a tile with no chunk has passed no barrier yet)
    float* out = a.out;
    const bool scale = a.out_scale != nullptr;
    const float alpha = a.alpha;
    const int co = n0 + wn * 32 + l31;
    const bool in = co < a.Cout;
    const int n_first = row_n[0], n_last = row_n[min(BM, Mcls - m0) - 1];
    const bool one_sample = n_first == n_last;
    const float mul = (scale && one_sample && in) ? a.out_scale[n_first * a.Cout + co] : 1.0f;
    const float bia = (a.act && a.bias && in) ? a.bias[co] : 0.0f;
    const bool row_scale = scale && !one_sample;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pix = row_pix[row];
            if (pix < 0 || !in) continue;
            float v = acc[tm][r] * alpha;
            if (scale && one_sample) v *= mul;
            if (row_scale) v *= a.out_scale[row_n[row] * a.Cout + co];
            if (a.act) v = epi_act(a.act, v + row_nz[row] + bia, a.act_alpha) * a.act_gain;
            out[(size_t)pix * a.Cout + co] = v;
        }
    }
    stamp(3);
}

// ------------------------------------------------------------------------------
// Weight-gradient kernel: dw[tap][ci][co] = sum_pixels xs[pixel(tap)][ci] * dys[pixel][co].
// GEMM view: M = Cin tile, N = Cout tile, K = pixels of the tap's parity class.
// grid = (ci tiles, co tiles, taps * splits)
struct WgradArgs {
    const float* x;
    const float* dy;
    float* out;  // dw (splits == 1) or workspace [splits][KH*KW*Cin*Cout]
    const float* in_scale;
    const float* out_scale;
    int N, H, W, Cin;
    int OH, OW, Cout;
    int KH, KW;
    int stride, up_shift;
    int pad_y, pad_x;
    int splits;
    int vecA, vecB, vecSA, vecSB;
    int vecY;   // 8 B stores usable (Cout even, destination 8 B aligned)
    int xcd_remap;  // XCD-aware block order (remap_xcd)
    float alpha;  // dw multiplier (applied here when splits == 1, else by the reduce kernel)
    const unsigned short* xp;    // bf16-piece form (conv_wgrad_planes_kernel): x * in_scale and dy * out_scale as [pixel][C/16][3][16] bf16
    const unsigned short* dyp;
};

// SCM (scale mode, host dispatch on the two pointers): 0 = neither in_scale nor out_scale (plain
// convolutions: no scale loads / addresses / multiplies in the loop), 1 = both (modulated conv),
// 2 = exactly one (the absent one is selected to 1.0).
//
// The reduction axis is the flattened pixel index kp <-> (n, qy, qx) of the tap's parity class.  Each
// loader row starts from one exact decode and then WALKS: a chunk advances every row by BK = 32
// pixels = (st_a1 samples, st_a2 rows, st_b columns), two conditional carries keep (qx, qy) in range,
// and the running byte offsets into x / dy / the scale rows move by precomputed (uniform) deltas --
// no integer division or multiply per chunk.
template <int BM, int BN, int WM, int WN, bool VEC, int SCM>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 4 : 2) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int LDA = BM + 4;
    constexpr int LDB = BN + 4;
    constexpr int A_ELEMS = BK * LDA;
    constexpr int B_ELEMS = BK * LDB;
    constexpr int MV = BM / 4, NV = BN / 4;
    constexpr int NT = WM * WN * 64;                   // threads: 4 waves (one per SIMD) or 8 (two per SIMD)
    constexpr int AROWS = NT / MV, BROWS = NT / NV;    // pixel rows per pass
    constexpr int AR = BK / AROWS, BR = BK / BROWS;    // float4 per thread
    constexpr bool SAME = (AROWS == BROWS);            // A and B loaders walk the same pixel rows
    constexpr int WR = SAME ? 1 : BR;                  // B-side walkers when they differ
    static_assert(AR >= 1 && BR >= 1, "tile too wide");

    __shared__ __attribute__((aligned(16))) float As[2 * A_ELEMS];
    __shared__ __attribute__((aligned(16))) float Bs[2 * B_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int up = 1 << a.up_shift;
    // logical block = (pixel slice, co tile, ci tile, tap) with the tap fastest: the blocks that read one pixel slice (the
    // same dy rows and the same x rows shifted by the tap) are neighbours and, through remap_xcd, share one XCD's L2
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    int tap = bz / a.splits, split = bz - tap * a.splits;
    if (a.xcd_remap) {
        const int gx = gridDim.x, gy = gridDim.y, taps = a.KH * a.KW;
        const int lin = remap_xcd(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        tap = lin % taps;
        int rest = lin / taps;
        bx = rest % gx; rest /= gx;
        by = rest % gy;
        split = rest / gy;
    }
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int py = (a.pad_y - ky) & (up - 1);  // stride == 1 whenever up > 1
    const int px = (a.pad_x - kx) & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Kpix = (QH > 0 && QW > 0) ? a.N * QH * QW : 0;
    const int chunks = (Kpix + BK - 1) / BK;
    const int c_begin = (int)(((long long)split * chunks) / a.splits);
    const int c_end = (int)(((long long)(split + 1) * chunks) / a.splits);
    const int m0 = bx * BM;  // ci
    const int n0 = by * BN;  // co

    const int amv = tid % MV, aprow0 = tid / MV;
    const int bnv = tid % NV, bprow0 = tid / NV;
    const int ci = m0 + 4 * amv, co = n0 + 4 * bnv;

    // input sample of (qy, qx) under this tap: iy = qy*s_in + cy (for up == 2 the class parity makes
    // py + ky - pad even, so the >> 1 of the zero-stuffed coordinate is exact)
    const int s_in = (up == 1) ? a.stride : 1;
    const int cy = (up == 1) ? ky - a.pad_y : (py + ky - a.pad_y) >> 1;
    const int cx = (up == 1) ? kx - a.pad_x : (px + kx - a.pad_x) >> 1;
    // one chunk = BK pixels = st_a1 samples + st_a2 rows + st_b columns
    const int dQW = max(QW, 1), dQH = max(QH, 1);
    const int st_b = BK % dQW, st_a = BK / dQW, st_a2 = st_a % dQH, st_a1 = st_a / dQH;
    const unsigned rowA = (unsigned)a.Cin * 4u, rowB = (unsigned)a.Cout * 4u;   // bytes per pixel
    const unsigned dA_step = (unsigned)((st_a1 * a.H + st_a2 * s_in) * a.W + st_b * s_in) * rowA;
    const unsigned dA_cx = (unsigned)(s_in * (a.W - QW)) * rowA;           // qx -= QW, qy += 1
    const unsigned dA_cy = (unsigned)((a.H - QH * s_in) * a.W) * rowA;     // qy -= QH, n += 1
    const unsigned dB_step = (unsigned)((st_a1 * a.OH + st_a2 * up) * a.OW + st_b * up) * rowB;
    const unsigned dB_cx = (unsigned)(up * (a.OW - QW)) * rowB;
    const unsigned dB_cy = (unsigned)((a.OH - QH * up) * a.OW) * rowB;
    const unsigned dSA_step = (unsigned)st_a1 * rowA, dSB_step = (unsigned)st_a1 * rowB;

    struct Walk { int kp, qx, qy; };
    Walk wa[AR], wb[WR];
    unsigned offA[AR], offSA[AR], offB[BR], offSB[BR];
    auto start = [&](int kp, Walk& w, int& nn) {
        nn = kp / (dQH * dQW);
        const int r = kp - nn * (dQH * dQW);
        w.kp = kp; w.qy = r / dQW; w.qx = r - w.qy * dQW;
    };
    auto advance = [&](Walk& w, bool& c1, bool& c2) {
        w.kp += BK;
        w.qx += st_b;
        c1 = w.qx >= QW;
        w.qx -= c1 ? QW : 0;
        w.qy += st_a2 + (c1 ? 1 : 0);
        c2 = w.qy >= QH;
        w.qy -= c2 ? QH : 0;
    };
#pragma unroll
    for (int i = 0; i < AR; i++) {
        int nn;
        start(c_begin * BK + aprow0 + AROWS * i, wa[i], nn);
        offA[i] = (unsigned)((nn * a.H + wa[i].qy * s_in + cy) * a.W + wa[i].qx * s_in + cx) * rowA + (unsigned)ci * 4u;
        offSA[i] = (unsigned)nn * rowA + (unsigned)ci * 4u;
    }
#pragma unroll
    for (int i = 0; i < BR; i++) {
        int nn;
        Walk w;
        start(c_begin * BK + bprow0 + BROWS * i, w, nn);
        if constexpr (!SAME) wb[i] = w;
        offB[i] = (unsigned)((nn * a.OH + w.qy * up + py) * a.OW + w.qx * up + px) * rowB + (unsigned)co * 4u;
        offSB[i] = (unsigned)nn * rowB + (unsigned)co * 4u;
    }

    float4 ra[AR], rsa[AR], rb[BR], rsb[BR];
    constexpr bool SC = (SCM != 0);
    const bool has_in_scale = a.in_scale != nullptr, has_out_scale = a.out_scale != nullptr;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)a.N * a.H * a.W * a.Cin * 4u);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, (unsigned)a.N * a.OH * a.OW * a.Cout * 4u);
    // an absent scale gets zero records: every load is out of range (no memory access), value unused
    const __amdgpu_buffer_rsrc_t rsi = make_rsrc(has_in_scale ? a.in_scale : a.x, has_in_scale ? (unsigned)a.N * a.Cin * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rso = make_rsrc(has_out_scale ? a.out_scale : a.dy, has_out_scale ? (unsigned)a.N * a.Cout * 4u : 0u);
    LoadAddr aa[AR], as_[AR], ab[BR], abs_[BR];
    // addresses of the walkers' current chunk, then one step forward
    auto prep_chunk = [&](bool live) {
        bool c1a[AR], c2a[AR];
#pragma unroll
        for (int i = 0; i < AR; i++) {
            const int iy = __mul24(wa[i].qy, s_in) + cy, ix = __mul24(wa[i].qx, s_in) + cx;
            const bool ok = live & (wa[i].kp < Kpix) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            const int left = ok ? (a.Cin - ci) : 0;
            aa[i] = make_addr_b<VEC>(offA[i], left);
            if constexpr (SC) as_[i] = make_addr_b<VEC>(offSA[i], left);
            if constexpr (SAME) {
                const int leftb = (live & (wa[i].kp < Kpix)) ? (a.Cout - co) : 0;
                ab[i] = make_addr_b<VEC>(offB[i], leftb);
                if constexpr (SC) abs_[i] = make_addr_b<VEC>(offSB[i], leftb);
            }
            advance(wa[i], c1a[i], c2a[i]);
            offA[i] += dA_step + (c1a[i] ? dA_cx : 0u) + (c2a[i] ? dA_cy : 0u);
            if constexpr (SC) offSA[i] += dSA_step + (c2a[i] ? rowA : 0u);
            if constexpr (SAME) {
                offB[i] += dB_step + (c1a[i] ? dB_cx : 0u) + (c2a[i] ? dB_cy : 0u);
                if constexpr (SC) offSB[i] += dSB_step + (c2a[i] ? rowB : 0u);
            }
        }
        if constexpr (!SAME) {
#pragma unroll
            for (int i = 0; i < BR; i++) {
                const int leftb = (live & (wb[i].kp < Kpix)) ? (a.Cout - co) : 0;
                ab[i] = make_addr_b<VEC>(offB[i], leftb);
                if constexpr (SC) abs_[i] = make_addr_b<VEC>(offSB[i], leftb);
                bool c1, c2;
                advance(wb[i], c1, c2);
                offB[i] += dB_step + (c1 ? dB_cx : 0u) + (c2 ? dB_cy : 0u);
                if constexpr (SC) offSB[i] += dSB_step + (c2 ? rowB : 0u);
            }
        }
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < AR; i++) {
            ra[i] = load4<VEC>(rx, aa[i]);
            if constexpr (SC) rsa[i] = load4<VEC>(rsi, as_[i]);
        }
#pragma unroll
        for (int i = 0; i < BR; i++) {
            rb[i] = load4<VEC>(rdy, ab[i]);
            if constexpr (SC) rsb[i] = load4<VEC>(rso, abs_[i]);
        }
    };
    auto store_chunk = [&](int buf) {
        float* A = As + buf * A_ELEMS;
        float* B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int i = 0; i < AR; i++) {
            float4 v = ra[i];
            if constexpr (SCM == 1) v = f4mul(v, rsa[i]);
            if constexpr (SCM == 2) v = f4mul(v, f4sel(has_in_scale, rsa[i]));
            *reinterpret_cast<float4*>(A + (aprow0 + AROWS * i) * LDA + 4 * amv) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; i++) {
            float4 v = rb[i];
            if constexpr (SCM == 1) v = f4mul(v, rsb[i]);
            if constexpr (SCM == 2) v = f4mul(v, f4sel(has_out_scale, rsb[i]));
            *reinterpret_cast<float4*>(B + (bprow0 + BROWS * i) * LDB + 4 * bnv) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.0f;

    if (c_begin < c_end) {
        prep_chunk(true);
        issue_loads();
        store_chunk(0);
        prep_chunk(c_begin + 1 < c_end);
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; c++) {
        const int cur = (c - c_begin) & 1;
        issue_loads();                        // chunk c+1 (predicated off on the last iteration)
        __builtin_amdgcn_sched_barrier(0);
        float af[TM][16], bf[TN][16];
        load_frag<TM, true, LDA>(As + cur * A_ELEMS, wm * (BM / WM), l31, h, af);
        load_frag<TN, true, LDB>(Bs + cur * B_ELEMS, wn * (BN / WN), l31, h, bf);
        mma_steps<TM, TN, 0, 8, (WM * WN == 4 && SCM != 0)>(af, bf, acc);
        prep_chunk(c + 2 < c_end);            // walk to chunk c+2 in the MFMA shadows
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(cur ^ 1);                 // interleaves with the second half (see conv_fwd_kernel)
        mma_steps<TM, TN, 8, 16, (WM * WN == 4 && SCM != 0)>(af, bf, acc);
        __syncthreads();
    }

    // epilogue: tile -> (ci, co) through tile_row (K-major images interleave the tiles)
    const size_t wsize = (size_t)a.KH * a.KW * a.Cin * a.Cout;
    float* out = a.out + (a.splits > 1 ? (size_t)split * wsize : (size_t)0) + (size_t)tap * a.Cin * a.Cout;
    const int cbase = n0 + wn * (BN / WN);
    const float alpha = (a.splits == 1) ? a.alpha : 1.0f;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int cir = m0 + wm * (BM / WM) + tile_row<TM, true>(tm, (r & 3) + 8 * (r >> 2) + 4 * h);
            if (cir >= a.Cin) continue;
            if constexpr (TN == 2) {
                if (a.vecY) {
                    const int c2 = cbase + 2 * l31;
                    if (c2 < a.Cout) *reinterpret_cast<float2*>(out + (size_t)cir * a.Cout + c2) = make_float2(acc[tm][0][r] * alpha, acc[tm][1][r] * alpha);
                    continue;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int cc = cbase + tile_row<TN, true>(tn, l31);
                if (cc < a.Cout) out[(size_t)cir * a.Cout + cc] = acc[tm][tn][r] * alpha;
            }
        }
    }
}

// ------------------------------------------------------------------------------
// The fp16 form's tile on FOUR waves (a wave owns 64 pixels x 64 channels: 2 x 2 matrix-instruction tiles, twelve products per 16-channel chunk), tap
// outermost as conv_fwd_planes_kernel<2, true>, same images, same LDS stage layout, same products and folds per output element.  What changes is the
// schedule: with 256 registers per lane the fragments of chunk c + 1 are read from LDS into a second register set right after the barrier that frees
// chunk c's stage, while the matrix instructions of chunk c run from the set read one step earlier -- no LDS latency between a barrier and the first
// product, the stage is free one step sooner (three chunks in flight behind the one consumed in a ring of three), eight fragment reads per twelve
// products instead of six per six, and the per-step scalar work is shared by twice the products.  The fragment reads are written as instructions
// (the compiler does not count them: the step's one `s_waitcnt lgkmcnt(0)` ahead of the barrier covers them, and it names the fragment registers as
// its outputs so that no product moves above it).
__global__ __launch_bounds__(256, 2) void conv_fwd_planes_w4_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 128, NS = 3;
    constexpr int IMG = 2 * 128 * 32, STAGE = 2 * IMG;          // one operand's LDS image [2 pieces][128 rows][32 B]; a stage = A + B
    constexpr unsigned PB = 64u;                                // bytes of one (pixel, 16-channel slice) in a piece image
    constexpr int TABS = 9 * BM * 4;                            // 1 / S of the input pixel each (tap, tile row) reads
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE + 3 * BM * 4 + TABS];
    int* row_pix = reinterpret_cast<int*>(smem + NS * STAGE);
    int* row_n = row_pix + BM;
    float* row_nz = reinterpret_cast<float*>(row_n + BM);
    float* tab_s = row_nz + BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int up = 1 << a.up_shift;
    auto stamp = [&](int k) {       // diagnostic only, as in conv_fwd_dma_kernel
        if (a.diag != nullptr && (threadIdx.x >> 6) == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if ((threadIdx.x & 63) == 0) a.diag[(size_t)blockIdx.x * 4 + k] = t;
        }
    };
    stamp(0);
    int bid = blockIdx.x;
    if (a.xcd_remap && bid < a.full_tiles) {        // placement as in conv_fwd_planes_kernel
        const int per_class = a.nx * a.ny;
        const int lo = (bid / per_class) * per_class;
        const int cnt = min(per_class, a.full_tiles - lo);
        bid = lo + remap_xcd(bid - lo, cnt);
        if (cnt == per_class && a.ny > 1 && a.stride > 1) {
            const int j = bid - lo;
            bid = lo + (j % a.ny) * a.nx + j / a.ny;
        }
    }
    const bool sliced = bid >= a.full_tiles;
    const int tail = bid - a.full_tiles;
    const int tile = sliced ? a.full_tiles + tail / a.splits : bid;
    const int split = sliced ? tail % a.splits : 0;
    const int nsplit = sliced ? a.splits : 1;
    const int mt = tile % a.nx, nt = (tile / a.nx) % a.ny, cls = tile / (a.nx * a.ny);
    const int py = cls >> a.up_shift, px = cls & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Mcls = a.N * QH * QW;
    const int m0 = mt * BM;
    if (m0 >= Mcls) return;
    const int n0 = nt * BN;
    const int ky0 = (a.pad_y - py * a.stride) & (up - 1);
    const int kx0 = (a.pad_x - px * a.stride) & (up - 1);
    const int nky = (ky0 < a.KH) ? ((a.KH - ky0 + up - 1) >> a.up_shift) : 0;
    const int nkx = (kx0 < a.KW) ? ((a.KW - kx0 + up - 1) >> a.up_shift) : 0;
    const int chunks = nky * nkx * a.cpt;                 // a.cpt = Cin / 16
    const int c_begin = sliced ? (int)(((long long)split * chunks) / nsplit) : 0;
    const int c_end = sliced ? (int)(((long long)(split + 1) * chunks) / nsplit) : chunks;
    const bool any = c_begin < c_end;

    const float inv_hw = 1.0f / (float)(QH * QW), inv_w = 1.0f / (float)QW;
    // DMA lane geometry: a wave instruction fills 32 rows x 32 B of one piece image; wave w fetches rows 32 w .. 32 w + 31 of both pieces of both operands
    // (four instructions per chunk); lane -> row 32 w + lane / 2, half (lane & 1) ^ ((row >> 3) & 1)
    const int drow = 32 * wave + (lane >> 1);
    const int dhalf = (lane & 1) ^ ((drow >> 3) & 1);
    const int ntap = nky * nkx;
    int ld_t0 = any ? c_begin / a.cpt : 0;
    int ld_cc = any ? c_begin - ld_t0 * a.cpt : 0;
    int ld_ta = any ? ld_t0 / nkx : 0;
    int ld_tb = any ? ld_t0 - ld_ta * nkx : 0;
    const unsigned xbytes = (unsigned)a.N * a.H * a.W * a.Cin * 4u, wbytes = (unsigned)a.KH * a.KW * a.Cin * a.Cout * 4u;   // host: both < OOB
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.xp), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, (int)wbytes, 0x00020000);
    const unsigned pixA = (unsigned)a.Cin * 4u;           // bytes per pixel
    unsigned baseA, maskA = 0u, voffB;
    {
        const int m = m0 + drow;
        const bool rok = m < Mcls;
        const int mm = rok ? m : 0;
        const int nn = div_small(mm, QH * QW, inv_hw);
        const int r = mm - nn * (QH * QW);
        const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
        const int vy0 = (qy * up + py) * a.stride - a.pad_y + ky0, vx0 = (qx * up + px) * a.stride - a.pad_x + kx0;
        const int iy0 = vy0 >> a.up_shift, ix0 = vx0 >> a.up_shift;
        baseA = (unsigned)((nn * a.H + iy0) * a.W + ix0) * pixA + (unsigned)dhalf * 16u;     // modulo 2^32; exact for every valid tap
        for (int ta = 0; ta < nky; ta++)
            for (int tb = 0; tb < nkx; tb++) {
                const int iy = iy0 + ta, ix = ix0 + tb;
                const bool ok = rok & (iy >= 0) & (ix >= 0) & (iy < a.H) & (ix < a.W);
                maskA |= ok ? (1u << (ta * nkx + tb)) : 0u;
            }
        const int co = n0 + drow;
        voffB = (co < a.Cout) ? (unsigned)co * PB + (unsigned)dhalf * 16u : OOB;
    }
    unsigned offA = OOB, soffB = 0u;
    typedef __attribute__((address_space(3))) void lds_void;
    bool dma_fresh = true;
    const unsigned sliceB = (unsigned)(a.KH * a.KW * a.Cout) * PB;          // bytes between two 16-channel slices of the filter image
    auto dma_prep = [&]() {        // addresses of the next chunk, then one step forward in (tap, slice) order
        if (dma_fresh || ld_cc == 0) {     // (wave-uniform) a full decode at the first chunk and at every tap start only
            const unsigned disp = (unsigned)(ld_ta * a.W + ld_tb) * pixA + (unsigned)ld_cc * PB;
            const unsigned bit = 1u << (ld_ta * nkx + ld_tb);
            offA = (maskA & bit) ? baseA + disp : OOB;
            const int ky = ky0 + (ld_ta << a.up_shift), kx = kx0 + (ld_tb << a.up_shift);
            soffB = __builtin_amdgcn_readfirstlane((unsigned)((ld_cc * (a.KH * a.KW) + ky * a.KW + kx) * a.Cout) * PB);
            dma_fresh = false;
        } else {        // the next slice of the same tap: both operands one slice further (an out-of-range marker stays out of range)
            offA += PB;
            soffB = __builtin_amdgcn_readfirstlane(soffB + sliceB);
        }
        ++ld_cc;
        const int w1 = (ld_cc == a.cpt) ? 1 : 0;
        ld_cc = w1 ? 0 : ld_cc;
        ld_tb += w1;
        const int w2 = (ld_tb == nkx) ? 1 : 0;
        ld_tb = w2 ? 0 : ld_tb;
        ld_ta += w2;
    };
    auto dma_pair = [&](int stage, int which) {       // which 0: the A pieces, 1: the B pieces of the chunk prepared last (the piece displacement rides in the scalar offset)
        unsigned char* D = smem + stage * STAGE + wave * 1024;
        if (which == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)D, 16, offA, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(D + 4096), 16, offA, 32, 0, 0);
        } else {
            const unsigned sB = __builtin_amdgcn_readfirstlane(soffB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(D + IMG), 16, voffB, sB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(D + IMG + 4096), 16, voffB, sB + 32u, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // 1 / S of the input pixel that tile row r reads under tap t, fetched BEFORE the first DMA instructions (vmcnt counts in order) and stored to LDS behind them
    float tabv[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float* rowinv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.xp) + (size_t)a.N * a.H * a.W * a.Cin * 4);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = tid + 256 * j;
            const int t = i >> 7, r = i & 127;
            if (t < ntap) {
                const int ta = t / nkx, tb = t - ta * nkx;
                const int m = m0 + r;
                const bool rok = m < Mcls;
                const int mm = rok ? m : 0;
                const int nn = div_small(mm, QH * QW, inv_hw);
                const int rr = mm - nn * (QH * QW);
                const int qy = div_small(rr, QW, inv_w), qx = rr - qy * QW;
                const int iy = (((qy * up + py) * a.stride - a.pad_y + ky0) >> a.up_shift) + ta, ix = (((qx * up + px) * a.stride - a.pad_x + kx0) >> a.up_shift) + tb;
                if (rok & (iy >= 0) & (ix >= 0) & (iy < a.H) & (ix < a.W)) tabv[j] = rowinv[(nn * a.H + iy) * a.W + ix];
            }
        }
    }
    if (any) {
#pragma unroll
        for (int j = 0; j < 3; j++) { dma_prep(); dma_pair(j, 0); dma_pair(j, 1); }      // chunks 0, 1, 2 of this tile's range
    }
#pragma unroll
    for (int j = 0; j < 5; j++)
        if (tid + 256 * j < 9 * BM) tab_s[tid + 256 * j] = tabv[j];
    if (tid < BM) {     // the epilogue's row tables, computed while the first chunks are in flight
        const int m = m0 + tid;
        int pix = -1, nn = 0;
        if (m < Mcls) {
            nn = div_small(m, QH * QW, inv_hw);
            const int r = m - nn * (QH * QW);
            const int qy = div_small(r, QW, inv_w), qx = r - qy * QW;
            pix = (nn * a.OH + (qy * up + py)) * a.OW + (qx * up + px);
        }
        row_pix[tid] = pix;
        row_n[tid] = nn;
        row_nz[tid] = noise_term(a, pix);
    }
    // fragment addresses (LDS bytes): row r of an image sits at [piece][2 r + (half ^ ((r >> 3) & 1))] x 16 B; the second tile of a wave is 32 rows = 1 KiB further
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void*)smem;
    const int ra = wm * 64 + l31, rb = wn * 64 + l31;
    const unsigned fa = lds0 + (unsigned)(2 * ra + (h ^ ((ra >> 3) & 1))) * 16u;
    const unsigned fb = lds0 + (unsigned)IMG + (unsigned)(2 * rb + (h ^ ((rb >> 3) & 1))) * 16u;
    stamp(1);

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    struct Frag { f16x8 a0[2], a1[2], b0[2], b1[2]; };       // pieces 0 / 1 of the wave's two pixel tiles and two channel tiles
    auto read_frags = [&](Frag& f, int stage) {              // eight 16-byte LDS reads, not waited for here
        const unsigned va = fa + (unsigned)(stage * STAGE), vb = fb + (unsigned)(stage * STAGE);
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.b0[0]) : "v"(vb));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.a0[0]) : "v"(va));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.b1[0]) : "v"(vb));
        asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(f.b1[1]) : "v"(vb));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f.a0[1]) : "v"(va));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f.b0[1]) : "v"(vb));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.a1[0]) : "v"(va));
        asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(f.a1[1]) : "v"(va));
    };
    // The accumulators hold the tile transposed (the FILTER fragment is the matrix instruction's first operand): a lane owns ONE pixel row (l31) of each
    // of its two pixel tiles, its 16 registers of a tile are output channels (r & 3) + 8 (r >> 2) + 4 h of the tile's 32.
    const float* tab_row = tab_s + wm * 64 + l31;
    f32x16 u[2][2];     // the cross terms p0a p1b + p1a p0b of the current tap, chained in the matrix pipe (2^-11 of the tap's sum)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) u[i][j][r] = 0.0f;
    int st = 0;                                       // stage of the chunk being consumed
    int ct = any ? c_begin / a.cpt : 0;               // its tap
    int cs = any ? c_begin - ct * a.cpt : 0;          // its slice
    float sc0 = 0.0f, sc1 = 0.0f;
    Frag F0, F1;
    auto fold = [&](f32x16& dst, const f32x16& t, float s) {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[r] = __builtin_fmaf(t[r], s, dst[r]);        // vector ALU: round to nearest
    };
    // TAIL: the last chunk of an odd count -- nothing is fetched behind it (a read into registers that nothing uses afterwards would land in whatever the
    // compiler keeps there by then: every fragment read here is into a set that a later `s_waitcnt` names)
    auto step = [&](Frag& cur, Frag& nxt, int c, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        int st1 = st;
        if constexpr (TAIL) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(cur.a0[0]), "+v"(cur.a0[1]), "+v"(cur.a1[0]), "+v"(cur.a1[1]), "+v"(cur.b0[0]), "+v"(cur.b0[1]), "+v"(cur.b1[0]), "+v"(cur.b1[1]) :: "memory");
        } else {
            // chunk c + 1 has landed (one younger chunk of four instructions in flight); this wave's reads of chunk c (issued one step ago) are in `cur`;
            // behind the barrier every wave has read chunk c: its stage takes chunk c + 3
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier"
                         : "+v"(cur.a0[0]), "+v"(cur.a0[1]), "+v"(cur.a1[0]), "+v"(cur.a1[1]), "+v"(cur.b0[0]), "+v"(cur.b0[1]), "+v"(cur.b1[0]), "+v"(cur.b1[1]) :: "memory");
            st1 = (st + 1 == NS) ? 0 : st + 1;
            read_frags(nxt, st1);
            dma_prep();
        }
        // Twelve products; the main term of a tile starts from an exact zero in one of two register sets and is folded by the vector ALU while the NEXT
        // main term runs in the matrix pipe (its own instruction has had two cross-term products to finish by then).
        f32x16 tA, tB;
        const auto SB = [] { __builtin_amdgcn_sched_barrier(0); };
        SB();
        tA = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[0], cur.a0[0], zero, 0, 0, 0);          // tile (0, 0)
        SB();
        u[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b1[0], cur.a0[0], u[0][0], 0, 0, 0);
        u[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b1[1], cur.a0[0], u[0][1], 0, 0, 0);
        SB();
        if constexpr (!TAIL) dma_pair(st, 0);
        SB();
        tB = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[1], cur.a0[0], zero, 0, 0, 0);          // (0, 1)
        SB();
        fold(acc[0][0], tA, sc0);
        SB();
        u[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b1[0], cur.a0[1], u[1][0], 0, 0, 0);
        u[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b1[1], cur.a0[1], u[1][1], 0, 0, 0);
        SB();
        if constexpr (!TAIL) dma_pair(st, 1);
        SB();
        tA = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[0], cur.a0[1], zero, 0, 0, 0);          // (1, 0)
        SB();
        fold(acc[0][1], tB, sc0);
        SB();
        u[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[0], cur.a1[0], u[0][0], 0, 0, 0);
        u[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[1], cur.a1[0], u[0][1], 0, 0, 0);
        SB();
        tB = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[1], cur.a0[1], zero, 0, 0, 0);          // (1, 1)
        SB();
        fold(acc[1][0], tA, sc1);
        SB();
        u[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[0], cur.a1[1], u[1][0], 0, 0, 0);
        u[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.b0[1], cur.a1[1], u[1][1], 0, 0, 0);
        SB();
        fold(acc[1][1], tB, sc1);
        ++cs;
        if (cs == a.cpt || c + 1 == c_end) {        // the tap (or this slice of the reduction) ends: its cross terms join the sum under the tap's scale
            const float f0 = sc0 * (1.0f / 2048.0f), f1 = sc1 * (1.0f / 2048.0f);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                fold(acc[0][j], u[0][j], f0); fold(acc[1][j], u[1][j], f1);
#pragma unroll
                for (int r = 0; r < 16; r++) { u[0][j][r] = 0.0f; u[1][j][r] = 0.0f; }
            }
            cs = 0; ++ct;
            const int tn_ = min(ct, ntap - 1);
            sc0 = tab_row[tn_ * BM]; sc1 = tab_row[tn_ * BM + 32];       // the next tap's scales (behind the barriers that published the table)
        }
        st = st1;
    };
    if (any) {
        // chunk 0 has landed (two younger chunks in flight), the scale table is stored: publish both, then the first fragments and scales
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(F0, 0);
        sc0 = tab_row[ct * BM]; sc1 = tab_row[ct * BM + 32];
        int c = c_begin;
        for (; c + 1 < c_end; c += 2) { step(F0, F1, c, std::false_type()); step(F1, F0, c + 1, std::false_type()); }
        if (c < c_end) step(F0, F1, c, std::true_type());
        else            // an even count: the set read behind the last chunk stays named until its reads have landed
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(F0.a0[0]), "+v"(F0.a0[1]), "+v"(F0.a1[0]), "+v"(F0.a1[1]), "+v"(F0.b0[0]), "+v"(F0.b0[1]), "+v"(F0.b1[0]), "+v"(F0.b1[1]) :: "memory");
    }
    drain_lds_dma();
    stamp(2);
    // ---- epilogue of the transposed tiles: every register quad is four consecutive output channels of the lane's pixel ----
    const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wp) + (size_t)a.KH * a.KW * a.Cin * a.Cout * 4);
    if (sliced && nsplit > 1) {
        float* wst = a.y + ((size_t)(tile - a.full_tiles) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const int cq = n0 + wn * 64 + tn * 32 + 4 * h;
            float4 wi[4];
#pragma unroll
            for (int q = 0; q < 4; q++) wi[q] = (cq + 8 * q < a.Cout) ? *reinterpret_cast<const float4*>(winv + cq + 8 * q) : f4zero();
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                const int row = wm * 64 + tm * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<float4*>(wst + row * BN + wn * 64 + tn * 32 + 8 * q + 4 * h) =
                        make_float4(acc[tm][tn][4 * q] * wi[q].x, acc[tm][tn][4 * q + 1] * wi[q].y, acc[tm][tn][4 * q + 2] * wi[q].z, acc[tm][tn][4 * q + 3] * wi[q].w);
            }
        }
        return;
    }
    __syncthreads();        // the row tables
    const bool scale = a.out_scale != nullptr;
    const float alpha = a.alpha;
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int cq = n0 + wn * 64 + tn * 32 + 4 * h;        // channel of register quad q: cq + 8 q
        float4 wi[4], bia[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            wi[q] = (cq + 8 * q < a.Cout) ? *reinterpret_cast<const float4*>(winv + cq + 8 * q) : f4zero();     // Cout % 4 == 0 (host)
            bia[q] = (a.act && a.bias && cq + 8 * q < a.Cout) ? *reinterpret_cast<const float4*>(a.bias + cq + 8 * q) : f4zero();
        }
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            const int row = wm * 64 + tm * 32 + l31;
            const int pix = row_pix[row];
            if (pix < 0) continue;
            const float* osc = a.out_scale + (size_t)row_n[row] * a.Cout;
            const float nz = row_nz[row];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = cq + 8 * q;
                if (co >= a.Cout) continue;
                float4 o = make_float4(acc[tm][tn][4 * q] * wi[q].x, acc[tm][tn][4 * q + 1] * wi[q].y, acc[tm][tn][4 * q + 2] * wi[q].z, acc[tm][tn][4 * q + 3] * wi[q].w);
                o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
                if (scale) { const float4 d = *reinterpret_cast<const float4*>(osc + co); o.x *= d.x; o.y *= d.y; o.z *= d.z; o.w *= d.w; }
                if (a.act) {
                    o.x = epi_act(a.act, o.x + nz + bia[q].x, a.act_alpha) * a.act_gain; o.y = epi_act(a.act, o.y + nz + bia[q].y, a.act_alpha) * a.act_gain;
                    o.z = epi_act(a.act, o.z + nz + bia[q].z, a.act_alpha) * a.act_gain; o.w = epi_act(a.act, o.w + nz + bia[q].w, a.act_alpha) * a.act_gain;
                }
                *reinterpret_cast<float4*>(a.out + (size_t)pix * a.Cout + co) = o;
            }
        }
    }
    stamp(3);
}

// Weight gradient in bf16-piece form (IGAN_CONV_PLANES=1; the arithmetic of conv_fwd_planes_kernel).  GEMM view as in
// conv_wgrad_kernel: M = 128 input channels, N = 128 output channels, K = the pixels of the tap's parity class, 16 per step.
// Both operands are read ALONG the pixel axis, which the piece images ([pixel][C/16][piece][16]) do not have contiguous: a stage
// holds them as they are -- per piece a [16 pixel rows][128 channels] bf16 image, 256 B per row -- and the fragments are read
// with ds_read_b64_tr_b16 (a 16-lane group fetches 4 rows x 16 columns and every lane receives one COLUMN: four consecutive
// pixels of its channel; two reads make the eight k values of a 32x32x16 operand; tools/tr_read_probe.hip).  The 16 B chunk
// (8 channels) ch of row r sits at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))): conflict-free transposed reads
// (cdna_hip_programming.md T10, image (b)).  A DMA instruction fills four rows of one piece (1 KiB); its lanes fetch the chunk
// that belongs in their slot; padding taps, the ragged end of the pixel axis and channel tails are out-of-range offsets (zeros).
// grid and split as conv_wgrad_kernel; partial tiles go to the same workspace and plain_reduce_kernel adds them in fixed order.
typedef __attribute__((ext_vector_type(4))) short s16x4;
// The weight-gradient kernel issues its LDS-DMA through inline assembly (round 4, second session).  Its fragments are read with the
// ds_read_b64_tr_b16 builtin, and the compiler -- which counts a `buffer_load ... lds` builtin as a pending write to LDS -- put an
// `s_waitcnt vmcnt(0)` in front of those reads in every step: each wave then waited for the chunk it had issued ONE step earlier, i.e. the
// two-deep prefetch was one deep.  (The forward kernel's plain ds_read_b128 do not get that wait.)  Issued from assembly the DMA is invisible to
// the compiler's counters and the kernel's own `s_waitcnt vmcnt(NP)` in front of the barrier is the only wait, as designed; vector-memory
// instructions the compiler does count (the epilogue's stores) only ever wait longer for it.  -DIGAN_WGRAD_ASM_DMA=0 restores the builtin.
#ifndef IGAN_WGRAD_ASM_DMA
#define IGAN_WGRAD_ASM_DMA 1
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 raw_rsrc_words(const void* base, unsigned bytes) {     // the words __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000) makes
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
template <int SOFF>      // the piece displacement rides in the SCALAR offset (an inline constant): the instruction's immediate offset would also move the LDS address
__device__ __forceinline__ void lds_dma16_asm(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {      // 16 B per lane to lds_addr + 16 * lane
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rsrc), "n"(SOFF) : "memory", "m0");
}
template <int NP>
__global__ __launch_bounds__(512, 4) void conv_wgrad_planes_kernel(WgradArgs a) {
    constexpr int TM = 2, WN = 4;
    static_assert(NP == 3 || NP == 2, "three bf16 pieces (six products) or two fp16 pieces (three products)");
    constexpr int IMG = NP * 4096, STAGE = 2 * IMG;            // one operand's LDS image [NP pieces][16 pixel rows][256 B]; a stage = A + B
    constexpr unsigned PB = NP * 32u;                          // bytes of one (pixel, 16-channel slice) in a piece image
    __shared__ __attribute__((aligned(1024))) unsigned char smem[P_NSTAGE * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int up = 1 << a.up_shift;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    int tap = bz / a.splits, split = bz - tap * a.splits;
    if (a.xcd_remap) {       // block order of conv_wgrad_kernel: the taps of one pixel slice are neighbours on one XCD
        const int gx = gridDim.x, gy = gridDim.y, taps = a.KH * a.KW;
        const int lin = remap_xcd(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        tap = lin % taps;
        int rest = lin / taps;
        bx = rest % gx; rest /= gx;
        by = rest % gy;
        split = rest / gy;
    }
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int py = (a.pad_y - ky) & (up - 1);
    const int px = (a.pad_x - kx) & (up - 1);
    const int QH = (a.OH - py + up - 1) >> a.up_shift;
    const int QW = (a.OW - px + up - 1) >> a.up_shift;
    const int Kpix = (QH > 0 && QW > 0) ? a.N * QH * QW : 0;
    const int chunks = (Kpix + PK - 1) / PK;
    const int c_begin = (int)(((long long)split * chunks) / a.splits);
    const int c_end = (int)(((long long)(split + 1) * chunks) / a.splits);
    const int m0 = bx * 128, n0 = by * 128;
    const int s_in = (up == 1) ? a.stride : 1;
    const int cy = (up == 1) ? ky - a.pad_y : (py + ky - a.pad_y) >> 1;
    const int cx = (up == 1) ? kx - a.pad_x : (px + kx - a.pad_x) >> 1;

    // ---- DMA lane geometry: wave w fills rows 4 (w & 3) .. +3 of its three (operand, piece) images; lane -> row 4 (w & 3) + lane / 16,
    // slot lane & 15, channel chunk = slot ^ x(row)
    const int drow = 4 * (wave & 3) + (lane >> 4);
    const int dch = (lane & 15) ^ (((drow & 3) << 2) | ((drow >> 2) & 3));
    const bool lowave = wave < 4;
    const unsigned rowA = (unsigned)a.Cin * (2u * NP), rowB = (unsigned)a.Cout * (2u * NP);       // bytes per pixel
    const bool chA = m0 + 8 * dch < a.Cin, chB = n0 + 8 * dch < a.Cout;
    const unsigned constA = (unsigned)((m0 >> 4) + (dch >> 1)) * PB + (unsigned)(dch & 1) * 16u;
    const unsigned constB = (unsigned)((n0 >> 4) + (dch >> 1)) * PB + (unsigned)(dch & 1) * 16u;
    const int dQW = max(QW, 1), dQH = max(QH, 1);
    const int st_b = PK % dQW, st_a = PK / dQW, st_a2 = st_a % dQH, st_a1 = st_a / dQH;     // one step = 16 pixels = st_a1 samples + st_a2 rows + st_b columns
    int kp = c_begin * PK + drow, wn_ = 0, wqy = 0, wqx = 0;      // this lane's pixel of the next chunk to fetch
    {
        wn_ = kp / (dQH * dQW);
        const int r = kp - wn_ * (dQH * dQW);
        wqy = r / dQW; wqx = r - wqy * dQW;
    }
    const unsigned xbytes = (unsigned)a.N * a.H * a.W * a.Cin * (2u * NP), dybytes = (unsigned)a.N * a.OH * a.OW * a.Cout * (2u * NP);   // host: both < OOB
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.xp), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.dyp), 0, (int)dybytes, 0x00020000);
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned offA = OOB, offB = OOB;
    unsigned char* dA = nullptr;
    // Round 6: the lane's byte offsets into the two images WALK with its pixel (modulo 2^32: exact whenever the pixel is valid): a step adds a wave-uniform delta and
    // one correction per carry.  Decoded anew every step -- ((n H + iy) W + ix) rowA and its twin -- they cost six quarter-rate integer multiplies (v_mul_lo_u32,
    // v_mad_u64_u32) per wave and step, in a loop whose vector work (10 instructions per matrix instruction) already outweighed its six products.  Same addresses, bit-identical results.
    unsigned linA, linB;
    {
        const int iy0 = wqy * s_in + cy, ix0 = wqx * s_in + cx;
        linA = (unsigned)((wn_ * a.H + iy0) * a.W + ix0) * rowA + constA;
        const int oy0 = (wqy << a.up_shift) + py, ox0 = (wqx << a.up_shift) + px;
        linB = (unsigned)((wn_ * a.OH + oy0) * a.OW + ox0) * rowB + constB;
    }
    const unsigned dA0 = (unsigned)((st_a1 * a.H + s_in * st_a2) * a.W + s_in * st_b) * rowA;       // st_a1 samples + st_a2 rows + st_b columns further
    const unsigned dA1 = (unsigned)(s_in * a.W - s_in * QW) * rowA;                                 // the column carry: one row down, QW columns back
    const unsigned dA2 = (unsigned)((a.H - s_in * QH) * a.W) * rowA;                                // the row carry: the next sample, QH rows back
    const unsigned dB0 = (unsigned)((st_a1 * a.OH + up * st_a2) * a.OW + up * st_b) * rowB;
    const unsigned dB1 = (unsigned)(up * a.OW - up * QW) * rowB;
    const unsigned dB2 = (unsigned)((a.OH - up * QH) * a.OW) * rowB;
    auto dma_prep = [&](int stage) {
        const int iy = __mul24(wqy, s_in) + cy, ix = __mul24(wqx, s_in) + cx;
        const bool live = kp < Kpix;
        const bool okA = live & chA & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        offA = okA ? linA : OOB;
        offB = (live & chB) ? linB : OOB;
        dA = smem + stage * STAGE + (wave & 3) * 1024;
        kp += PK;                                   // walk to the same row of the next chunk
        wqx += st_b;
        const bool c1 = wqx >= QW;
        wqx -= c1 ? QW : 0;
        wqy += st_a2 + (c1 ? 1 : 0);
        const bool c2 = wqy >= QH;
        wqy -= c2 ? QH : 0;
        linA += dA0 + (c1 ? dA1 : 0u) + (c2 ? dA2 : 0u);
        linB += dB0 + (c1 ? dB1 : 0u) + (c2 ? dB2 : 0u);
    };
#if IGAN_WGRAD_ASM_DMA
    const u32x4 wx = raw_rsrc_words(a.xp, xbytes), wdy = raw_rsrc_words(a.dyp, dybytes);
    auto dma_piece = [&](int j) {       // as in conv_fwd_planes_kernel: waves 0-3 A pieces 0, 2 and B piece 1; waves 4-7 A piece 1 and B pieces 0, 2 (NP == 2: A 0, B 1 / A 1, B 0)
        const unsigned A = (unsigned)(uintptr_t)(lds_void*)dA, B = A + IMG;
        if constexpr (NP == 2) {
            if (lowave) {
                if (j == 0) lds_dma16_asm<0>(wx, A, offA);
                if (j == 1) lds_dma16_asm<32>(wdy, B + 4096, offB);
            } else {
                if (j == 0) lds_dma16_asm<32>(wx, A + 4096, offA);
                if (j == 1) lds_dma16_asm<0>(wdy, B, offB);
            }
        } else
        if (lowave) {
            if (j == 0) lds_dma16_asm<0>(wx, A, offA);
            if (j == 1) lds_dma16_asm<64>(wx, A + 2 * 4096, offA);
            if (j == 2) lds_dma16_asm<32>(wdy, B + 4096, offB);
        } else {
            if (j == 0) lds_dma16_asm<32>(wx, A + 4096, offA);
            if (j == 1) lds_dma16_asm<0>(wdy, B, offB);
            if (j == 2) lds_dma16_asm<64>(wdy, B + 2 * 4096, offB);
        }
    };
#else
    auto dma_piece = [&](int j) {       // as in conv_fwd_planes_kernel: waves 0-3 A pieces 0, 2 and B piece 1; waves 4-7 A piece 1 and B pieces 0, 2
        unsigned char* A = dA;
        unsigned char* B = dA + IMG;
        if constexpr (NP == 2) {        // waves 0-3 A piece 0 and B piece 1, waves 4-7 A piece 1 and B piece 0
            if (lowave) {
                if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)A, 16, offA, 0, 0, 0);
                if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void*)(B + 4096), 16, offB, 32, 0, 0);
            } else {
                if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 4096), 16, offA, 32, 0, 0);
                if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void*)B, 16, offB, 0, 0, 0);
            }
        } else
        if (lowave) {
            if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)A, 16, offA, 0, 0, 0);
            if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 2 * 4096), 16, offA, 64, 0, 0);
            if (j == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void*)(B + 4096), 16, offB, 32, 0, 0);
        } else {
            if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(A + 4096), 16, offA, 32, 0, 0);
            if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void*)B, 16, offB, 0, 0, 0);
            if (j == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void*)(B + 2 * 4096), 16, offB, 64, 0, 0);
        }
    };

#endif
    // ---- transposed fragment reads: lane 16 g + 4 q + p supplies row q, columns 4 p .. 4 p + 3 of its group's block
    // (k group g >> 1, channels 16 (g & 1) .. + 15 of the 32-wide fragment); the second read takes the block four rows below
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    int fA[TM][2], fB[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int row = 8 * (tg >> 1) + 4 * hf + tq;
        const int xr = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
            const int chunk = wm * 8 + tm * 4 + 2 * (tg & 1) + (tp >> 1);
            fA[tm][hf] = 256 * row + 16 * (chunk ^ xr) + 8 * (tp & 1);
        }
        const int chunkb = wn * 4 + 2 * (tg & 1) + (tp >> 1);
        fB[hf] = IMG + 256 * row + 16 * (chunkb ^ xr) + 8 * (tp & 1);
    }
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr8 = [&](const unsigned char* base, int o0, int o1) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][r] = 0.0f;
    if (c_begin < c_end) {
        dma_prep(0); dma_piece(0); dma_piece(1); if constexpr (NP == 3) dma_piece(2);
        dma_prep(1); dma_piece(0); dma_piece(1); if constexpr (NP == 3) dma_piece(2);
    }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int st = 0;
    f32x16 u[TM];       // NP == 2: the cross terms, as in conv_fwd_planes_kernel
    if constexpr (NP == 2) {
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int r = 0; r < 16; r++) u[tm][r] = 0.0f;
        for (int c = c_begin; c < c_end; c++) {
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            const int nst = st >= 1 ? st - 1 : P_NSTAGE - 1;
            const unsigned char* S = smem + st * STAGE;
            f16x8 a0[TM], a1[TM];
            const f16x8 b0 = __builtin_bit_cast(f16x8, tr8(S, fB[0], fB[1]));
            const f16x8 b1 = __builtin_bit_cast(f16x8, tr8(S + 4096, fB[0], fB[1]));
#pragma unroll
            for (int tm = 0; tm < TM; tm++) {
                a0[tm] = __builtin_bit_cast(f16x8, tr8(S, fA[tm][0], fA[tm][1]));
                a1[tm] = __builtin_bit_cast(f16x8, tr8(S + 4096, fA[tm][0], fA[tm][1]));
            }
            dma_prep(nst);
            f32x16 t;           // one set of registers for the main term of both tiles, as in conv_fwd_planes_kernel
            __builtin_amdgcn_sched_barrier(0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[0], b0, zero, 0, 0, 0);
            u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[0], b1, u[0], 0, 0, 0);
            u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[1], b1, u[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma_piece(0); dma_piece(1);
            acc[0] += t;
            __builtin_amdgcn_sched_barrier(0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[1], b0, zero, 0, 0, 0);
            u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[0], b0, u[0], 0, 0, 0);
            u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[1], b0, u[1], 0, 0, 0);
            acc[1] += t;
            st = (st + 1 == P_NSTAGE) ? 0 : st + 1;
        }
        // sum = (main + 2^-11 cross) / (S_ci S_co): the scales are per channel of x in_scale (rows) and of dy out_scale (columns), constant along the
        // reduction, powers of two (exact); applied one after the other so that no intermediate leaves fp32's range
        const float* xinv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.xp) + (size_t)a.N * a.H * a.W * a.Cin * 4);
        const float* dyinv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dyp) + (size_t)a.N * a.OH * a.OW * a.Cout * 4);
        const int co_ = n0 + wn * 32 + l31;
        const float inv_b = (co_ < a.Cout) ? dyinv[co_] : 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ci = m0 + wm * 64 + tm * 32 + 8 * q + 4 * h;       // rows of register quad q: ci .. ci + 3 (Cin % 32 == 0: all in or all out)
                const float4 ia = (ci < a.Cin) ? *reinterpret_cast<const float4*>(xinv + ci) : f4zero();
                acc[tm][4 * q + 0] = ((acc[tm][4 * q + 0] + u[tm][4 * q + 0] * (1.0f / 2048.0f)) * ia.x) * inv_b;
                acc[tm][4 * q + 1] = ((acc[tm][4 * q + 1] + u[tm][4 * q + 1] * (1.0f / 2048.0f)) * ia.y) * inv_b;
                acc[tm][4 * q + 2] = ((acc[tm][4 * q + 2] + u[tm][4 * q + 2] * (1.0f / 2048.0f)) * ia.z) * inv_b;
                acc[tm][4 * q + 3] = ((acc[tm][4 * q + 3] + u[tm][4 * q + 3] * (1.0f / 2048.0f)) * ia.w) * inv_b;
            }
    } else
    for (int c = c_begin; c < c_end; c++) {
        asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        const int nst = st >= 1 ? st - 1 : P_NSTAGE - 1;
        const unsigned char* S = smem + st * STAGE;
        bf16x8 af[TM][3], bfr[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            bfr[q] = tr8(S + q * 4096, fB[0], fB[1]);
#pragma unroll
            for (int tm = 0; tm < TM; tm++) af[tm][q] = tr8(S + q * 4096, fA[tm][0], fA[tm][1]);
        }
        dma_prep(nst);
        f32x16 t[TM];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[0], zero, 0, 0, 0);
        int g = 0;
#pragma unroll
        for (int o = 1; o < 3; o++)
#pragma unroll
            for (int i = 0; i <= o; i++) {
                if (g == 2) { __builtin_amdgcn_sched_barrier(0); dma_piece(0); dma_piece(1); dma_piece(2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int tm = 0; tm < TM; tm++) t[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][i], bfr[o - i], t[tm], 0, 0, 0);
                ++g;
            }
#pragma unroll
        for (int tm = 0; tm < TM; tm++) acc[tm] += t[tm];
        st = (st + 1 == P_NSTAGE) ? 0 : st + 1;
    }
    drain_lds_dma();
    // epilogue: rows = input channels, columns = output channels (contiguous across lanes)
    const size_t wsize = (size_t)a.KH * a.KW * a.Cin * a.Cout;
    float* out = a.out + (a.splits > 1 ? (size_t)split * wsize : (size_t)0) + (size_t)tap * a.Cin * a.Cout;
    const float alpha = (a.splits == 1) ? a.alpha : 1.0f;
    const int co = n0 + wn * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ci = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < a.Cin && co < a.Cout) out[(size_t)ci * a.Cout + co] = acc[tm][r] * alpha;
        }
}

// y[i] = alpha * sum_k ws[k][i], fixed order.  Four partial sums so that four loads are in flight per lane (a single
// running sum is `splits` dependent L2 round trips: 256 slices took 200 us for a 1.5 KB result).
__global__ __launch_bounds__(256) void plain_reduce_kernel(const float* ws, float* y, int total, int splits, float alpha) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            const float a0 = ws[(size_t)k * total + i], a1 = ws[(size_t)(k + 1) * total + i];
            const float a2 = ws[(size_t)(k + 2) * total + i], a3 = ws[(size_t)(k + 3) * total + i];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; k < splits; k++) s0 += ws[(size_t)k * total + i];
        y[i] = ((s0 + s1) + (s2 + s3)) * alpha;
    }
}

// ---- host-side planning --------------------------------------------------------
struct FwdTile { int BM, BN; };

FwdTile pick_fwd_tile(int Mmax, int Cout) {
    FwdTile t;
    t.BN = (Cout > 64) ? 128 : (Cout > 32 ? 64 : 32);
    t.BM = (Mmax <= 32 && t.BN == 128) ? 32 : 128;
    return t;
}

int fwd_geometry_check(const igan_conv2d_params* p) {
    IGAN_REQUIRE(p->x && p->w && p->y, "conv2d: null buffer");
    IGAN_REQUIRE(p->N >= 1 && p->H >= 1 && p->W >= 1 && p->Cin >= 1, "conv2d: input dims must be positive");
    IGAN_REQUIRE(p->OH >= 1 && p->OW >= 1 && p->Cout >= 1, "conv2d: output dims must be positive");
    IGAN_REQUIRE(p->KH >= 1 && p->KW >= 1, "conv2d: kernel must be at least 1x1");
    IGAN_REQUIRE(p->stride >= 1 && p->up >= 1, "conv2d: stride and up must be at least 1");
    IGAN_REQUIRE(p->stride == 1 || p->up == 1, "conv2d: at most one of stride, up may exceed 1");
    if (!(p->up == 1 || p->up == 2)) return igan::fail(IGAN_ERR_UNSUPPORTED, "conv2d: up must be 1 or 2 (got %d)", p->up);
    // operands are addressed with 32-bit byte offsets through buffer descriptors: < 2 GiB each
    IGAN_REQUIRE((long long)p->N * p->H * p->W * p->Cin * 4 <= 0x7FFFFFF0LL, "conv2d: input too large (2 GiB per operand)");
    IGAN_REQUIRE((long long)p->N * p->OH * p->OW * p->Cout <= INT32_MAX, "conv2d: output too large");
    IGAN_REQUIRE((long long)p->KH * p->KW * p->Cin * p->Cout * 4 <= 0x7FFFFFF0LL, "conv2d: filter too large (2 GiB per operand)");
    IGAN_REQUIRE(p->act >= 0 && p->act <= 3, "conv2d: fused epilogue act must be 0 (none), 1 linear, 2 relu or 3 lrelu");
    IGAN_REQUIRE(p->act == 0 || p->act_gain > 0.0f, "conv2d: fused epilogue gain must be positive");
    IGAN_REQUIRE(p->noise == nullptr || (p->act != 0 && p->noise_strength != nullptr), "conv2d: noise needs the fused epilogue (act != 0) and a strength scalar");
    return IGAN_OK;
}

// The 128x128 (and 128x64) tile runs with 8 wavefronts (2 x 4 of 64x32 accumulators, two per SIMD, four with both resident
// workgroups) rather than 4 (2 x 2 of 64x64): the matrix pipe of a SIMD then always has a second wave of the
// same workgroup to issue from while one waits on LDS or the barrier (+3 % on the layer mix, more for a
// workgroup that is alone on its CU).  IGAN_CONV_8WAVE=0 / IGAN_WGRAD_8WAVE=0 select the 4-wave form (A/B runs).
bool eight_waves(const char* env) {
    const char* v = getenv(env);
    return !(v && atoi(v) == 0);
}

// 1x1 convolution on a 1x1 map = a dense layer; with at most 32 rows and no scales it goes to dense_small.hip
bool is_small_dense(const igan_conv2d_params* p) {
    return p->H == 1 && p->W == 1 && p->OH == 1 && p->OW == 1 && p->KH == 1 && p->KW == 1 && p->stride == 1 && p->up == 1 &&
           p->pad_y == 0 && p->pad_x == 0 && !p->in_scale && !p->out_scale && p->act == 0 &&
           igan::dense_small_ok(p->N, p->Cin, p->Cout, p->x, p->w, p->w_transposed != 0);
}
bool is_small_dense_wgrad(const igan_conv2d_wgrad_params* p) {
    return p->H == 1 && p->W == 1 && p->OH == 1 && p->OW == 1 && p->KH == 1 && p->KW == 1 && p->stride == 1 && p->up == 1 &&
           p->pad_y == 0 && p->pad_x == 0 && !p->in_scale && !p->out_scale &&
           igan::dense_small_wgrad_ok(p->N, p->Cout, p->dy, p->dw);
}

void fwd_counts(const igan_conv2d_params* p, int& Mmax, int& chunks_max, int& nclass) {
    const int up = p->up;
    nclass = up * up;
    const int QH = (p->OH + up - 1) / up, QW = (p->OW + up - 1) / up;
    Mmax = p->N * QH * QW;
    const int cpt = (p->Cin + BK - 1) / BK;
    // largest class tap count: ceil(KH/up)*ceil(KW/up)
    chunks_max = ((p->KH + up - 1) / up) * ((p->KW + up - 1) / up) * cpt;
}

template <int BM, int BN, int WM, int WN, bool WT, bool VEC>
void launch_fwd2(hipStream_t stream, const ConvArgs& a, dim3 grid) {
    if (a.in_scale) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, WT, VEC, true>), grid, dim3(WM * WN * 64), 0, stream, a);
    else hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, WT, VEC, false>), grid, dim3(WM * WN * 64), 0, stream, a);
}

template <int BM, int BN, int WM, int WN>
void launch_fwd(hipStream_t stream, const ConvArgs& a, dim3 grid, bool wt, bool vec) {
    if (wt) {
        if (vec) launch_fwd2<BM, BN, WM, WN, true, true>(stream, a, grid);
        else launch_fwd2<BM, BN, WM, WN, true, false>(stream, a, grid);
    } else {
        if (vec) launch_fwd2<BM, BN, WM, WN, false, true>(stream, a, grid);
        else launch_fwd2<BM, BN, WM, WN, false, false>(stream, a, grid);
    }
}

}  // namespace

namespace {

// CUs of the device: tiles are dealt to CUs round-robin, two resident per CU (LDS / VGPR).
int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int n = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        (void)hipGetLastError();      // a build box without a device must not leave a sticky error behind
        cus = n;
    }
    return cus;
}

// Tile list of a forward-type launch: T tiles, of which the last `rem` (the partial round) may be sliced.
struct TileList { int nx, ny, T, rem; };
TileList tile_list(const igan_conv2d_params* p, const FwdTile& t, int Mmax, int nclass) {
    TileList l;
    l.nx = igan::ceil_div(Mmax, t.BM);
    l.ny = igan::ceil_div(p->Cout, t.BN);
    l.T = l.nx * l.ny * nclass;
    l.rem = l.T % device_cus();
    return l;
}


// Does this launch take the LDS-DMA form of the 128x128 tile (conv_fwd_dma_kernel)?  16 B paths with Cin % 32 == 0, 8-wave
// tiles, and -- when an input scale is present -- the scale rows of the samples one tile can touch fitting its LDS table.
bool use_dma_kernel(const igan_conv2d_params* p, const FwdTile& t, bool walk) {
    static const bool dma = !(getenv("IGAN_CONV_DMA") && atoi(getenv("IGAN_CONV_DMA")) == 0);      // A/B switch
    if (!dma || t.BM != 128 || t.BN != 128 || !walk || !eight_waves("IGAN_CONV_8WAVE")) return false;
    if ((long long)p->N * p->OH * p->OW >= (1LL << 24)) return false;          // the kernel's row decode divides in float (div_small)
    if (p->in_scale) {      // the scale rows of all samples a tile can touch must fit the kernel's LDS table (2048 floats)
        const int up = p->up;
        for (int c = 0; c < up * up; c++) {
            const int qh = (p->OH - c / up + up - 1) / up, qw = (p->OW - c % up + up - 1) / up;
            if (qh <= 0 || qw <= 0) continue;
            const int samples = 127 / (qh * qw) + 2;          // a 128-row tile starting anywhere inside a sample
            if ((long long)std::min(samples, p->N) * p->Cin > 2048) return false;
        }
    }
    return true;
}

bool walk_ok(const igan_conv2d_params* p) {
    static const bool walk = !(getenv("IGAN_CONV_WALK") && atoi(getenv("IGAN_CONV_WALK")) == 0);     // A/B switch
    const bool wt = p->w_transposed != 0;
    const bool vecA = (p->Cin % 4 == 0) && (((uintptr_t)p->x & 15) == 0);
    const bool vecS = (p->Cin % 4 == 0) && (((uintptr_t)p->in_scale & 15) == 0);
    const bool vecB = ((wt ? p->Cin : p->Cout) % 4 == 0) && (((uintptr_t)p->w & 15) == 0);
    return walk && vecA && vecB && (p->in_scale == nullptr || vecS) && (p->Cin % BK == 0);
}

// The bf16-piece form is the DEFAULT for the shapes below (round 4); IGAN_CONV_PLANES=0 runs every convolution on the fp32 instruction.
// Which piece form: IGAN_CONV_PLANES unset or 2 = two fp16 pieces, three products (see "TWO-PIECE fp16 form" above; the default since the second
// session of round 4), 1 = three bf16 pieces, six products (the default before it), 0 = none.  Same shapes, same buffers for both piece forms.
int planes_mode() {
    static const int mode = [] { const char* v = getenv("IGAN_CONV_PLANES"); const int m = v ? atoi(v) : 2; return m == 0 ? 0 : (m == 1 ? 1 : 2); }();
    return mode;
}
bool planes_enabled() { return planes_mode() != 0; }

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// bf16 form: the piece image of x [P][C] (times scale [P / HW][C]); `out` holds P * C * 6 bytes.
void launch_piece_image(hipStream_t stream, const float* x, const float* scale, unsigned short* out, int P_, int HW, int C) {
    const int cpp = C / PK, total = P_ * cpp;
    hipLaunchKernelGGL(to_planes_kernel, dim3(igan::ceil_div(total, 256)), dim3(256), 0, stream, x, scale, out, total, cpp, C, HW);
}
// fp16 form, forward / data gradient: the ROW image (one scale per pixel) of x [P][C]; `out` holds rows_part_bytes(P, C).  C / 16 a power of two <= 64.
// `colmax` (optional, IGAN_COLMAX_FLOATS(C) floats): the tensor's per-channel maxima as a by-product, for the weight gradient's column image of the same tensor
void launch_row_image(hipStream_t stream, const float* x, const float* scale, unsigned short* out, int P_, int HW, int C, float* colmax) {
    const int cpp = C / PK, total = P_ * cpp;
    float* rowinv = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(out) + (size_t)P_ * C * 4);
    const int nchunks = igan::ceil_div(total, 256);
    const int grid = colmax ? std::min(nchunks, H_COLBLOCKS) : nchunks;
    hipLaunchKernelGGL(rows_f16_kernel, dim3(grid), dim3(256), 0, stream, x, scale, out, rowinv, colmax, total, cpp, ilog2(cpp), C, HW);
}
// the same maxima from a pass of their own (a call that was asked for them but does not write a row image)
void launch_colmax_only(hipStream_t stream, const float* x, const float* scale, float* colmax, int P_, int HW, int C) {
    const int C4 = C / 4, ppi = 256 / C4;
    const int blocks = std::min(H_COLBLOCKS, igan::ceil_div(P_, ppi));
    hipLaunchKernelGGL(cols_amax_kernel, dim3(blocks), dim3(256), 0, stream, x, scale, colmax + 4, P_, C4, ilog2(C4), HW, colmax);
}
// fp16 form, weight gradient: the COLUMN image (one scale per channel) of x [P][C]; `out` holds cols_part_bytes(P, C).  C / 4 a power of two <= 256.
// `colmax` (optional): the channel maxima the caller holds from the call that wrote this tensor's row image -- the pass over the tensor is skipped
void launch_col_image(hipStream_t stream, const float* x, const float* scale, unsigned short* out, int P_, int HW, int C, const float* colmax) {
    const int cpp = C / PK, total = P_ * cpp, C4 = C / 4;
    float* inv = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(out) + (size_t)P_ * C * 4);
    float* sc = inv + C;
    float* partial = sc + C;
    const int ppi = 256 / C4;
    const int blocks = std::min(H_COLBLOCKS, igan::ceil_div(P_, ppi));
    if (colmax == nullptr) {
        hipLaunchKernelGGL(cols_amax_kernel, dim3(blocks), dim3(256), 0, stream, x, scale, partial, P_, C4, ilog2(C4), HW, (float*)nullptr);
        hipLaunchKernelGGL(cols_finalize_kernel, dim3(igan::ceil_div(C, 8)), dim3(256), 0, stream, (const float*)partial, inv, sc, blocks, C, (const float*)nullptr);
    } else
        hipLaunchKernelGGL(cols_finalize_kernel, dim3(igan::ceil_div(C, 8)), dim3(256), 0, stream, colmax + 4, inv, sc, 0, C, colmax);
    hipLaunchKernelGGL(cols_f16_kernel, dim3(igan::ceil_div(total, 256)), dim3(256), 0, stream, x, scale, out, (const float*)sc, total, cpp, C, HW);
}
// A/B switches inside the fp16 form's forward / data-gradient tile: IGAN_F16_TAP_OUTER=0 the slice-outermost reduction, IGAN_F16_W4=0 the eight-wave tile
bool f16_tap_outer() { static const int v = getenv("IGAN_F16_TAP_OUTER") ? atoi(getenv("IGAN_F16_TAP_OUTER")) : 1; return v != 0; }
bool f16_w4() { static const int v = getenv("IGAN_F16_W4") ? atoi(getenv("IGAN_F16_W4")) : 1; return v != 0; }
void launch_filter_image(hipStream_t stream, const float* w, unsigned short* wp, bool wt, int taps, int KW_, int Nn, int K) {
    const int wtotal = taps * Nn * (K / PK);
    if (planes_mode() == 2) {
        float* inv = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(wp) + (size_t)taps * Nn * K * 4);
        float* partial = inv + Nn;
        if (wt) {
            hipLaunchKernelGGL((filter_amax_kernel<true>), dim3(igan::ceil_div(taps * Nn, 4)), dim3(256), 0, stream, w, partial, taps, Nn, K);
            hipLaunchKernelGGL((filter_planes_f16_kernel<true>), dim3(igan::ceil_div(wtotal, 256)), dim3(256), 0, stream, w, wp, inv, (const float*)partial, taps, taps, KW_, Nn, K);
        } else {
            hipLaunchKernelGGL((filter_amax_kernel<false>), dim3(taps * H_KSK, igan::ceil_div(Nn, 64)), dim3(256), 0, stream, w, partial, taps, Nn, K);
            hipLaunchKernelGGL((filter_planes_f16_kernel<false>), dim3(igan::ceil_div(wtotal, 256)), dim3(256), 0, stream, w, wp, inv, (const float*)partial, taps * H_KSK, taps, KW_, Nn, K);
        }
    } else {
        if (wt) hipLaunchKernelGGL((filter_planes_kernel<true>), dim3(igan::ceil_div(wtotal, 256)), dim3(256), 0, stream, w, wp, taps, KW_, Nn, K);
        else hipLaunchKernelGGL((filter_planes_kernel<false>), dim3(igan::ceil_div(wtotal, 256)), dim3(256), 0, stream, w, wp, taps, KW_, Nn, K);
    }
}

// Does this launch take the bf16-piece form (conv_fwd_planes_kernel)?  the form switched on, the 128x128 tile, Cin % 32 == 0, both
// piece images addressable with 32-bit offsets below the out-of-range marker -- and a reduction deep enough to pay for writing
// the piece images: 3x3 taps on at least 128 channels (taps * Cin >= 1152) over at least 1024 output rows.  Measured per layer
// (tools/conv_layers.py): the 1x1 Skip convolutions and the 4x4 layers lose (D 128 Skip 77 -> 220 us), everything from 16x16 Conv1 up gains.
// Row threshold 1024 (round 6).  Round 5 measured it (fp16 form: three products and a quicker filter image): the 8x8 layers at 24 samples forward 78 -> 70, data gradient
// 78 -> 58, weight gradient 93 -> 71 us, G 16 Conv0_up forward 153 -> 92 us, bench +1.2 % -- and held it back because the path-length step's gradients of two separate runs of
// BASELINE config 2 sat 4-15x further from the fp64 oracle at 1024.  Round 6 (profiles/r06_second_order.txt): no call of that step has 1024-2047 rows (its gradient bucket is
// bit-identical under both thresholds); the two runs' STATES differed (beta1 = 0), and on one state every form reads the same deviation.  IGAN_PLANES_MIN_ROWS /
// IGAN_WGRAD_PLANES_MIN_ROWS move the two thresholds.
// planes_shape_ok() is what the PLAN sizes the workspace by (shapes only: plans are cached per shape); the launch also needs 16 B aligned operands.
bool planes_shape_ok(const igan_conv2d_params* p, const FwdTile& t, int Mmax) {
    if (!planes_enabled() || t.BM != 128 || t.BN != 128 || p->Cin % BK != 0) return false;
    static const int min_rows = getenv("IGAN_PLANES_MIN_ROWS") ? atoi(getenv("IGAN_PLANES_MIN_ROWS")) : 1024;
    if (p->KH * p->KW == 1 || (long long)p->KH * p->KW * p->Cin < 1152 || Mmax < min_rows) return false;      // 1x1: the Skip layers and the nearest-neighbour distance GEMM stay on the fp32 instruction
    if ((long long)p->N * p->OH * p->OW >= (1LL << 24)) return false;
    if ((long long)p->N * p->H * p->W * p->Cin * 6 >= 0x7FFFFF00LL || (long long)p->KH * p->KW * p->Cin * p->Cout * 6 >= 0x7FFFFF00LL) return false;
    // fp16 form: the pixel's scale is shared by shuffles among the Cin / 16 threads of a pixel (a power of two within one wave), the kernel's scale
    // table holds nine taps, and the transposed tile stores four output channels per lane and register quad
    if (planes_mode() == 2 && (!pow2(p->Cin / PK) || p->Cin / PK > 64 || p->KH * p->KW > 9 || p->Cout % 4 != 0)) return false;
#ifdef IGAN_DIAGNOSTIC      // bisecting by layer class (variant builds only: make variant VARIANT=diag DEFS=-DIGAN_DIAGNOSTIC)
    {
        static const int only_cin = getenv("IGAN_PLANES_ONLY_CIN") ? atoi(getenv("IGAN_PLANES_ONLY_CIN")) : 0;
        static const int no_act = getenv("IGAN_PLANES_NO_ACT") ? atoi(getenv("IGAN_PLANES_NO_ACT")) : 0;            // 1: not with a fused epilogue; 2: only with one
        static const int no_scale = getenv("IGAN_PLANES_NO_SCALE") ? atoi(getenv("IGAN_PLANES_NO_SCALE")) : 0;      // 1: not modulated; 2: only modulated
        static const int kind = getenv("IGAN_PLANES_KIND") ? atoi(getenv("IGAN_PLANES_KIND")) : 0;                  // 1: stride 1 / up 1 only; 2: stride 2 only; 3: up 2 only
        static const int wt = getenv("IGAN_PLANES_WT") ? atoi(getenv("IGAN_PLANES_WT")) : 0;                        // 1: forward calls only; 2: data gradients only
        if (only_cin && p->Cin != only_cin) return false;
        if ((no_act == 1 && p->act != 0) || (no_act == 2 && p->act == 0)) return false;
        if ((no_scale == 1 && p->in_scale != nullptr) || (no_scale == 2 && p->in_scale == nullptr)) return false;
        if ((kind == 1 && (p->stride != 1 || p->up != 1)) || (kind == 2 && p->stride != 2) || (kind == 3 && p->up != 2)) return false;
        if ((wt == 1 && p->w_transposed) || (wt == 2 && !p->w_transposed)) return false;
    }
#endif
    return true;
}
bool use_planes_kernel(const igan_conv2d_params* p, const FwdTile& t, int Mmax) {
    if (!planes_shape_ok(p, t, Mmax) || (((uintptr_t)p->x | (uintptr_t)p->w | (uintptr_t)p->in_scale | (uintptr_t)p->x_pieces) & 15) != 0) return false;
    if (planes_mode() == 2 && (((uintptr_t)p->y | (uintptr_t)p->out_scale | (uintptr_t)p->bias) & 15) != 0) return false;      // 16 B epilogue accesses
    return true;
}
// room for the piece images behind the partial tiles: the bf16 form's are 6 bytes per element, the fp16 form's a row image of x and a filter image with their scales
size_t planes_x_floats(const igan_conv2d_params* p) {
    if (planes_mode() == 2) return rows_part_bytes((size_t)p->N * p->H * p->W, (size_t)p->Cin) / 4;
    return (size_t)p->N * p->H * p->W * p->Cin * 6 / 4;
}
size_t planes_w_floats(const igan_conv2d_params* p) {
    if (planes_mode() == 2) return filter_part_bytes((size_t)p->KH * p->KW, (size_t)p->Cout, (size_t)p->Cin) / 4;
    return (size_t)p->KH * p->KW * p->Cin * p->Cout * 6 / 4;
}

}  // namespace

// Diagnostic hook (tools/conv_phases.py): when set, every forward-type launch writes 4 time stamps per workgroup (entry, main
// loop start, main loop end, exit; 100 MHz ticks) to this buffer.  Not part of the operator surface.
static unsigned long long* g_conv_diag = nullptr;
extern "C" void igan_debug_set_conv_diag(unsigned long long* p) { g_conv_diag = p; }

static int conv2d_plan_tiles(const igan_conv2d_params* p, int* splits, int* sliced_tiles, size_t* workspace_floats);

extern "C" int igan_conv2d_plan(const igan_conv2d_params* p, int* splits, int* sliced_tiles, size_t* workspace_floats) {
    IGAN_REQUIRE(p && splits && sliced_tiles && workspace_floats, "conv2d_plan: null argument");
    if (int rc = fwd_geometry_check(p)) return rc;
    *splits = 1;
    *sliced_tiles = 0;
    *workspace_floats = 0;
    if (is_small_dense(p) || igan::thin_conv_kind(p)) return IGAN_OK;
    if (int rc = conv2d_plan_tiles(p, splits, sliced_tiles, workspace_floats)) return rc;
    int Mmax, chunks_max, nclass;
    fwd_counts(p, Mmax, chunks_max, nclass);
    if (planes_shape_ok(p, pick_fwd_tile(Mmax, p->Cout), Mmax))      // the piece images of x and of the filter follow the partial tiles
        *workspace_floats += planes_x_floats(p) + planes_w_floats(p);
    return IGAN_OK;
}

static int conv2d_plan_tiles(const igan_conv2d_params* p, int* splits, int* sliced_tiles, size_t* workspace_floats) {
    int Mmax, chunks_max, nclass;
    fwd_counts(p, Mmax, chunks_max, nclass);
    const FwdTile t = pick_fwd_tile(Mmax, p->Cout);
    const TileList l = tile_list(p, t, Mmax, nclass);
    {   // whole rounds: nothing to slice -- unless the CUs end on an unpaired tile (odd tiles per CU) and the A/B switch
        // IGAN_SLICE_ODD=1 asks for that round to be cut in two (both workgroup slots of a CU stay busy to the end)
        static const bool odd = getenv("IGAN_SLICE_ODD") && atoi(getenv("IGAN_SLICE_ODD")) == 1;
        if (l.rem == 0 && !(odd && (l.T / device_cus()) % 2 == 1)) return IGAN_OK;
    }
    {   // A/B switch: IGAN_SLICE_BIG=0 leaves layers with at least one whole round unsliced
        static const bool big = !(getenv("IGAN_SLICE_BIG") && atoi(getenv("IGAN_SLICE_BIG")) == 0);
        if (!big && l.T >= device_cus()) return IGAN_OK;
    }
    // Which trailing tiles to slice, and how finely, by a small cost model (measured on the 128x128 tile):
    // two co-resident workgroups finish a chunk each in 4.4 us, a lone one in 3.3 us (the matrix pipe is
    // per SIMD, so a pair is only 1.5x as efficient as a single).  A CU that holds n blocks of a tile's
    // work each therefore needs  pairs(n) = (n/2) * t2 + (n%2) * t1.  Slicing the partial round plus k whole
    // rounds (k = 0: the tail only) s ways leaves F = T - rem - k*CUs whole tiles and puts
    // ceil((T-F)*s/CUs) slices of 1/s tile on the busiest CU.  Against the gain stands the partial-tile
    // traffic of the fix-up ((2s+1) x the sliced tiles' bytes through HBM) and its launch.
    const int cus = device_cus();
    const double shape = (t.BM == 128 ? 1.0 : 0.4) * (t.BN / 128.0 > 0.5 ? 1.0 : 0.5);
    const double t2 = chunks_max * 4.4e-6 * shape, t1 = chunks_max * 3.3e-6 * shape;
    auto pairs = [&](int n) { return (n / 2) * t2 + (n % 2) * t1; };
    const double bw = 4.0e12;
    const int cand[] = {2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32, 48, 64};
    int best_s = 1, best_sl = 0;
    double best = pairs(igan::ceil_div(l.T, cus));   // unsliced: the busiest CU has one tile of the partial round
    for (int k = 0; k <= std::min(l.T / cus, 3); k++) {
        const int sl = l.rem + k * cus;       // sliced tiles
        if (sl == 0) continue;
        const int whole = (l.T - sl) / cus;   // whole tiles per CU
        for (int c : cand) {
            if (c > std::max(1, chunks_max / 2)) break;
            const double tt = pairs(whole) + pairs(igan::ceil_div(sl * c, cus)) / c +
                              (2.0 * c + 1.0) * 4.0 * sl * t.BM * t.BN / bw + 3e-6;
            if (tt < best * 0.95) { best = tt; best_s = c; best_sl = sl; }   // prefer less slicing unless clearly (>5 %) better
        }
    }
    *splits = best_s;
    *sliced_tiles = best_sl;
    *workspace_floats = (best_s > 1) ? (size_t)best_sl * best_s * t.BM * t.BN : 0;
    return IGAN_OK;
}

// Which instantiation igan_conv2d() launches for these parameters (host-only; for profiling tools:
// the string equals the kernel name rocprofv3 reports, up to the anonymous namespace prefix).
extern "C" int igan_conv2d_kernel_name(const igan_conv2d_params* p, char* buf, int buflen) {
    IGAN_REQUIRE(p && buf && buflen > 0, "conv2d_kernel_name: null argument");
    if (int rc = fwd_geometry_check(p)) return rc;
    if (is_small_dense(p)) {
        snprintf(buf, (size_t)buflen, "dense_small_kernel<%d, %s>", igan::dense_small_rows(p->N), p->w_transposed ? "true" : "false");
        return IGAN_OK;
    }
    if (const int kind = igan::thin_conv_kind(p)) {
        snprintf(buf, (size_t)buflen, "%s<%d>", kind == 1 ? "thin_out_kernel" : "thin_in_kernel", p->KH * p->KW);
        return IGAN_OK;
    }
    int Mmax, chunks_max, nclass;
    fwd_counts(p, Mmax, chunks_max, nclass);
    const FwdTile t = pick_fwd_tile(Mmax, p->Cout);
    const bool wt = p->w_transposed != 0;
    const bool vecA = (p->Cin % 4 == 0) && (((uintptr_t)p->x & 15) == 0);
    const bool vecS = (p->Cin % 4 == 0) && (((uintptr_t)p->in_scale & 15) == 0);
    const bool vecB = ((wt ? p->Cin : p->Cout) % 4 == 0) && (((uintptr_t)p->w & 15) == 0);
    const bool vec = vecA && vecB && (p->in_scale == nullptr || vecS);
    if (use_planes_kernel(p, t, Mmax)) {
        snprintf(buf, (size_t)buflen, (planes_mode() == 2 && f16_tap_outer() && f16_w4()) ? "conv_fwd_planes_w4_kernel" : "conv_fwd_planes_kernel");
        return IGAN_OK;
    }
    if (use_dma_kernel(p, t, walk_ok(p))) {
        snprintf(buf, (size_t)buflen, "conv_fwd_dma_kernel<%s, %s>", wt ? "true" : "false", p->in_scale ? "true" : "false");
        return IGAN_OK;
    }
    int wm = 2, wn = 2;
    if (t.BM == 128 && t.BN == 128 && eight_waves("IGAN_CONV_8WAVE")) wn = 4;
    if (t.BM == 128 && t.BN == 64 && eight_waves("IGAN_CONV_8WAVE")) { wm = 4; wn = 2; }
    if (t.BM == 128 && t.BN == 32) { wm = 4; wn = 1; }
    if (t.BM == 32) { wm = 1; wn = 4; }
    snprintf(buf, (size_t)buflen, "conv_fwd_kernel<%d, %d, %d, %d, %s, %s, %s>", t.BM, t.BN, wm, wn, wt ? "true" : "false",
             vec ? "true" : "false", p->in_scale ? "true" : "false");
    return IGAN_OK;
}

extern "C" int igan_conv2d(igan_stream_t stream_, const igan_conv2d_params* p) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(p != nullptr, "conv2d: null params");
    if (int rc = fwd_geometry_check(p)) return rc;
    if (p->x_colmax != nullptr) {       // the by-product of the fp16 form's row image; any other path of this call computes it by a pass of its own, so that the buffer is always valid
        IGAN_REQUIRE(igan_colmax_floats(p->N, p->H * p->W, p->Cin) != 0, "conv2d: x_colmax given, but igan_colmax_floats() says this process / tensor takes none");
        IGAN_REQUIRE((((uintptr_t)p->x_colmax | (uintptr_t)p->x | (uintptr_t)p->in_scale) & 15) == 0, "conv2d: x_colmax, x and in_scale must be 16-byte aligned");
    }
    auto colmax_own_pass = [&]() { if (p->x_colmax) launch_colmax_only(stream, p->x, p->in_scale, p->x_colmax, p->N * p->H * p->W, p->H * p->W, p->Cin); };
    if (is_small_dense(p)) {
        colmax_own_pass();
        IGAN_REQUIRE(p->noise == nullptr, "conv2d: the fused noise epilogue is not offered on the small dense path");
        dense_small(stream, p->x, p->w, p->y, p->N, p->Cin, p->Cout, p->w_transposed != 0, p->alpha);
        IGAN_LAUNCH_CHECK("conv2d dense launch");
        return IGAN_OK;
    }
    if (const int kind = thin_conv_kind(p)) {
        colmax_own_pass();
        IGAN_REQUIRE(p->noise == nullptr, "conv2d: the fused noise epilogue is not offered on the thin-channel path");
        thin_conv(stream, p, kind);
        IGAN_LAUNCH_CHECK("conv2d thin-channel launch");
        return IGAN_OK;
    }
    int Mmax, chunks_max, nclass;
    fwd_counts(p, Mmax, chunks_max, nclass);
    const FwdTile t = pick_fwd_tile(Mmax, p->Cout);
    const TileList l = tile_list(p, t, Mmax, nclass);
    int splits = std::max(1, p->splits);
    const int sliced = std::min(std::max(p->sliced_tiles, 0), l.T);   // trailing tiles cut along the reduction axis
    if (sliced == 0) splits = 1;
    const size_t partial_floats = (splits > 1) ? (size_t)sliced * splits * t.BM * t.BN : 0;
    // the variant needs its piece images' room behind the partial tiles; a caller that did not provide it gets the fp32 kernel
    const bool planes = use_planes_kernel(p, t, Mmax) && p->workspace != nullptr &&
                        p->workspace_floats >= partial_floats + planes_x_floats(p) + planes_w_floats(p);
    if (splits > 1) {
        IGAN_REQUIRE(p->workspace != nullptr, "conv2d: splits > 1 needs a workspace");
        IGAN_REQUIRE(p->workspace_floats >= partial_floats, "conv2d: workspace too small");
    }
    if (splits > 1 || planes) IGAN_REQUIRE((((uintptr_t)p->workspace) & 15) == 0, "conv2d: workspace must be 16-byte aligned");

    ConvArgs a;
    a.x = p->x; a.w = p->w;
    a.y = (splits > 1) ? p->workspace : nullptr;
    a.out = p->y;
    a.in_scale = p->in_scale; a.out_scale = p->out_scale;
    a.N = p->N; a.H = p->H; a.W = p->W; a.Cin = p->Cin;
    a.OH = p->OH; a.OW = p->OW; a.Cout = p->Cout;
    a.KH = p->KH; a.KW = p->KW;
    a.stride = p->stride; a.up_shift = (p->up == 2) ? 1 : 0;
    a.pad_y = p->pad_y; a.pad_x = p->pad_x;
    a.splits = splits;
    a.nx = l.nx; a.ny = l.ny;
    a.full_tiles = (splits > 1) ? l.T - sliced : l.T;
    a.cpt = ceil_div(p->Cin, BK);
    a.Mtot = p->N * p->OH * p->OW;
    a.vecA = (p->Cin % 4 == 0) && (((uintptr_t)p->x & 15) == 0);
    a.vecS = (p->Cin % 4 == 0) && (((uintptr_t)p->in_scale & 15) == 0);
    if (p->w_transposed) a.vecB = (p->Cin % 4 == 0) && (((uintptr_t)p->w & 15) == 0);
    else a.vecB = (p->Cout % 4 == 0) && (((uintptr_t)p->w & 15) == 0);
    a.vecY = (p->Cout % 2 == 0) && ((((uintptr_t)p->y | (uintptr_t)p->out_scale) & 7) == 0);
    a.alpha = p->alpha;
    {   // A/B switch: IGAN_XCD_REMAP=0 keeps the plain block order
        static const bool remap = !(getenv("IGAN_XCD_REMAP") && atoi(getenv("IGAN_XCD_REMAP")) == 0);
        a.xcd_remap = remap ? 1 : 0;
        static const bool walk = !(getenv("IGAN_CONV_WALK") && atoi(getenv("IGAN_CONV_WALK")) == 0);     // A/B switch
        a.diag = g_conv_diag;
        static const int stagger = getenv("IGAN_CONV_STAGGER") ? atoi(getenv("IGAN_CONV_STAGGER")) : 0;     // experiment
        a.stagger = stagger;
        static const bool bscale = !(getenv("IGAN_CONV_BSCALE") && atoi(getenv("IGAN_CONV_BSCALE")) == 0);     // A/B switch
        a.b_scale = bscale ? 1 : 0;
        static const bool pprio = !(getenv("IGAN_CONV_PROLOGUE_PRIO") && atoi(getenv("IGAN_CONV_PROLOGUE_PRIO")) == 0);     // A/B switch
        a.prio = pprio ? 1 : 0;
        a.walk = (walk && a.vecA && a.vecB && (a.in_scale == nullptr || a.vecS) && (p->Cin % BK == 0)) ? 1 : 0;
    }
    a.bias = p->bias; a.act = p->act; a.act_alpha = p->act_alpha; a.act_gain = p->act_gain;
    a.noise = p->act ? p->noise : nullptr; a.noise_strength = p->noise_strength; a.noise_bcast = p->noise_bcast;
    a.xp = nullptr; a.wp = nullptr;
    a.diag_mode = 0;
#ifdef IGAN_DIAGNOSTIC
    a.diag_mode = getenv("IGAN_DIAG_MODE") ? atoi(getenv("IGAN_DIAG_MODE")) : 0;
#endif

    dim3 grid(a.full_tiles + (l.T - a.full_tiles) * splits);
    const bool wt = p->w_transposed != 0;
    const bool vec = a.vecA && a.vecB && (a.in_scale == nullptr || a.vecS);
    bool launched = false;
    if (!planes) colmax_own_pass();
    if (planes) {       // piece form: write the two piece images, then the tile kernel
        IGAN_REQUIRE(p->x_pieces == nullptr || planes_mode() == 1, "conv2d: the two-piece fp16 form writes its own images (one scale per pixel here, per channel in the weight gradient): x_pieces must be NULL");
        IGAN_REQUIRE(p->x_pieces == nullptr || p->x_pieces_bytes == planes_x_floats(p) * 4, "conv2d: x_pieces is not the image of this x (x_pieces_bytes != N*H*W*Cin*6)");
        const unsigned short* xp = reinterpret_cast<const unsigned short*>(p->x_pieces);
        unsigned short* wp = reinterpret_cast<unsigned short*>(p->workspace + partial_floats + planes_x_floats(p));
        const int cpp = p->Cin / PK;
        if (xp == nullptr) {         // no image from the caller: write it behind the partial tiles
            unsigned short* own = reinterpret_cast<unsigned short*>(p->workspace + partial_floats);
            if (a.diag_mode & 1) {}      // DIAGNOSTIC: stale image
            else if (planes_mode() == 2) launch_row_image(stream, p->x, p->in_scale, own, p->N * p->H * p->W, p->H * p->W, p->Cin, p->x_colmax);
            else launch_piece_image(stream, p->x, p->in_scale, own, p->N * p->H * p->W, p->H * p->W, p->Cin);
            xp = own;
        }
        // (the filter image beside the x image on a second stream, forked and joined by events -- two branches inside the captured graphs -- is correct and
        // 5.6 % SLOWER in the bench: 333.9 -> 315.1 img/s, profiles/r05_small_layers.txt section 3)
        if (p->w_pieces != nullptr) {     // ABI v9: a constant filter's image, kept by the caller (igan_filter_image): nothing to write
            IGAN_REQUIRE(p->w_pieces_bytes == igan_filter_image_bytes(p->KH, p->KW, p->Cin, p->Cout) && (((uintptr_t)p->w_pieces) & 15) == 0,
                         "conv2d: w_pieces is not an image igan_filter_image() wrote for this filter (size / alignment)");
            wp = reinterpret_cast<unsigned short*>(const_cast<void*>(p->w_pieces));
        } else if (!(a.diag_mode & 1)) launch_filter_image(stream, p->w, wp, wt, p->KH * p->KW, p->KW, p->Cout, p->Cin);
        a.xp = xp; a.wp = wp;
        a.cpt = cpp;
        if (a.diag_mode & 2) {}          // DIAGNOSTIC: image kernels only
        else if (planes_mode() == 2) {
            // reduction order of the fp16 form (see the kernel): tap outermost (the cross terms are folded once per tap); A/B switch IGAN_F16_TAP_OUTER=0:
            // slice outermost, every step folds its cross terms.  Measured (profiles/r05_f16_rowscale.txt): kernel 490.6 -> 457.2 us on G 128 Conv1
            // (Cin 128), 451.4 -> 428.0 us on G 32 Conv1 (Cin 512); whole layer list forward 186.0 -> 194.4, data gradient 189.2 -> 198.0 TFLOP/s.
            const bool tapo = f16_tap_outer();
            if (tapo && f16_w4()) hipLaunchKernelGGL(conv_fwd_planes_w4_kernel, grid, dim3(256), 0, stream, a);       // the four-wave tile (fragments of the next chunk prefetched into registers): whole calls 4-5 % shorter, bench +1.1 % (profiles/r05_w4_tile.txt)
            else if (tapo) hipLaunchKernelGGL((conv_fwd_planes_kernel<2, true>), grid, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((conv_fwd_planes_kernel<2, false>), grid, dim3(512), 0, stream, a);
        } else hipLaunchKernelGGL((conv_fwd_planes_kernel<3>), grid, dim3(512), 0, stream, a);
        IGAN_LAUNCH_CHECK("conv2d (piece form) launch");
        launched = true;
    } else
    if (use_dma_kernel(p, t, a.walk != 0)) {       // LDS-DMA form of the 128x128 tile
        // (the in-register bf16 split forms of round 2 -- IGAN_CONV_BF16X3, PIECES = 2 / 3 of this kernel -- are no longer instantiated: measure-only, superseded by the
        // piece kernels, and the only kernels of the library beside the 16-wave experiment that spilled to scratch: tools/asm_scan.py, HISTORY.md section 8)
        if (wt) {
            if (a.in_scale) hipLaunchKernelGGL((conv_fwd_dma_kernel<true, true>), grid, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((conv_fwd_dma_kernel<true, false>), grid, dim3(512), 0, stream, a);
        } else {
            if (a.in_scale) hipLaunchKernelGGL((conv_fwd_dma_kernel<false, true>), grid, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((conv_fwd_dma_kernel<false, false>), grid, dim3(512), 0, stream, a);
        }
        IGAN_LAUNCH_CHECK("conv2d (LDS-DMA) launch");
        launched = true;
    }
    if (launched) {}
    else if (t.BM == 128 && t.BN == 128 && eight_waves("IGAN_CONV_8WAVE")) launch_fwd<128, 128, 2, 4>(stream, a, grid, wt, vec);
    else if (t.BM == 128 && t.BN == 128) launch_fwd<128, 128, 2, 2>(stream, a, grid, wt, vec);
    else if (t.BM == 128 && t.BN == 64 && eight_waves("IGAN_CONV_8WAVE")) launch_fwd<128, 64, 4, 2>(stream, a, grid, wt, vec);
    else if (t.BM == 128 && t.BN == 64) launch_fwd<128, 64, 2, 2>(stream, a, grid, wt, vec);
    else if (t.BM == 128 && t.BN == 32) launch_fwd<128, 32, 4, 1>(stream, a, grid, wt, vec);
    else launch_fwd<32, 128, 1, 4>(stream, a, grid, wt, vec);
    IGAN_LAUNCH_CHECK("conv2d launch");

    if (splits > 1) {
        // enough row groups per tile that the fix-up itself fills the machine
        const int rp = std::max(2, std::min(t.BM, t.BM / ceil_div(1024, sliced)));
        hipLaunchKernelGGL(conv_fixup_kernel, dim3(sliced, ceil_div(t.BM, rp)), dim3(256), 0, stream, a, t.BM, t.BN, rp);
        IGAN_LAUNCH_CHECK("conv2d fix-up launch");
    }
    return IGAN_OK;
}

namespace {

int wgrad_geometry_check(const igan_conv2d_wgrad_params* p) {
    IGAN_REQUIRE(p->x && p->dy && p->dw, "conv2d_wgrad: null buffer");
    IGAN_REQUIRE(p->N >= 1 && p->H >= 1 && p->W >= 1 && p->Cin >= 1, "conv2d_wgrad: input dims must be positive");
    IGAN_REQUIRE(p->OH >= 1 && p->OW >= 1 && p->Cout >= 1, "conv2d_wgrad: output dims must be positive");
    IGAN_REQUIRE(p->KH >= 1 && p->KW >= 1, "conv2d_wgrad: kernel must be at least 1x1");
    IGAN_REQUIRE(p->stride >= 1 && p->up >= 1, "conv2d_wgrad: stride and up must be at least 1");
    IGAN_REQUIRE(p->stride == 1 || p->up == 1, "conv2d_wgrad: at most one of stride, up may exceed 1");
    if (!(p->up == 1 || p->up == 2)) return igan::fail(IGAN_ERR_UNSUPPORTED, "conv2d_wgrad: up must be 1 or 2 (got %d)", p->up);
    IGAN_REQUIRE((long long)p->N * p->H * p->W * p->Cin * 4 <= 0x7FFFFFF0LL, "conv2d_wgrad: input too large (2 GiB per operand)");
    IGAN_REQUIRE((long long)p->N * p->OH * p->OW * p->Cout * 4 <= 0x7FFFFFF0LL, "conv2d_wgrad: output gradient too large (2 GiB per operand)");
    IGAN_REQUIRE((long long)p->KH * p->KW * p->Cin * p->Cout <= INT32_MAX, "conv2d_wgrad: filter too large");
    return IGAN_OK;
}

template <int BM, int BN, int WM, int WN, bool VEC>
void launch_wgrad2(hipStream_t stream, const WgradArgs& a, dim3 grid, int scm) {
    if constexpr (BM == 128 && BN == 128 && WM * WN == 8) {      // the eight-wave 128x128 tile exists unscaled only (its scaled forms spilled: the dispatcher sends them to four waves)
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, VEC, 0>), grid, dim3(WM * WN * 64), 0, stream, a);
    } else {
        if (scm == 0) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, VEC, 0>), grid, dim3(WM * WN * 64), 0, stream, a);
        else if (scm == 1) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, VEC, 1>), grid, dim3(WM * WN * 64), 0, stream, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, VEC, 2>), grid, dim3(WM * WN * 64), 0, stream, a);
    }
}
template <int BM, int BN, int WM, int WN>
void launch_wgrad(hipStream_t stream, const WgradArgs& a, dim3 grid, bool vec, int scm) {
    if (vec) launch_wgrad2<BM, BN, WM, WN, true>(stream, a, grid, scm);
    else launch_wgrad2<BM, BN, WM, WN, false>(stream, a, grid, scm);
}

struct WgTile { int BM, BN; };
WgTile pick_wg_tile(int Cin, int Cout) {
    WgTile t;
    t.BM = (Cin > 32) ? 128 : 32;
    t.BN = (Cout > 32) ? 128 : 32;
    if (t.BM == 32 && t.BN == 32) t.BN = 128;  // only three instantiations exist
    return t;
}

int wgrad_splits(const igan_conv2d_wgrad_params* p) {
    // One block per (tap, Cin tile, Cout tile, pixel slice).  The kernel runs 2 workgroups per CU
    // (LDS / VGPR), i.e. 512 co-resident workgroups on 256 CUs: size the grid to whole rounds of 512
    // so that no round runs half empty (a 774-block grid ran 1.5 rounds: 25 % of the machine idle
    // for a third of the time).
    const WgTile t = pick_wg_tile(p->Cin, p->Cout);
    const long long tiles = (long long)igan::ceil_div(p->Cin, t.BM) * igan::ceil_div(p->Cout, t.BN) * p->KH * p->KW;
    const int up = p->up;
    const long long kpix = (long long)p->N * ((p->OH + up - 1) / up) * ((p->OW + up - 1) / up);
    const int chunks = (int)((kpix + BK - 1) / BK);
    const int max_by_work = std::max(1, chunks / 4);          // >= 4 chunks per block
    int s;
    if (tiles >= 512) s = 1;
    else {
        s = (int)(512 / tiles);                               // one full round
        if (s > max_by_work) s = max_by_work;
        else if (chunks / s > 96 && 1024 / tiles <= max_by_work) s = (int)(1024 / tiles);   // long slices: two rounds
    }
    s = std::min(s, 256);
    return std::max(s, 1);
}

// The weight gradient's bf16-piece form (conv_wgrad_planes_kernel): same switch and the same kind of shapes as the forward one --
// 3x3 filters between at least 128 channels on each side (the 128x128 tile), channel counts in whole 32s, a pixel axis of at least
// 1024 (round 6; 2048 before), both piece images below the out-of-range marker.
int wgrad_min_rows() {
    static const int v = getenv("IGAN_WGRAD_PLANES_MIN_ROWS") ? atoi(getenv("IGAN_WGRAD_PLANES_MIN_ROWS")) : 1024;
    return v;
}
bool wgrad_planes_shape_ok(const igan_conv2d_wgrad_params* p) {
    static const bool wg = !(getenv("IGAN_WGRAD_PLANES") && atoi(getenv("IGAN_WGRAD_PLANES")) == 0);      // A/B switch inside the piece form
    if (!planes_enabled() || !wg || p->KH * p->KW == 1 || p->Cin < 128 || p->Cout < 128 || p->Cin % 32 != 0 || p->Cout % 32 != 0) return false;
#ifdef IGAN_DIAGNOSTIC
    {   // DIAGNOSTIC ONLY (bisecting by layer class): IGAN_WGRAD_PLANES_CIN=<n> keeps the piece form for weight gradients with Cin == n only,
        // IGAN_WGRAD_PLANES_KIND=plain|stride|up for stride 1 without up-sampling / stride 2 / up 2 only
        static const int only_cin = getenv("IGAN_WGRAD_PLANES_CIN") ? atoi(getenv("IGAN_WGRAD_PLANES_CIN")) : 0;
        static const char* kind = getenv("IGAN_WGRAD_PLANES_KIND");
        if (only_cin && p->Cin != only_cin) return false;
        if (kind && kind[0] == 'p' && (p->stride != 1 || p->up != 1)) return false;
        if (kind && kind[0] == 's' && p->stride != 2) return false;
        if (kind && kind[0] == 'u' && p->up != 2) return false;
    }
#endif
    if ((long long)p->N * p->OH * p->OW < wgrad_min_rows() * (long long)p->up * p->up) return false;
    if ((long long)p->N * p->H * p->W * p->Cin * 6 >= 0x7FFFFF00LL || (long long)p->N * p->OH * p->OW * p->Cout * 6 >= 0x7FFFFF00LL) return false;
    // fp16 form: the column-maximum pass gives every thread one channel quad (C / 4 a power of two <= 256)
    if (planes_mode() == 2 && (!pow2(p->Cin) || !pow2(p->Cout) || p->Cin > 1024 || p->Cout > 1024)) return false;
    return true;
}
size_t wgrad_planes_x_floats(const igan_conv2d_wgrad_params* p) {
    if (planes_mode() == 2) return cols_part_bytes((size_t)p->N * p->H * p->W, (size_t)p->Cin) / 4;
    return (size_t)p->N * p->H * p->W * p->Cin * 6 / 4;
}
size_t wgrad_planes_dy_floats(const igan_conv2d_wgrad_params* p) {
    if (planes_mode() == 2) return cols_part_bytes((size_t)p->N * p->OH * p->OW, (size_t)p->Cout) / 4;
    return (size_t)p->N * p->OH * p->OW * p->Cout * 6 / 4;
}

}  // namespace

extern "C" int igan_conv2d_wgrad_plan(const igan_conv2d_wgrad_params* p, int* splits, size_t* workspace_floats) {
    IGAN_REQUIRE(p && splits && workspace_floats, "conv2d_wgrad_plan: null argument");
    if (int rc = wgrad_geometry_check(p)) return rc;
    if (is_small_dense_wgrad(p)) { *splits = 1; *workspace_floats = 0; return IGAN_OK; }
    if (const int kind = igan::thin_wgrad_kind(p)) {
        *splits = 2;                           // "uses the workspace"; the pixel blocking is internal
        *workspace_floats = igan::thin_wgrad_workspace(p, kind);
        return IGAN_OK;
    }
    const int s = wgrad_splits(p);
    *splits = s;
    *workspace_floats = (s > 1) ? (size_t)s * p->KH * p->KW * p->Cin * p->Cout : 0;
    if (wgrad_planes_shape_ok(p))       // the piece images of x and dy follow the partial filters
        *workspace_floats += wgrad_planes_x_floats(p) + wgrad_planes_dy_floats(p);
    return IGAN_OK;
}

// Which kernel family igan_conv2d_wgrad() runs for these parameters (host-only, for the profiling tools; ABI v6).
extern "C" int igan_conv2d_wgrad_kernel_name(const igan_conv2d_wgrad_params* p, char* buf, int buflen) {
    using namespace igan;
    IGAN_REQUIRE(p && buf && buflen > 0, "conv2d_wgrad_kernel_name: null argument");
    if (int rc = wgrad_geometry_check(p)) return rc;
    const char* name = "conv_wgrad_kernel";
    if (is_small_dense_wgrad(p)) name = "dense_small_wgrad_kernel";
    else if (thin_wgrad_kind(p)) name = "thin_wgrad_kernel";
    else if (wgrad_planes_shape_ok(p) && (((uintptr_t)p->x | (uintptr_t)p->dy | (uintptr_t)p->in_scale | (uintptr_t)p->out_scale | (uintptr_t)p->x_pieces | (uintptr_t)p->dy_pieces) & 15) == 0)
        name = "conv_wgrad_planes_kernel";
    snprintf(buf, (size_t)buflen, "%s", name);
    return IGAN_OK;
}

extern "C" int igan_conv2d_wgrad(igan_stream_t stream_, const igan_conv2d_wgrad_params* p) {
    using namespace igan;
    hipStream_t stream = (hipStream_t)stream_;
    IGAN_REQUIRE(p != nullptr, "conv2d_wgrad: null params");
    if (int rc = wgrad_geometry_check(p)) return rc;
    if (is_small_dense_wgrad(p)) {
        dense_small_wgrad(stream, p->x, p->dy, p->dw, p->N, p->Cin, p->Cout, p->alpha);
        IGAN_LAUNCH_CHECK("conv2d_wgrad dense launch");
        return IGAN_OK;
    }
    if (const int kind = thin_wgrad_kind(p)) {
        if (p->workspace && p->workspace_floats >= thin_wgrad_workspace(p, kind) && (((uintptr_t)p->workspace) & 15) == 0) {
            thin_wgrad(stream, p, kind);
            IGAN_LAUNCH_CHECK("conv2d_wgrad thin-channel launch");
            return IGAN_OK;
        }
    }
    const int splits = std::max(1, p->splits);
    const size_t wsize = (size_t)p->KH * p->KW * p->Cin * p->Cout;
    if (splits > 1) {
        IGAN_REQUIRE(p->workspace != nullptr, "conv2d_wgrad: splits > 1 needs a workspace");
        IGAN_REQUIRE(p->workspace_floats >= (size_t)splits * wsize, "conv2d_wgrad: workspace too small");
    }
    if (p->x_colmax != nullptr) IGAN_REQUIRE(igan_colmax_floats(p->N, p->H * p->W, p->Cin) != 0 && (((uintptr_t)p->x_colmax) & 15) == 0, "conv2d_wgrad: x_colmax is not a buffer igan_colmax_floats() sized for this x");
    if (p->dy_colmax != nullptr) IGAN_REQUIRE(igan_colmax_floats(p->N, p->OH * p->OW, p->Cout) != 0 && (((uintptr_t)p->dy_colmax) & 15) == 0, "conv2d_wgrad: dy_colmax is not a buffer igan_colmax_floats() sized for this dy");
    WgradArgs a;
    a.x = p->x; a.dy = p->dy;
    a.out = (splits > 1) ? p->workspace : p->dw;
    a.in_scale = p->in_scale; a.out_scale = p->out_scale;
    a.N = p->N; a.H = p->H; a.W = p->W; a.Cin = p->Cin;
    a.OH = p->OH; a.OW = p->OW; a.Cout = p->Cout;
    a.KH = p->KH; a.KW = p->KW;
    a.stride = p->stride; a.up_shift = (p->up == 2) ? 1 : 0;
    a.pad_y = p->pad_y; a.pad_x = p->pad_x;
    a.splits = splits;
    a.vecA = (p->Cin % 4 == 0) && (((uintptr_t)p->x & 15) == 0);
    a.vecB = (p->Cout % 4 == 0) && (((uintptr_t)p->dy & 15) == 0);
    a.vecSA = (p->Cin % 4 == 0) && (((uintptr_t)p->in_scale & 15) == 0);
    a.vecSB = (p->Cout % 4 == 0) && (((uintptr_t)p->out_scale & 15) == 0);

    a.vecY = (p->Cout % 2 == 0) && (((uintptr_t)a.out & 7) == 0);
    a.alpha = p->alpha;
    {
        static const bool remap = !(getenv("IGAN_XCD_REMAP") && atoi(getenv("IGAN_XCD_REMAP")) == 0);
        a.xcd_remap = remap ? 1 : 0;
    }

    const WgTile t = pick_wg_tile(p->Cin, p->Cout);
    dim3 grid(ceil_div(p->Cin, t.BM), ceil_div(p->Cout, t.BN), p->KH * p->KW * splits);
    a.xp = nullptr; a.dyp = nullptr;
    const size_t partial_floats = (splits > 1) ? (size_t)splits * wsize : 0;
    if (wgrad_planes_shape_ok(p) && p->workspace != nullptr && (((uintptr_t)p->workspace | (uintptr_t)p->x | (uintptr_t)p->dy | (uintptr_t)p->in_scale | (uintptr_t)p->out_scale | (uintptr_t)p->x_pieces | (uintptr_t)p->dy_pieces) & 15) == 0 &&
        p->workspace_floats >= partial_floats + wgrad_planes_x_floats(p) + wgrad_planes_dy_floats(p)) {
        IGAN_REQUIRE((p->x_pieces == nullptr && p->dy_pieces == nullptr) || planes_mode() == 1, "conv2d_wgrad: the two-piece fp16 form writes its own images (one scale per channel): x_pieces / dy_pieces must be NULL");
        IGAN_REQUIRE(p->x_pieces == nullptr || p->x_pieces_bytes == wgrad_planes_x_floats(p) * 4, "conv2d_wgrad: x_pieces is not the image of this x (x_pieces_bytes != N*H*W*Cin*6)");
        IGAN_REQUIRE(p->dy_pieces == nullptr || p->dy_pieces_bytes == wgrad_planes_dy_floats(p) * 4, "conv2d_wgrad: dy_pieces is not the image of this dy (dy_pieces_bytes != N*OH*OW*Cout*6)");
        const unsigned short* xp = reinterpret_cast<const unsigned short*>(p->x_pieces);
        const unsigned short* dyp = reinterpret_cast<const unsigned short*>(p->dy_pieces);
        if (xp == nullptr) {
            unsigned short* own = reinterpret_cast<unsigned short*>(p->workspace + partial_floats);
            if (planes_mode() == 2) launch_col_image(stream, p->x, p->in_scale, own, p->N * p->H * p->W, p->H * p->W, p->Cin, p->x_colmax);
            else launch_piece_image(stream, p->x, p->in_scale, own, p->N * p->H * p->W, p->H * p->W, p->Cin);
            xp = own;
        }
        if (dyp == nullptr) {
            unsigned short* own = reinterpret_cast<unsigned short*>(p->workspace + partial_floats + wgrad_planes_x_floats(p));
            if (planes_mode() == 2) launch_col_image(stream, p->dy, p->out_scale, own, p->N * p->OH * p->OW, p->OH * p->OW, p->Cout, p->dy_colmax);
            else launch_piece_image(stream, p->dy, p->out_scale, own, p->N * p->OH * p->OW, p->OH * p->OW, p->Cout);
            dyp = own;
        }
        a.xp = xp; a.dyp = dyp;
        if (planes_mode() == 2) hipLaunchKernelGGL((conv_wgrad_planes_kernel<2>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((conv_wgrad_planes_kernel<3>), grid, dim3(512), 0, stream, a);
        IGAN_LAUNCH_CHECK("conv2d_wgrad (piece form) launch");
        if (splits > 1) {
            const int total = (int)wsize;
            const int rg = std::min(ceil_div(total, 256), 2048);
            hipLaunchKernelGGL(plain_reduce_kernel, dim3(rg), dim3(256), 0, stream, (const float*)p->workspace, p->dw, total, splits, p->alpha);
            IGAN_LAUNCH_CHECK("conv2d_wgrad reduce launch");
        }
        return IGAN_OK;
    }
    const bool vec = a.vecA && a.vecB && (a.in_scale == nullptr || a.vecSA) && (a.out_scale == nullptr || a.vecSB);
    const int scm = (a.in_scale && a.out_scale) ? 1 : ((a.in_scale || a.out_scale) ? 2 : 0);
    // 8 waves pay on the short pixel axes (32x32 and below: +1.5 %), 4 waves on the 128x128 layers (+3-4 %): measured, tools/conv_bench.py
    const bool long_axis = (long long)p->OH * p->OW >= 128LL * 128LL;
    // (round 6) the scaled forms (scm != 0: the modulated layers below the piece form's row threshold) take the four-wave tile: at the 128 registers of eight waves they
    // spill ten registers to scratch inside the loop (tools/asm_scan.py), at 256 they do not
    if (t.BM == 128 && t.BN == 128 && eight_waves("IGAN_WGRAD_8WAVE") && !long_axis && scm == 0) launch_wgrad<128, 128, 2, 4>(stream, a, grid, vec, scm);
    else if (t.BM == 128 && t.BN == 128) launch_wgrad<128, 128, 2, 2>(stream, a, grid, vec, scm);
    else if (t.BM == 128 && t.BN == 32) launch_wgrad<128, 32, 4, 1>(stream, a, grid, vec, scm);
    else launch_wgrad<32, 128, 1, 4>(stream, a, grid, vec, scm);
    IGAN_LAUNCH_CHECK("conv2d_wgrad launch");
    if (splits > 1) {
        const int total = (int)wsize;
        const int rg = std::min(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(plain_reduce_kernel, dim3(rg), dim3(256), 0, stream, (const float*)p->workspace, p->dw, total, splits, p->alpha);
        IGAN_LAUNCH_CHECK("conv2d_wgrad reduce launch");
    }
    return IGAN_OK;
}

// Diagnostic (two-piece fp16 variant): elements imaged so far below the exact window / in all; synchronises the device.  reset != 0 zeroes both.
static int f16_window_read(unsigned long long (&v)[4], int reset) {
    using namespace igan;
    if (hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(g_f16_below_window), 8) != hipSuccess || hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(g_f16_imaged), 8) != hipSuccess ||
        hipMemcpyFromSymbol(&v[2], HIP_SYMBOL(g_f16_below_window_cols), 8) != hipSuccess || hipMemcpyFromSymbol(&v[3], HIP_SYMBOL(g_f16_imaged_cols), 8) != hipSuccess) return IGAN_ERR_HIP;
    if (reset) {
        const unsigned long long z = 0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16_below_window), &z, 8) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_f16_imaged), &z, 8) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(g_f16_below_window_cols), &z, 8) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_f16_imaged_cols), &z, 8) != hipSuccess) return IGAN_ERR_HIP;
    }
    return IGAN_OK;
}
extern "C" int igan_debug_f16_window(unsigned long long* below, unsigned long long* imaged, int reset) {      // both kinds of image together
    unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
    if (int rc = f16_window_read(v, reset)) return rc;
    if (below) *below = v[0] + v[2];
    if (imaged) *imaged = v[1] + v[3];
    return IGAN_OK;
}
extern "C" int igan_debug_f16_window_by_kind(unsigned long long* out4, int reset) {      // ABI v8: {rows below, rows imaged, columns below, columns imaged}
    using namespace igan;
    IGAN_REQUIRE(out4 != nullptr, "debug_f16_window_by_kind: null argument");
    unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
    if (int rc = f16_window_read(v, reset)) return rc;
    for (int i = 0; i < 4; i++) out4[i] = v[i];
    return IGAN_OK;
}

extern "C" int igan_conv_piece_form(void) { return planes_mode(); }

// ABI v8: floats of the buffer a caller passes as x_colmax (igan_conv2d) and hands on as x_colmax / dy_colmax (igan_conv2d_wgrad) for a tensor [N, HW, C]; 0 = this
// process / tensor has no use for one (not the fp16 form; a channel count the column image does not take).
extern "C" size_t igan_colmax_floats(int N, int HW, int C) {
    using namespace igan;
    if (planes_mode() != 2 || N < 1 || HW < 1 || C < 16 || C > 1024 || (C & (C - 1)) != 0) return 0;
    // no weight gradient of a tensor with fewer pixels than the piece form's row threshold takes the column image (wgrad_planes_shape_ok: N * OH * OW >= threshold * up^2,
    // and an up-sampling layer's input has a quarter of its output's pixels): the maxima would be two dead kernels and up to 2 MB of saved state per small layer
    if ((long long)N * HW < wgrad_min_rows()) return 0;
    return 4 + (size_t)H_COLBLOCKS * C;
}

// ABI v9: the filter image as a caller-kept buffer (constant weights: the LPIPS network's 13 filters are imaged 36 times per generator step otherwise).
// The conditions are the filter's share of planes_shape_ok(); whether a given call takes the piece form also depends on its rows -- a call that does not ignores the image.
extern "C" size_t igan_filter_image_bytes(int KH, int KW, int Cin, int Cout) {
    using namespace igan;
    if (!planes_enabled() || KH < 1 || KW < 1 || Cin < 1 || Cout < 1 || KH * KW == 1 || Cin % BK != 0 || (long long)KH * KW * Cin < 1152) return 0;
    if ((long long)KH * KW * Cin * Cout * 6 >= 0x7FFFFF00LL) return 0;
    if (planes_mode() == 2) {
        if (!pow2(Cin / PK) || Cin / PK > 64 || KH * KW > 9 || Cout % 4 != 0) return 0;
        return filter_part_bytes((size_t)KH * KW, (size_t)Cout, (size_t)Cin);
    }
    return (size_t)KH * KW * Cin * Cout * 6;
}

extern "C" int igan_filter_image(igan_stream_t stream_, const float* w, void* out, int KH, int KW, int Cin, int Cout, int w_transposed) {
    using namespace igan;
    IGAN_REQUIRE(w != nullptr && out != nullptr, "filter_image: null buffer");
    IGAN_REQUIRE(igan_filter_image_bytes(KH, KW, Cin, Cout) != 0, "filter_image: igan_filter_image_bytes() says this process / filter takes none");
    IGAN_REQUIRE(((((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0, "filter_image: w and out must be 16-byte aligned");
    launch_filter_image((hipStream_t)stream_, w, reinterpret_cast<unsigned short*>(out), w_transposed != 0, KH * KW, KW, Cout, Cin);
    IGAN_LAUNCH_CHECK("filter_image launch");
    return IGAN_OK;
}

extern "C" int igan_conv_pieces_wanted(int KH, int KW, int Cin, int Cout) {
    using namespace igan;
    return (planes_enabled() && KH * KW > 1 && Cin >= 128 && Cout >= 128 && Cin % 32 == 0 && Cout % 32 == 0) ? 1 : 0;
}

extern "C" int igan_pieces_image_ok(int N, int HW, int C) {
    using namespace igan;
    // only the bf16 form shares an image between calls: the fp16 form scales a tensor per pixel for the forward / data-gradient kernel and per channel for
    // the weight gradient, so every call writes the image it needs
    return (planes_mode() == 1 && C >= 128 && C % 32 == 0 && (long long)N * HW >= 2048 && (long long)N * HW * C * 6 < 0x7FFFFF00LL) ? 1 : 0;
}

extern "C" int igan_to_pieces(igan_stream_t stream_, const float* x, const float* scale, void* out, int N, int HW, int C) {
    using namespace igan;
    IGAN_REQUIRE(x != nullptr && out != nullptr, "to_pieces: null buffer");
    IGAN_REQUIRE(N >= 1 && HW >= 1 && C >= 16 && C % 16 == 0, "to_pieces: C must be a positive multiple of 16");
    IGAN_REQUIRE((((uintptr_t)x | (uintptr_t)scale | (uintptr_t)out) & 15) == 0, "to_pieces: buffers must be 16-byte aligned");
    IGAN_REQUIRE((long long)N * HW * C * 6 < 0x7FFFFF00LL, "to_pieces: image too large (32-bit offsets)");
    if (planes_mode() != 1) return igan::fail(IGAN_ERR_UNSUPPORTED, "to_pieces: only the bf16-piece form (IGAN_CONV_PLANES=1) takes caller-written images (igan_pieces_image_ok() says so)");
    launch_piece_image((hipStream_t)stream_, x, scale, reinterpret_cast<unsigned short*>(out), N * HW, HW, C);
    IGAN_LAUNCH_CHECK("to_pieces launch");
    return IGAN_OK;
}
