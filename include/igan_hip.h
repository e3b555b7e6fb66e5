/*
 * igan_hip.h -- C ABI of libigan_hip.so: the MI355X (gfx950) kernels behind the
 * InclusiveGAN G/D forward+backward hot path.
 *
 * This is the drop-in boundary for the reference's two native plugin mechanisms:
 *   (1) the TF custom ops loaded by dnnlib/tflib/custom_ops.py:87-167
 *       (UpFirDn2D  -> dnnlib/tflib/ops/upfirdn_2d.cu:310-324,
 *        FusedBiasAct -> dnnlib/tflib/ops/fused_bias_act.cu:174-186), and
 *   (2) the convolution / matmul / reduction / optimizer arithmetic the reference
 *       obtains from TensorFlow+cuDNN (networks_stylegan2.py:46,60,120,
 *       upfirdn_2d.py:291,332, optimizer.py:318-332) and the nearest-neighbour
 *       search it obtains from dci_code (dci.h:76-89).
 *
 * Conventions (same contract as the reference ops):
 *   - plain pointers and sizes only; no torch / TF types;
 *   - the CALLER owns every buffer (inputs, outputs, workspaces); nothing is
 *     allocated, retained or freed by the library;
 *   - every entry point enqueues work on the given hipStream_t and returns
 *     without synchronising (graph-capturable, re-entrant);
 *   - return value 0 == IGAN_OK; on failure a thread-local message is available
 *     from igan_last_error() (invalid arguments mirror the reference's
 *     OP_REQUIRES checks, e.g. upfirdn_2d.cu:228-229,241-244,252,256,266);
 *   - element counts are limited to int32 like the reference (upfirdn_2d.cu:243); operands the
 *     conv / 1-NN kernels read through buffer descriptors are limited to 2 GiB each.
 *   - all activation tensors are fp32, channel-minor ("NHWC"): [N, H, W, C].
 */
#ifndef IGAN_HIP_H
#define IGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout, a signature or the set of exports changes (2: fused conv epilogue fields, fp64 nearest-neighbour state;
 * 5 / 6: piece images and their sizes; 7: igan_conv_piece_form, igan_debug_f16_window; 8: the two-piece fp16 form scales every tensor per
 * pixel (forward / data gradient) or per channel (weight gradient) and writes its own images -- caller-written images (igan_to_pieces, x_pieces,
 * dy_pieces) belong to the bf16-piece form only; igan_conv2d_params gains x_colmax, igan_conv2d_wgrad_params x_colmax / dy_colmax at their ends;
 * 9: igan_conv2d_params gains w_pieces / w_pieces_bytes -- a caller-kept FILTER image for weights that never change (igan_filter_image_bytes, igan_filter_image)). */
#define IGAN_ABI_VERSION 9

typedef void* igan_stream_t; /* hipStream_t */

enum igan_status {
    IGAN_OK = 0,
    IGAN_ERR_INVALID_ARGUMENT = 1, /* reference: errors::InvalidArgument */
    IGAN_ERR_HIP = 2,              /* reference: errors::Internal(cudaGetErrorName) */
    IGAN_ERR_UNSUPPORTED = 3
};

int igan_abi_version(void);
/* sizeof() of the parameter structs as this library was compiled, by id: 0 upfirdn2d, 1 fused_bias_act, 2 conv2d,
 * 3 conv2d_wgrad, 4 dense, 5 dense_wgrad, 6 taps; 0 for an unknown id.  A binding checks these against its own
 * struct declarations at load time, so that a stale library can never be driven with a newer struct (or vice versa). */
size_t igan_struct_size(int which);
const char* igan_last_error(void);

/* ------------------------------------------------------------------------
 * upfirdn2d: pad -> zero-insert upsample -> FIR -> decimate.
 * Replaces UpFirDn2DOp<T>::Compute + UpFirDn2DKernel_{small,large}
 * (dnnlib/tflib/ops/upfirdn_2d.cu:33-59,64-207,232-307).
 *   y[m,oy,ox,c] = sum_{ky,kx} xup_pad[m, oy*downy+ky, ox*downx+kx, c] * k[kH-1-ky, kW-1-kx]
 * outW/outH are derived exactly as upfirdn_2d.cu:254-255 and must match.
 * `k` is a HOST pointer (<= 64 taps; the reference wrapper always builds the
 * filter from a NumPy constant, upfirdn_2d.py:123-124); it is copied into the
 * kernel arguments, so the call stays graph-capturable.
 */
typedef struct igan_upfirdn2d_params {
    const float* x; /* [majorDim, inH, inW, minorDim] device */
    const float* k; /* [kernelH, kernelW] HOST */
    float* y;       /* [majorDim, outH, outW, minorDim] device */
    int upx, upy, downx, downy;
    int padx0, padx1, pady0, pady1;
    int majorDim, inH, inW, minorDim;
    int kernelH, kernelW;
    int outH, outW;
} igan_upfirdn2d_params;

int igan_upfirdn2d(igan_stream_t stream, const igan_upfirdn2d_params* p);
/* The half instantiation the reference registers next to float (REGISTER_KERNEL_BUILDER ... Eigen::half, upfirdn_2d.cu:323-324):
 * `x` and `y` hold IEEE binary16 values behind the float-typed fields (cast the pointers); `k` stays a host float array and is
 * rounded to half like the reference's `k: T` input; loads widen to float, the accumulation is float, the store rounds to half
 * (upfirdn_2d.cu:101,114).  General kernel only (no configuration of the training path runs in half). */
int igan_upfirdn2d_f16(igan_stream_t stream, const igan_upfirdn2d_params* p);
/* The same with the synthesis layer's epilogue fused into the store (layers whose FIR sits between the up-convolution and the
 * bias, networks_stylegan2.py:349-357 with up=True):  y = act(upfirdn(x) + noise[m, oy, ox] * strength[0] + bias[c]) * gain,
 * act in {1 linear, 2 relu, 3 lrelu} as igan_bias_act_noise_*; noise [majorDim or 1 (noise_bcast), outH, outW] or NULL with
 * strength.  Offered on the FIR fast path only (up = down = 1, taps <= 4x4, minorDim % 4 == 0); IGAN_ERR_UNSUPPORTED otherwise. */
int igan_upfirdn2d_ban(igan_stream_t stream, const igan_upfirdn2d_params* p, const float* noise, const float* strength,
                       int noise_bcast, const float* bias, int act, float alpha, float gain);

/* ------------------------------------------------------------------------
 * fused_bias_act: y = act(x + b[(i / stepB) % sizeB]) * gain  (grad == 0) and its
 * first / second derivative forms (grad == 1, 2) selected by act*10+grad.
 * Replaces FusedBiasActOp<T>::Compute + FusedBiasActKernel
 * (dnnlib/tflib/ops/fused_bias_act.cu:22-40,42-116,139-171).
 * act index follows fused_bias_act.py:20-30 (1 linear, 2 relu, 3 lrelu, 4 tanh,
 * 5 sigmoid, 6 elu, 7 selu, 8 softplus, 9 swish).
 */
typedef struct igan_fused_bias_act_params {
    const float* x;   /* [sizeX] */
    const float* b;   /* [sizeB] or NULL */
    const float* ref; /* [sizeX] or NULL (required iff grad != 0) */
    float* y;         /* [sizeX] */
    int grad;
    int act;
    float alpha;
    float gain;
    int sizeX;
    int sizeB;
    int stepB;
} igan_fused_bias_act_params;

int igan_fused_bias_act(igan_stream_t stream, const igan_fused_bias_act_params* p);
/* The half instantiation (fused_bias_act.cu:185-186): x / b / ref / y hold IEEE binary16 values behind the float-typed fields;
 * every element is widened to float, evaluated with the same table, and rounded on store (fused_bias_act.cu:56-61,113). */
int igan_fused_bias_act_f16(igan_stream_t stream, const igan_fused_bias_act_params* p);

/* Bias gradient: db[c] = sum over all i with (i / stepB) % sizeB == c of dx[i].
 * Replaces the TF reduce_sum pair of fused_bias_act.py:137-146.
 * Deterministic (fixed reduction order). `partial` is a caller-owned workspace of
 * igan_bias_grad_workspace_floats(sizeX, sizeB, stepB) floats. */
size_t igan_bias_grad_workspace_floats(int sizeX, int sizeB, int stepB);
int igan_bias_grad(igan_stream_t stream, const float* dx, float* db, float* partial,
                   int sizeX, int sizeB, int stepB);

/* Fused synthesis-layer epilogue (training/networks_stylegan2.py:351-357) on channel-minor data
 * x[rows = N*H*W][C], C % 4 == 0:
 *     y = act(x + noise[row] * (*strength) + b[c]) * gain          act in {1 linear, 2 relu, 3 lrelu}
 * noise/strength NULL -> plain apply_bias_act (networks_stylegan2.py:66-68).  The backward produces, in
 * ONE pass over (dy, y):  dx = dy * gain * act'(y)   (FusedBiasAct grad=1 with ref = y),
 *     db[c] = sum_rows dx   (fused_bias_act.py:137-146),   *dstrength = sum_rows noise[row] * sum_c dx.
 * db / dstrength may be NULL when not wanted.  Deterministic; `workspace` holds
 * igan_bias_act_noise_workspace_floats(rows, C) floats. */
size_t igan_bias_act_noise_workspace_floats(int rows, int C);
int igan_bias_act_noise_fwd(igan_stream_t stream, const float* x, const float* noise, const float* strength,
                            const float* b, float* y, int rows, int C, int act, float alpha, float gain);
int igan_bias_act_noise_bwd(igan_stream_t stream, const float* dy, const float* y, const float* noise,
                            float* dx, float* db, float* dstrength, float* workspace,
                            int rows, int C, int act, float alpha, float gain);

/* The same backward when the epilogue was fused into a modulated convolution (igan_conv2d with act / noise / out_scale:
 * y = act(d[n,c] * z + noise * strength + b) * gain, z never stored): additionally
 *     dd[n][c] = sum_{pixels of n} dx * z,      z = (pre - b[c] - noise * strength) / d[n][c],
 * the demodulation gradient of modulated_conv2d_layer (networks_stylegan2.py:105-107,126), with the pre-activation value
 * recovered from y (act 1 linear or 3 lrelu only).  y, dy, dx: [N, HW, C]; dscale = d [N, C]; noise [N or 1, HW] with
 * noise_bcast; workspace of igan_bias_act_noise_dd_workspace_floats(N, HW, C) floats.  One pass + one small reduce. */
size_t igan_bias_act_noise_dd_workspace_floats(int N, int HW, int C);
int igan_bias_act_noise_bwd_dd(igan_stream_t stream, const float* dy, const float* y, const float* noise, const float* strength,
                               const float* b, const float* dscale, float* dx, float* db, float* dstrength, float* dd,
                               float* workspace, int noise_bcast, int N, int HW, int C, int act, float alpha, float gain);

/* ------------------------------------------------------------------------
 * conv2d (implicit GEMM on the matrix cores, fp32 sums).  Three arithmetic forms, chosen once per process by IGAN_CONV_PLANES
 * (igan_conv_piece_form() tells which): 0 = every convolution on the fp32 matrix instruction (exact fp32 FMA chain); 1 = the large 3x3 layers from
 * three bf16 pieces per operand (all 24 significand bits, six products); 2 (default) = the large 3x3 layers from two fp16 pieces per operand
 * (<= 1 ulp of the fp32 operand, three products) under power-of-two scales that follow every axis a matrix-instruction chain does not sum over:
 * one per pixel of x in_scale and one per output channel of the filter in the forward / data-gradient kernel, one per channel of each operand in
 * the weight gradient.  Small, 1x1 and thin layers always run on the fp32 instruction.
 * One entry point covers what the reference gets from tf.nn.conv2d (SAME stride 1,
 * networks_stylegan2.py:60,120; VALID stride 2, upfirdn_2d.py:332),
 * tf.nn.conv2d_transpose (VALID stride 2, upfirdn_2d.py:291), tf.matmul
 * (networks_stylegan2.py:46, as a 1x1 conv on [N,1,1,C]) and their input
 * gradients:
 *
 *   y[n,oy,ox,co] = out_scale[n,co] * sum_{ky,kx,ci}
 *        xup[n, oy*stride + ky - pad_y, ox*stride + kx - pad_x, ci] * in_scale[n,ci] * W(ky,kx,ci,co)
 *   xup[n,v,u,ci] = x[n, v/up, u/up, ci] if v%up==0 && u%up==0 && in range else 0
 *
 * (cross-correlation, like tf.nn.conv2d).  w_transposed == 0: W(ky,kx,ci,co) =
 * w[ky][kx][ci][co] (HWIO, networks_stylegan2.py:23).  w_transposed == 1: the
 * buffer is the FORWARD layer's HWIO weight [KH][KW][Cout][Cin] and
 * W(ky,kx,ci,co) = w[KH-1-ky][KW-1-kx][co][ci] (spatial flip + channel swap) --
 * i.e. the data-gradient of a forward conv with (stride,up,pad) is this same entry
 * point called with (stride'=up, up'=stride, pad'=K-1-pad, w_transposed=1).
 * At most one of stride, up may exceed 1.  in_scale / out_scale (optional)
 * implement StyleGAN2 modulation / demodulation in the non-fused form of
 * networks_stylegan2.py:112,126.
 *
 * Tail slicing: the output is a list of tiles dealt to the CUs; when the list does not fill a whole number
 * of rounds, the last `sliced_tiles` tiles are cut into `splits` slices along the reduction axis (taps x Cin)
 * so that the last round occupies every CU.  Their partial tiles go to `workspace` (tile-compact,
 * sliced_tiles * splits tiles) and a fix-up kernel sums them in fixed order (bit-reproducible).
 * igan_conv2d_plan() chooses `splits`, `sliced_tiles` and the workspace size for a shape; pass them back.
 */
typedef struct igan_conv2d_params {
    const float* x;         /* [N, H, W, Cin] */
    const float* w;         /* see above */
    float* y;               /* [N, OH, OW, Cout] */
    const float* in_scale;  /* [N, Cin] or NULL */
    const float* out_scale; /* [N, Cout] or NULL */
    float* workspace;       /* igan_conv2d_plan()'s workspace_floats floats, 16-byte aligned; NULL iff the plan asked for none.  It
                             * holds the partial tiles of the sliced tail (splits > 1) and, for the shapes the library runs in a
                             * piece form (forms 1 and 2 above), the piece images of x and of the filter behind them (with their
                             * scales in form 2); a launch that gets no room for the images runs the fp32 kernel */
    size_t workspace_floats;
    int N, H, W, Cin;
    int OH, OW, Cout;
    int KH, KW;
    int stride, up;
    int pad_y, pad_x;
    int w_transposed;
    int splits;             /* reduction slices of each sliced tile (from igan_conv2d_plan; 1 = none) */
    int sliced_tiles;       /* how many trailing tiles of the tile list are sliced (from igan_conv2d_plan) */
    float alpha;            /* y is multiplied by alpha (the layers' runtime weight scale, networks_stylegan2.py:30-36,
                             * rides here instead of in a separate w * coef pass); 1.0f for a plain convolution */
    const float* bias;      /* fused epilogue (act != 0): y = act(y + bias[co]) * act_gain, the apply_bias_act that follows the
                             * convolution (networks_stylegan2.py:66-68); bias may be NULL */
    int act;                /* 0 = no epilogue; 1 linear, 2 relu, 3 lrelu (as igan_bias_act_noise_*) */
    float act_alpha, act_gain;
    const float* noise;     /* fused epilogue of a synthesis layer (networks_stylegan2.py:351-357): with act != 0,
                             * y = act(y + noise[n, oy, ox] * noise_strength[0] + bias[co]) * act_gain; NULL = no noise */
    const float* noise_strength; /* device scalar */
    int noise_bcast;        /* 1: noise is [1, OH, OW], shared by the batch (the layer's stored noise); 0: [N, OH, OW] */
    const void* x_pieces;   /* bf16-piece form (IGAN_CONV_PLANES=1) only: the piece image of x * in_scale, written by igan_to_pieces(),
                             * so that a caller who runs several convolutions on one tensor writes its image once.  NULL otherwise:
                             * the fp16 form rejects a non-NULL image (ABI v8: its images are scaled per pixel here and per channel in
                             * the weight gradient, so no image serves two calls), form 0 ignores it */
    size_t x_pieces_bytes;  /* its size, N * H * W * Cin * 6 (ABI v6): an image of any other size is rejected, never read */
    float* x_colmax;        /* ABI v8, fp16 form, optional OUTPUT (igan_colmax_floats(N, H * W, Cin) floats, 16-byte aligned; NULL = none): the per-channel maxima of
                             * |x * in_scale| -- a by-product of the row image this call writes of x (a pass of its own when the call takes another path), which a later
                             * igan_conv2d_wgrad() of the SAME tensor takes as x_colmax / dy_colmax instead of a pass of its own over it */
    const void* w_pieces;   /* ABI v9, piece forms, optional: the image of THIS call's filter (same w, KH, KW, Cin, Cout, w_transposed) written earlier by
                             * igan_filter_image() -- for weights that are constants of the run (the LPIPS network's: 36 filter images per generator step
                             * otherwise); the call then writes none.  NULL = the call images its filter itself.  Ignored by calls that run on the fp32 instruction */
    size_t w_pieces_bytes;  /* its size, igan_filter_image_bytes(KH, KW, Cin, Cout): an image of any other size is rejected, never read */
} igan_conv2d_params;

int igan_conv2d_plan(const igan_conv2d_params* p, int* splits, int* sliced_tiles, size_t* workspace_floats);
int igan_conv2d(igan_stream_t stream, const igan_conv2d_params* p);
/* Host-only: name of the kernel instantiation igan_conv2d() launches for these parameters (as reported
 * by rocprofv3, minus the anonymous-namespace prefix).  For profiling tools. */
int igan_conv2d_kernel_name(const igan_conv2d_params* p, char* buf, int buflen);

/* Weight gradient of the op above (same geometry fields):
 *   dw[ky,kx,ci,co] = sum_{n,oy,ox} xup[n, oy*stride+ky-pad_y, ox*stride+kx-pad_x, ci]
 *                                   * in_scale[n,ci] * dy[n,oy,ox,co] * out_scale[n,co]
 * written in HWIO [KH][KW][Cin][Cout].  The pixel axis is always reduced through
 * the caller's workspace in fixed order (bit-reproducible);
 * igan_conv2d_wgrad_plan() returns the split count and workspace size (the partial filters of the pixel
 * slices; in the piece forms also the piece images of x and dy, as for igan_conv2d). */
typedef struct igan_conv2d_wgrad_params {
    const float* x;         /* [N, H, W, Cin] */
    const float* dy;        /* [N, OH, OW, Cout] */
    float* dw;              /* [KH, KW, Cin, Cout] */
    const float* in_scale;  /* [N, Cin] or NULL */
    const float* out_scale; /* [N, Cout] or NULL */
    float* workspace;
    size_t workspace_floats;
    int N, H, W, Cin;
    int OH, OW, Cout;
    int KH, KW;
    int stride, up;
    int pad_y, pad_x;
    int splits;
    float alpha;            /* dw is multiplied by alpha (see igan_conv2d_params) */
    const void* x_pieces;   /* bf16-piece form (IGAN_CONV_PLANES=1) only: piece images of x * in_scale and dy * out_scale; NULL otherwise (as igan_conv2d_params) */
    const void* dy_pieces;
    size_t x_pieces_bytes;  /* N * H * W * Cin * 6 and N * OH * OW * Cout * 6 (ABI v6): checked before an image is read */
    size_t dy_pieces_bytes;
    const float* x_colmax;  /* ABI v8, fp16 form, optional INPUTS (NULL = none): the buffers an igan_conv2d() call on the same x * in_scale / dy * out_scale filled (x_colmax there); */
    const float* dy_colmax; /* the column image of that operand then skips its maxima pass.  Bit-identical results either way (a maximum does not depend on the order). */
} igan_conv2d_wgrad_params;

int igan_conv2d_wgrad_plan(const igan_conv2d_wgrad_params* p, int* splits, size_t* workspace_floats);
int igan_conv2d_wgrad(igan_stream_t stream, const igan_conv2d_wgrad_params* p);
/* Host-only (ABI v6): the kernel family igan_conv2d_wgrad() runs for these parameters ("conv_wgrad_kernel",
 * "conv_wgrad_planes_kernel", "thin_wgrad_kernel", "dense_small_wgrad_kernel").  For profiling tools. */
int igan_conv2d_wgrad_kernel_name(const igan_conv2d_wgrad_params* p, char* buf, int buflen);

/* ABI v6: the library's own answer to "will a layer with this filter take a piece form?" (the batch-independent part of the
 * rule: 3x3 taps, both channel counts >= 128 and whole 32s, a piece form switched on) and "is a tensor [N, HW, C] one igan_to_pieces() can
 * image for it?" (form 1 only: always 0 in forms 0 and 2) -- so that a host does not restate the rules.  Both return 0 / 1. */
int igan_conv_pieces_wanted(int KH, int KW, int Cin, int Cout);
int igan_pieces_image_ok(int N, int HW, int C);
/* ABI v7: which piece form this process runs (read once from IGAN_CONV_PLANES): 0 = none (every convolution on the fp32 matrix instruction),
 * 1 = three bf16 pieces / six products, 2 = two fp16 pieces (the default when the variable is unset): p0 = fp16(v S), p1 = fp16((v S - p0) 2^11)
 * with a power-of-two S per scale group (ABI v8: a pixel's channel vector / a filter's output channel in the forward and data-gradient kernel, a
 * channel's pixels in the weight gradient -- never a whole tensor), three products; the operand to 2^-23 (exactly in three cases of four) for every
 * element within 2^26 of the largest of its own group. */
int igan_conv_piece_form(void);
/* ABI v9: the filter image of a convolution call as a caller-kept buffer.  igan_filter_image_bytes: its size for a call with this filter geometry (Cin / Cout are the
 * CALL's: a data-gradient call has them swapped against its forward layer), 0 = this process / filter takes none (form 0; a 1x1 or thin filter; a channel count outside
 * the piece form).  igan_filter_image: writes it (16-byte aligned `out`; the same kernels igan_conv2d runs for its own image, so results are bit-identical). */
size_t igan_filter_image_bytes(int KH, int KW, int Cin, int Cout);
int igan_filter_image(igan_stream_t stream, const float* w, void* out, int KH, int KW, int Cin, int Cout, int w_transposed);
/* ABI v8: size in floats of an x_colmax / dy_colmax buffer for a tensor [N, HW, C]; 0 = none is taken (not the fp16 form, a channel count outside it, or fewer
 * pixels than any weight gradient that takes the column image has: N * HW below the piece form's row threshold). */
size_t igan_colmax_floats(int N, int HW, int C);
/* Diagnostic for form 2 (synchronises the device): non-zero elements imaged so far BELOW the exact window (|v S| < 2^-12: more than 2^26 below the
 * largest magnitude of the element's own scale group) and elements imaged in all; reset != 0 zeroes both counters. */
int igan_debug_f16_window(unsigned long long* below, unsigned long long* imaged, int reset);
/* ABI v8: the same counters by kind of image, out4 = {rows below, rows imaged, columns below, columns imaged}: a ROW image (forward / data gradient) scales a
 * pixel's channel vector, a COLUMN image (weight gradient) a channel's pixels -- there the group runs along the summed axis, where an element 2^26 below the
 * group's largest stands next to that largest element in every sum it enters. */
int igan_debug_f16_window_by_kind(unsigned long long* out4, int reset);

/* bf16-piece form (IGAN_CONV_PLANES=1) only -- IGAN_ERR_UNSUPPORTED in the other forms: the piece image of a channel-minor tensor
 * x [N, HW, C] (times scale [N, C] when given), `out` = N * HW * C * 6 bytes, 16-byte aligned, C % 16 == 0.  The convolution entry
 * points write the images they need themselves; a caller that feeds one tensor to several of them (dy to the data and the
 * weight gradient, x to the forward pass and the weight gradient) writes it once with this and passes it as x_pieces / dy_pieces. */
int igan_to_pieces(igan_stream_t stream, const float* x, const float* scale, void* out, int N, int HW, int C);

/* ------------------------------------------------------------------------
 * Small-batch dense layers with the StyleGAN2 style-path arithmetic folded in (M <= 64 rows; larger
 * batches go through igan_conv2d as 1x1 convolutions):
 *     y[m,n] = epi( alpha * sum_k pro(x)[m,k] * W(k,n) ),   W = w[k][n], or w[n][k] when w_transposed
 * prologue (on x, element-wise):   NONE; SQUARE x^2; DEMOD_GRAD pro_scale * x * x2^3
 * epilogue:  SCALE  v
 *            BIAS   v + bias_scale * bias[n] + add_const                 (style affine + 1, networks_stylegan2.py:99-101)
 *            RSQRT  rsqrt(v + eps)                                         (demodulation coefficients, :105-107)
 *            STYLE_GRAD  e1[m,n] + 2 * e2[m,n] * v, and, if colsum != NULL, colsum[n] = bias_scale * sum_m y[m,n]
 * so that, with q = s^2 . wsq and d = rsqrt(c^2 q + eps):
 *     s  = BIAS(x=w_lat, W=A)                       d  = RSQRT(SQUARE s, W=wsq, alpha=c^2)
 *     ds = STYLE_GRAD(DEMOD_GRAD(dd, d; pro_scale=-c^2/2), W=wsq^T, e1=ds_conv, e2=s)   [+ bias gradient]
 * Weight-gradient form:  dw[k,n] = alpha * sum_m pro_a(a)[m,k] * pro_b(b)[m,n]   (pro_a NONE|SQUARE,
 * pro_b NONE|DEMOD_GRAD with b2).  K % 4 == 0 (dense), N % 4 == 0 (weight gradient); x / x2 / b / b2 / dw
 * 16-byte aligned.  Deterministic. */
enum { IGAN_DENSE_PRO_NONE = 0, IGAN_DENSE_PRO_SQUARE = 1, IGAN_DENSE_PRO_DEMOD_GRAD = 2 };
enum { IGAN_DENSE_EPI_SCALE = 0, IGAN_DENSE_EPI_BIAS = 1, IGAN_DENSE_EPI_RSQRT = 2, IGAN_DENSE_EPI_STYLE_GRAD = 3 };
typedef struct igan_dense_params {
    const float* x;         /* [M, K], row stride ldx floats (ldx % 4 == 0) */
    const float* x2;        /* DEMOD_GRAD: [M, K] contiguous, else NULL */
    const float* w;         /* [K, N], or [N, K] when w_transposed */
    float* y;               /* [M, N], row stride ldy floats */
    const float* bias;      /* BIAS: [N] */
    const float* e1;        /* STYLE_GRAD: [M, N] contiguous or NULL (= 0) */
    const float* e2;        /* STYLE_GRAD: [M, N] contiguous */
    float* colsum;          /* STYLE_GRAD: [N] or NULL */
    int ldx, ldy;
    int M, K, N;
    int w_transposed;
    int prologue, epilogue;
    float alpha, pro_scale, bias_scale, add_const, eps;
} igan_dense_params;
int igan_dense_small(igan_stream_t stream, const igan_dense_params* p);

typedef struct igan_dense_wgrad_params {
    const float* a;         /* [M, K], row stride lda floats */
    const float* b;         /* [M, N] contiguous */
    const float* b2;        /* DEMOD_GRAD: [M, N] contiguous, else NULL */
    float* dw;              /* [K, N] */
    int lda;
    int M, K, N;
    int pro_a, pro_b;
    float alpha, pro_scale;
} igan_dense_wgrad_params;
int igan_dense_small_wgrad(igan_stream_t stream, const igan_dense_wgrad_params* p);

/* Grouped forms: `count` (<= IGAN_DENSE_MAX_GROUPS) independent problems in ONE launch -- the 18 style affines, the 12
 * demodulations, and each stage of their backward, of one generator pass.  The groups of a dense launch share
 * `w_transposed`; sizes may differ per group. */
#define IGAN_DENSE_MAX_GROUPS 24
#define IGAN_DENSE_MAX_ROWS 64
int igan_dense_small_grouped(igan_stream_t stream, const igan_dense_params* groups, int count);
int igan_dense_small_wgrad_grouped(igan_stream_t stream, const igan_dense_wgrad_params* groups, int count);

/* out[i] = sum_t w[t*n + i]^2  (sum over the filter taps of the squared weights: the [Cin,Cout] matrix of the
 * demodulation, :105) and out[t*n + i] = scale * w[t*n + i] * v[i] (its gradient back onto the filter). */
int igan_sumsq_taps(igan_stream_t stream, const float* w, float* out, int taps, int n);
int igan_bcast_mul_taps(igan_stream_t stream, const float* w, const float* v, float* out, int taps, int n, float scale);
typedef struct igan_taps_params {
    const float* w;         /* [taps, n] */
    const float* v;         /* bcast_mul: [n]; sumsq: unused (NULL) */
    float* out;             /* sumsq: [n]; bcast_mul: [taps, n] */
    int taps, n;
    float scale;            /* bcast_mul only */
} igan_taps_params;
int igan_sumsq_taps_grouped(igan_stream_t stream, const igan_taps_params* groups, int count);
int igan_bcast_mul_taps_grouped(igan_stream_t stream, const igan_taps_params* groups, int count);

/* Per-sample channel dot products of two channel-minor tensors a, b [N, HW, C] (C % 4 == 0):
 *     dot[n,c] = sum_hw a[n,hw,c] * b[n,hw,c];   if out != NULL: out[n,hw,c] = b[n,hw,c] * s[n,c]
 * (out may alias b; s may be NULL = 1).  These are the style / demodulation gradients of
 * modulated_conv2d_layer in its non-fused form (networks_stylegan2.py:112,126): ds = dot(x, g) with
 * dx = g * s, and dd = dot(dy, y) / d.  Deterministic; workspace of
 * igan_scale_dot_workspace_floats(N, HW, C) floats. */
size_t igan_scale_dot_workspace_floats(int N, int HW, int C);
int igan_scale_dot(igan_stream_t stream, const float* a, const float* b, const float* s, float* out,
                   float* dot, float* workspace, int N, int HW, int C);

/* LPIPS per-layer distance (Zhang et al. 2018; role of lpips.get_output_for, training/loss.py:31,41) on raw
 * channel-minor VGG features fa, fb [N, HW, C], C in {64,128,256,512}, lin[C] >= 0:
 *     partial[n][j] = sum over block j's pixels of sum_c lin_c (fa_c/(|fa|+1e-10) - fb_c/(|fb|+1e-10))^2
 * (igan_lpips_layer_blocks(N, HW) partial sums per sample; caller adds them and divides by HW), and
 *     dfa = g[n] * d(distance sum)/d fa        (swap fa, fb for the other argument).  One pass each. */
int igan_lpips_layer_blocks(int N, int HW);
int igan_lpips_layer_fwd(igan_stream_t stream, const float* fa, const float* fb, const float* lin,
                         float* partial, int N, int HW, int C);
int igan_lpips_layer_bwd(igan_stream_t stream, const float* fa, const float* fb, const float* lin,
                         const float* g, float* dfa, int N, int HW, int C);
/* The same over a table of P sample pairs: row p compares sample ia[p] of fa with sample ib[p] of fb (NULL table =
 * identity) -- all four distances of the G loss (rec_1/real_1, rec_2/real_2, interp/real_2, interp/real_1, loss.py:31,41) in
 * one launch per layer.  Forward writes partial[p * partial_stride + j], j < blocks (caller picks blocks <= HW and lays the
 * layers' columns side by side so that one row sum finishes the distance).  Backward writes (accumulate = 0) or adds
 * (accumulate = 1) g[p] * d(distance sum)/d fa into sample ia[p] of dfa; two rows of one launch must not name the same
 * ia[p]. */
int igan_lpips_pairs_fwd(igan_stream_t stream, const float* fa, const float* fb, const float* lin, const int* ia,
                         const int* ib, float* partial, int partial_stride, int blocks, int P, int HW, int C);
int igan_lpips_pairs_bwd(igan_stream_t stream, const float* fa, const float* fb, const float* lin, const int* ia,
                         const int* ib, const float* g, float* dfa, int accumulate, int P, int HW, int C);

/* 2x2 max-pool between the VGG blocks of the LPIPS network (inside lpips.get_output_for, training/loss.py:31,41;
 * the network pickle is absent from the reference tree: restated) on channel-minor x [N, H, W, C] -> y [N, H/2, W/2, C],
 * and its gradient fused with the sum over the pooled map's two consumers (the map is also an LPIPS tap):
 *     dx = dskip + route(dy to the first maximum of each window in order (0,0) (0,1) (1,0) (1,1); NaN wins)
 * dskip may be NULL (= 0); dx may alias dskip.  H, W even, C % 4 == 0. */
int igan_maxpool2x2_fwd(igan_stream_t stream, const float* x, float* y, int N, int H, int W, int C);
int igan_maxpool2x2_bwd(igan_stream_t stream, const float* x, const float* dy, const float* dskip, float* dx,
                        int N, int H, int W, int C);

/* ------------------------------------------------------------------------
 * minibatch_stddev_layer statistics (networks_stylegan2.py:132-144), NHWC input
 * x[N, H, W, C], group size G (N % G == 0, M = N / G, num_new_features = 1):
 *   stat[m] = mean_{c,h,w} sqrt( mean_g (x[g*M+m] - mean_g x)^2 + 1e-8 )
 * fwd writes y[N, H, W, C+1] = concat(x, stat[n % M]); the statistic is reduced over position slices through
 * `workspace` (igan_mbstd_workspace_floats(N,H,W,C,G) floats, fixed order).
 * bwd takes dy[N,H,W,C+1] and returns dx[N,H,W,C] (pass-through + statistic path).
 */
size_t igan_mbstd_workspace_floats(int N, int H, int W, int C, int G);
int igan_mbstd_fwd(igan_stream_t stream, const float* x, float* y, float* workspace,
                   int N, int H, int W, int C, int G);
int igan_mbstd_bwd(igan_stream_t stream, const float* x, const float* dy, float* dx,
                   int N, int H, int W, int C, int G);

/* ------------------------------------------------------------------------
 * Exact streaming 1-nearest-neighbour (replaces dci_add + dci_query as used at
 * training/training_loop.py:367-368,398 with num_neighbours = 1).
 *   (best_d2[q], best_idx[q]) = lexicographic min over candidates c of (|query_q - cand_c|^2, idx_base + c)
 * with the squared distance of the winner -- and of every candidate that could be the winner -- computed
 * as a direct difference in fp64 (compute_dist, dci_code/src/util.c:62-69, before its sqrt), i.e. the
 * result equals an fp64 brute-force search (ties go to the lower index).  The fp32 MFMA product
 * |q|^2 + |c|^2 - 2 q.c only screens: candidates whose error interval (relative half-width
 * 2^-22 * sqrt(dim) of |q|^2 + |c|^2) cannot reach the best known upper bound are dropped, the rest are
 * measured exactly.  Initialise best_d2 to +infinity and best_idx to INT32_MAX; candidates can be streamed
 * batch after batch with increasing idx_base, the running minimum is order-independent, so the result is
 * deterministic.  A candidate with a non-finite product or distance never wins.
 * qnorm / cnorm are the squared row norms (igan_row_sqnorm, fp64 accumulate).  The caller takes sqrt
 * (Euclidean distance) of best_d2.
 */
int igan_row_sqnorm(igan_stream_t stream, const float* a, float* out, int rows, int dim);
int igan_nn1_update(igan_stream_t stream, const float* query, const float* qnorm,
                    const float* cand, const float* cnorm, double* best_d2, int* best_idx,
                    float* dots /* caller workspace, nq*nc floats */,
                    int nq, int nc, int dim, int idx_base);

/* ------------------------------------------------------------------------
 * Device-side time stamps (measurement support, not part of the reference's surface): igan_stamp writes the constant
 * 100 MHz counter (10 ns ticks) into *slot in stream order, so two stamps bracket whatever was launched between them --
 * also inside a captured hipGraph, where host events cannot be placed.  igan_stamp_accumulate adds, for pairs
 * first .. first+count-1, stamps[2i+1] - stamps[2i] into acc[i]; captured at the end of a graph it turns every replay
 * into one more sample per launch without host involvement. */
int igan_stamp(igan_stream_t stream, unsigned long long* slot);
/* Diagnostic: while p != NULL every igan_conv2d launch of the MFMA forward kernel writes, per workgroup, 4 ticks of the same
 * counter (kernel entry, main-loop start, main-loop end, exit) to p[4 * workgroup + k]; the buffer must hold 4 * grid words. */
void igan_debug_set_conv_diag(unsigned long long* p);
int igan_stamp_accumulate(igan_stream_t stream, const unsigned long long* stamps, unsigned long long* acc, int first, int count);

/* ------------------------------------------------------------------------
 * Flat-bucket optimizer step (dnnlib/tflib/optimizer.py:237-239,318-332):
 *   flag[0] = all(isfinite(grad));  if flag: Adam update; else: skip.
 *   m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; w -= lr_t * m / (sqrt(v) + eps)
 * lr_t = lr * sqrt(1 - b2pow') / (1 - b1pow') with b?pow' = b?pow * beta? read from
 * the device-resident pow_state[2] = {b1pow, b2pow} (initialise to {1, 1}); the
 * powers advance only when the step is applied, like TF's beta-power slots, and
 * no host synchronisation is needed to learn whether it was.
 * igan_finite_check ORs a non-zero into flag[0] when any element is non-finite
 * (flag must be zeroed by the caller before the first check of a step).
 * igan_ema: dst = src + (dst - src) * beta  (Network.setup_as_moving_average_of,
 * dnnlib/tflib/network.py:341-351).
 */
int igan_finite_check(igan_stream_t stream, const float* g, int n, int* flag);
int igan_adam_step(igan_stream_t stream, float* w, const float* g, float* m, float* v,
                   int n, float lr, float beta1, float beta2, float eps,
                   float* pow_state, const int* skip_flag);
int igan_ema(igan_stream_t stream, float* dst, const float* src, int n, float beta);

/* Running mean of a training scalar (role of dnnlib/tflib/autosummary.py:45-74): acc[0] += number of finite values among
 * x[0..n), acc[1] += their sum (double accumulators that live across hipGraph replays; non-finite values are ignored, :64).
 * One launch, fixed-order reduction. */
int igan_summary_accumulate(igan_stream_t stream, const float* x, int n, double* acc);

#ifdef __cplusplus
}
#endif
#endif /* IGAN_HIP_H */
