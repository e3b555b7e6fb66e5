"""CPU: the arithmetic of the two-piece fp16 form of the large 3x3 convolutions (csrc/conv2d_mfma.hip "TWO-PIECE fp16 form"),
restated in NumPy bit for bit -- the scale chosen from a scale group's largest magnitude (round 5: a pixel's channel vector, a filter's output channel,
a channel's pixels -- never a whole tensor; scale_from_amax / inv_scale_from_amax: exponent arithmetic on
the float's bits), the split (split2: p0 = fp16(v S), p1 = fp16((v S - p0) 2^11), both round to nearest even) and the product rule
a b = (Sa Sb)^-1 [p0a p0b + 2^-11 (p0a p1b + p1a p0b)] -- and the claims the header makes about them: the scale is an exact power of two that puts the
largest magnitude in [2^14, 2^15); the pieces reconstruct v S to ONE unit in the last place of the fp32 value (2^-23 |v S|; exactly in about three
cases of four: 11 + 11 significand bits + the sign of the second piece = 23 bits always, the 24th when the residual fits) inside the window
|v S| >= 2^-12 and to 2^-36 absolute below it; the dropped term is at most 2^-22 |a b|; a dot product formed this way is closer to fp64 than a
sequential fp32 FMA chain.  (The GPU side of the same
statements: tests/test_gpu_planes_variant.py.)"""
import numpy as np


def scale_bits(amax):
    """(S, 1 / S) as the kernels form them: e = biased exponent of amax clamped to [15, 254]; S = 2^(141 - e), 1 / S = 2^(e - 141)."""
    e = int((np.float32(amax).view(np.uint32) >> 23) & 0xFF)
    e = max(15, min(e, 254))
    S = np.uint32((268 - e) << 23).view(np.float32)
    inv = np.uint32((e - 14) << 23).view(np.float32)
    return S, inv


def split2(vs):
    vs = np.asarray(vs, np.float32)
    p0 = vs.astype(np.float16)
    p1 = ((vs - p0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return p0, p1


def test_scale_is_an_exact_power_of_two_that_normalises_the_largest_magnitude():
    rng = np.random.default_rng(0)
    for amax in list(np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 200)).astype(np.float32)) + [np.float32(1.0), np.float32(2.0 ** 14), np.float32(65504.0), np.float32(3e38)]:
        S, inv = scale_bits(amax)
        assert float(S) * float(inv) == 1.0
        m, _ = np.frexp(float(S))
        assert m == 0.5                                                    # a power of two: multiplying by it is exact
        assert 2.0 ** 14 <= float(amax) * float(S) < 2.0 ** 15, (amax, S)
    for amax in (0.0, 1e-45, 2.0 ** -120):                                 # nothing / subnormal / below 2^-112: the largest shift that keeps 1 / S normal
        S, inv = scale_bits(np.float32(amax))
        assert float(S) == 2.0 ** 126 and float(inv) == 2.0 ** -126


def test_two_pieces_hold_the_value_to_one_ulp_inside_the_window_and_2m36_absolute_below_it():
    rng = np.random.default_rng(1)
    mag = np.exp2(rng.uniform(-30, 15, 200000))
    vs = (mag * rng.choice([-1.0, 1.0], mag.shape) * rng.uniform(0.5, 1.0, mag.shape)).astype(np.float32)
    vs = vs[np.abs(vs) < 32768.0]
    p0, p1 = split2(vs)
    assert np.all(np.isfinite(p0.astype(np.float32))) and np.all(np.isfinite(p1.astype(np.float32)))
    normal = np.abs(vs) >= 2.0 ** -14
    assert np.all(np.abs(p1[normal].astype(np.float64)) <= np.abs(p0[normal].astype(np.float64)))   # the second piece is stored 2^11 up and still not larger than the first
    rec = p0.astype(np.float64) + p1.astype(np.float64) / 2048.0
    err = np.abs(rec - vs.astype(np.float64))
    inside = np.abs(vs) >= 2.0 ** -12
    rel = err[inside] / np.abs(vs[inside].astype(np.float64))
    assert np.all(rel <= 2.0 ** -23)                                       # never more than one unit in the last place of the fp32 value
    assert 0.70 < float((rel == 0).mean()) < 0.80                          # and exact in about three cases of four
    assert float(np.sqrt((rel ** 2).mean())) < 5e-8                        # rms: the size of one fp32 rounding (2^-24 / sqrt(3) = 3.4e-8)
    assert np.all(err[~inside] <= 2.0 ** -36)                              # below the window: fp16's subnormal spacing of the second piece (2^-24 / 2^11), one bit less per binade


def _conv1d_pairs(x, w):
    """The forward kernel's arithmetic (conv_fwd_planes_kernel<2>) on a 1-D three-tap convolution  y[m, n] = sum_t sum_c x[m + t - 1, c] w[t, c, n]:
    one scale per PIXEL of x (its channel vector: rows_f16_kernel) and per OUTPUT CHANNEL of w (filter_planes_f16_kernel); the reduction runs 16-channel
    slice outermost, taps inside; per step the main term and the two cross terms start from zero in the matrix pipe (modelled: exact sums rounded to
    fp32), the vector ALU forms t' = fp32(t + v / 2048) and acc = fma(t', 1 / S_pixel, acc); the epilogue multiplies by 1 / S_n."""
    P, C = x.shape
    T, _, N = w.shape
    Sp = np.array([scale_bits(np.abs(x[p]).max()) for p in range(P)], np.float32)           # [P, 2]: S, 1 / S
    Sn = np.array([scale_bits(np.abs(w[:, :, n]).max()) for n in range(N)], np.float32)
    a0, a1 = split2(x * Sp[:, :1]); b0, b1 = split2(w * Sn[None, None, :, 0])
    a0, a1, b0, b1 = (t.astype(np.float64) for t in (a0, a1, b0, b1))
    acc = np.zeros((P, N), np.float32)
    for c0 in range(0, C, 16):
        cs = slice(c0, c0 + 16)
        for t in range(T):
            sh = t - T // 2
            lo, hi = max(0, -sh), min(P, P - sh)                                              # output rows whose tap-t pixel exists (others read zeros)
            src = slice(lo + sh, hi + sh)
            main = (a0[src, cs] @ b0[t, cs]).astype(np.float32)
            cross = (a0[src, cs] @ b1[t, cs] + a1[src, cs] @ b0[t, cs]).astype(np.float32)
            step = (main.astype(np.float64) + cross.astype(np.float64) / 2048.0).astype(np.float32)
            acc[lo:hi] = (acc[lo:hi].astype(np.float64) + step.astype(np.float64) * Sp[src, 1:].astype(np.float64)).astype(np.float32)
    return (acc.astype(np.float64) * Sn[None, :, 1].astype(np.float64)).astype(np.float32)


def _conv1d_fp32_chain(x, w):
    """The exact-fp32 matrix instruction: one fp32 FMA per term, in the kernel's order (tap outermost, channels inside)."""
    P, C = x.shape
    T, _, N = w.shape
    acc = np.zeros((P, N), np.float32)
    for t in range(T):
        sh = t - T // 2
        lo, hi = max(0, -sh), min(P, P - sh)
        for c in range(C):
            acc[lo:hi] = (acc[lo:hi].astype(np.float64) + np.outer(x[lo + sh:hi + sh, c].astype(np.float64), w[t, c].astype(np.float64))).astype(np.float32)
    return acc


def _conv1d_exact(x, w, absolute=False):
    x = x.astype(np.float64); w = w.astype(np.float64)
    if absolute:
        x, w = np.abs(x), np.abs(w)
    P = x.shape[0]
    y = np.zeros((P, w.shape[2]))
    for t in range(w.shape[0]):
        sh = t - w.shape[0] // 2
        lo, hi = max(0, -sh), min(P, P - sh)
        y[lo:hi] += x[lo + sh:hi + sh] @ w[t]
    return y


def test_forward_arithmetic_has_no_window_across_pixels_or_output_channels():
    """Round 5: scales per pixel and per output channel.  Measured per element against its own sum of magnitudes, the fp16 form is no worse than the
    fp32 FMA chain on tensors with an outlier of 2^30, a dead pixel run at 2^-30, a dead output channel, a dead input channel -- the inputs on which
    ONE scale per tensor (round 4) lost up to a bit per binade beyond 2^26 (the GPU side of the same statement: tests/test_gpu_f16_dynamic_range.py)."""
    rng = np.random.default_rng(5)
    P, C, N = 96, 128, 40
    def base():
        return rng.standard_normal((P, C)).astype(np.float32), (rng.standard_normal((3, C, N)) / np.sqrt(3 * C)).astype(np.float32)
    cases = {}
    x, w = base(); cases['randn'] = (x, w)
    x, w = base(); x[17, 5] *= np.float32(2.0 ** 30); w[1, 9, 3] *= np.float32(2.0 ** 30); cases['outlier 2^30'] = (x, w)
    x, w = base(); x[40:60] *= np.float32(2.0 ** -30); cases['pixels at 2^-30'] = (x, w)
    x, w = base(); w[:, :, 7] *= np.float32(2.0 ** -30); w[:, 11, :] *= np.float32(2.0 ** -30); x[:, 3] *= np.float32(2.0 ** -30); cases['channels at 2^-30'] = (x, w)
    x, w = base(); w *= np.float32(2.0 ** -100); cases['filters at 2^-100'] = (x, w)
    for name, (x, w) in cases.items():
        exact, mag = _conv1d_exact(x, w), _conv1d_exact(x, w, absolute=True)
        e16 = np.abs(_conv1d_pairs(x, w) - exact) / mag
        e32 = np.abs(_conv1d_fp32_chain(x, w) - exact) / mag
        assert e16.max() <= 2.0 * e32.max() and np.sqrt((e16 ** 2).mean()) <= np.sqrt((e32 ** 2).mean()), (name, e16.max(), e32.max())
        assert e16.max() < 2.0 ** -21, (name, e16.max())


def test_one_scale_per_tensor_would_fail_the_same_inputs():
    """The sensitivity of the check above: with ONE power-of-two scale for the whole of x (round 4's form) the run of pixels at 2^-30 of the rest loses
    four of its 24 bits (2^30 / 2^26): measured per element against its own sum of magnitudes, ten times the per-pixel form's error on those rows."""
    rng = np.random.default_rng(6)
    P, C, N = 64, 128, 24
    x = rng.standard_normal((P, C)).astype(np.float32); w = (rng.standard_normal((3, C, N)) / np.sqrt(3 * C)).astype(np.float32)
    x[20:40] *= np.float32(2.0 ** -30)
    S, inv = scale_bits(np.abs(x).max())
    p0, p1 = split2(x * S)
    xq = ((p0.astype(np.float64) + p1.astype(np.float64) / 2048.0) * float(inv))            # what the per-tensor image holds
    rows = slice(22, 38)                                                                      # outputs that read dark pixels only
    exact, mag = _conv1d_exact(x, w)[rows], _conv1d_exact(x, w, absolute=True)[rows]
    e_tensor = (np.abs(_conv1d_exact(xq, w)[rows] - exact) / mag).max()
    e_pixel = (np.abs(_conv1d_pairs(x, w)[rows] - exact) / mag).max()
    assert e_tensor > 2.0 ** -22 > 4.0 * e_pixel, (e_tensor, e_pixel)         # measured: 2^-21.3 against 2^-24.6


def test_weight_gradient_arithmetic_scales_per_channel():
    """conv_wgrad_planes_kernel<2>: both operands are summed over pixels, so each carries one scale per CHANNEL (cols_f16_kernel); the main term is folded
    per 16-pixel step, the cross terms are chained over the whole reduction, the epilogue multiplies row ci by 1 / S_ci and column co by 1 / S_co."""
    rng = np.random.default_rng(7)
    P, Ci, Co = 512, 48, 40
    x = rng.standard_normal((P, Ci)).astype(np.float32); dy = rng.standard_normal((P, Co)).astype(np.float32)
    x[:, 5] *= np.float32(2.0 ** -30); dy[:, 9] *= np.float32(2.0 ** -30); x[100, 7] *= np.float32(2.0 ** 30); dy[300, 2] *= np.float32(2.0 ** 30)
    Sx = np.array([scale_bits(np.abs(x[:, c]).max()) for c in range(Ci)], np.float32); Sd = np.array([scale_bits(np.abs(dy[:, c]).max()) for c in range(Co)], np.float32)
    a0, a1 = split2(x * Sx[None, :, 0]); b0, b1 = split2(dy * Sd[None, :, 0])
    a0, a1, b0, b1 = (t.astype(np.float64) for t in (a0, a1, b0, b1))
    acc = np.zeros((Ci, Co), np.float32); u = np.zeros((Ci, Co), np.float32)
    for p0_ in range(0, P, 16):
        ps = slice(p0_, p0_ + 16)
        acc = (acc.astype(np.float64) + (a0[ps].T @ b0[ps]).astype(np.float32)).astype(np.float32)
        u = (u.astype(np.float64) + a0[ps].T @ b1[ps] + a1[ps].T @ b0[ps]).astype(np.float32)
    got = (((acc.astype(np.float64) + u.astype(np.float64) / 2048.0) * Sx[:, 1:].astype(np.float64)) * Sd[None, :, 1].astype(np.float64)).astype(np.float32)
    exact = x.astype(np.float64).T @ dy.astype(np.float64); mag = np.abs(x.astype(np.float64)).T @ np.abs(dy.astype(np.float64))
    chain = np.zeros((Ci, Co), np.float32)
    for p in range(P):
        chain = (chain.astype(np.float64) + np.outer(x[p].astype(np.float64), dy[p].astype(np.float64))).astype(np.float32)
    e16, e32 = np.abs(got - exact) / mag, np.abs(chain - exact) / mag
    assert e16.max() <= 2.0 * e32.max() and e16.max() < 2.0 ** -21, (e16.max(), e32.max())


def test_product_rule_and_dot_product_accuracy():
    rng = np.random.default_rng(2)
    M, N, K = 64, 48, 1152
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) * 0.03).astype(np.float32)
    Sa, ia = scale_bits(np.abs(x).max()); Sb, ib = scale_bits(np.abs(w).max())
    a0, a1 = split2(x * Sa); b0, b1 = split2(w * Sb)
    a0, a1, b0, b1 = (t.astype(np.float64) for t in (a0, a1, b0, b1))
    # the dropped term of one product: |p1| <= |p0| on both sides, so 2^-22 |p1a p1b| <= 2^-22 |p0a p0b|
    dropped = np.abs(a1[:, :1] * b1[:1, :]) / 2048.0 ** 2
    assert np.all(dropped <= 2.0 ** -22 * np.abs(a0[:, :1] * b0[:1, :]))
    # the kernel's order: per 16-deep step the main term from zero (fp32 result), folded by an fp32 add; the cross terms chained in fp32
    acc = np.zeros((M, N), np.float32); u = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        acc = (acc.astype(np.float64) + (a0[:, s] @ b0[s]).astype(np.float32)).astype(np.float32)
        u = (u.astype(np.float64) + a0[:, s] @ b1[s] + a1[:, s] @ b0[s]).astype(np.float32)
    got = ((acc.astype(np.float64) + u.astype(np.float64) / 2048.0) * float(ia) * float(ib)).astype(np.float32)
    exact = x.astype(np.float64) @ w.astype(np.float64)
    chain = np.zeros((M, N), np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + np.outer(x[:, k].astype(np.float64), w[k].astype(np.float64))).astype(np.float32)
    e_pairs = np.linalg.norm(got - exact) / np.linalg.norm(exact)
    e_chain = np.linalg.norm(chain - exact) / np.linalg.norm(exact)
    assert e_pairs < 3e-7 and e_pairs < e_chain, (e_pairs, e_chain)
