"""CPU: the arithmetic of the two-piece fp16 form of the large 3x3 convolutions (csrc/conv2d_mfma.hip "TWO-PIECE fp16 form"),
restated in NumPy bit for bit -- the scale chosen from a tensor's largest magnitude (scale_from_amax / inv_scale_from_amax: exponent arithmetic on
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
