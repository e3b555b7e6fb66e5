"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

fp64 values of SINGLE output elements of the convolution family by direct gather, for spot checks at sizes where the
whole-tensor fp64 oracle would take minutes.  The definitions are those the reference gets from TensorFlow
(NHWC cross-correlation, `tf.nn.conv2d` networks_stylegan2.py:60,120 and upfirdn_2d.py:332; `tf.nn.conv2d_transpose`
with the pre-flipped filter of upfirdn_2d.py:286-291 = zero-insertion + full correlation; `tf.matmul` :46) and from
`tf.gradients` of them, with the modulation / demodulation scalings of the non-fused modulated_conv2d_layer
(networks_stylegan2.py:112,126) and the runtime weight scale (:30-36) as `alpha`:

    y[n,co,oy,ox]  = alpha * d[n,co] * sum_{ky,kx,ci} xup[n,ci, oy*stride+ky-pad, ox*stride+kx-pad] * s[n,ci] * w[ky,kx,ci,co]
    xup[n,ci,v,u]  = x[n,ci,v/up,u/up] if v % up == 0 and u % up == 0 and inside, else 0
    dx[n,ci,iy,ix] = d y / d x contracted with dy      dw[ky,kx,ci,co] = d y / d w contracted with dy

Arrays are logical NCHW NumPy (any float dtype, promoted to float64); w is HWIO.  Nothing here shares code or loop
structure with the HIP kernels (no tiles, no parity classes, no im2col).
"""
import numpy as np


def _f64(a):
    return None if a is None else np.asarray(a, dtype=np.float64)


def forward_samples(x, w, idx, stride, up, pad, s=None, d=None, alpha=1.0):
    """idx: int array [m, 4] of (n, co, oy, ox) -> float64 [m]."""
    x, w, s, d = _f64(x), _f64(w), _f64(s), _f64(d)
    N, Cin, H, W = x.shape
    KH, KW = w.shape[:2]
    out = np.empty(len(idx), dtype=np.float64)
    for i, (n, co, oy, ox) in enumerate(idx):
        acc = 0.0
        for ky in range(KH):
            v = oy * stride + ky - pad
            if v < 0 or v % up or v // up >= H:
                continue
            for kx in range(KW):
                u = ox * stride + kx - pad
                if u < 0 or u % up or u // up >= W:
                    continue
                col = x[n, :, v // up, u // up]
                if s is not None:
                    col = col * s[n]
                acc += float(col @ w[ky, kx, :, co])
        out[i] = acc * alpha * (d[n, co] if d is not None else 1.0)
    return out


def dgrad_samples(dy, w, idx, in_hw, stride, up, pad, s=None, d=None, alpha=1.0):
    """idx: [m, 4] of (n, ci, iy, ix) -> d<y, dy>/dx at those input elements, float64 [m]."""
    dy, w, s, d = _f64(dy), _f64(w), _f64(s), _f64(d)
    N, Cout, OH, OW = dy.shape
    KH, KW = w.shape[:2]
    out = np.empty(len(idx), dtype=np.float64)
    for i, (n, ci, iy, ix) in enumerate(idx):
        acc = 0.0
        for ky in range(KH):
            t = iy * up - ky + pad          # = oy * stride
            if t < 0 or t % stride or t // stride >= OH:
                continue
            for kx in range(KW):
                r = ix * up - kx + pad
                if r < 0 or r % stride or r // stride >= OW:
                    continue
                g = dy[n, :, t // stride, r // stride]
                if d is not None:
                    g = g * d[n]
                acc += float(g @ w[ky, kx, ci, :])
        out[i] = acc * alpha * (s[n, ci] if s is not None else 1.0)
    return out


def wgrad_samples(x, dy, idx, stride, up, pad, s=None, d=None, alpha=1.0):
    """idx: [m, 4] of (ky, kx, ci, co) -> d<y, dy>/dw at those filter elements, float64 [m]."""
    x, dy, s, d = _f64(x), _f64(dy), _f64(s), _f64(d)
    N, Cin, H, W = x.shape
    _, Cout, OH, OW = dy.shape
    out = np.empty(len(idx), dtype=np.float64)
    for i, (ky, kx, ci, co) in enumerate(idx):
        oy = np.arange(OH); ox = np.arange(OW)
        v = oy * stride + ky - pad; u = ox * stride + kx - pad
        oky = (v >= 0) & (v % up == 0) & (v // up < H)
        okx = (u >= 0) & (u % up == 0) & (u // up < W)
        if not oky.any() or not okx.any():
            out[i] = 0.0
            continue
        xs = x[:, ci][:, (v[oky] // up)][:, :, (u[okx] // up)]            # [N, oy', ox']
        g = dy[:, co][:, oky][:, :, okx]
        per_n = np.einsum('nyx,nyx->n', xs, g)
        if s is not None:
            per_n = per_n * s[:, ci]
        if d is not None:
            per_n = per_n * d[:, co]
        out[i] = float(per_n.sum()) * alpha
    return out
