"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

CPU restatement (NumPy / PyTorch-CPU, any float dtype) of the reference's resampling ops:

  upfirdn_2d_loops   index arithmetic of the general CUDA kernel, dnnlib/tflib/ops/upfirdn_2d.cu:64-117
                     (pure-Python loops: small cases only)
  upfirdn_2d_ref     the TF-ops formulation `_upfirdn_2d_ref`, dnnlib/tflib/ops/upfirdn_2d.py:66-101
                     (zero-insert, pad/crop, VALID conv with the flipped filter, decimate)
  setup_kernel / simple_upfirdn_2d / filter_2d / upsample_2d / downsample_2d /
  upsample_conv_2d / conv_downsample_2d      upfirdn_2d.py:144-362, literally, including the
                     conv2d_transpose + filter flip/regroup of :286-291
  upfirdn_2d_grad_params                     the custom-gradient parameters, upfirdn_2d.py:123-128

Pinning: the reference holds no test or golden vector for this path, and
TensorFlow is not installable here, so the restatement is pinned by (i) the two independent
formulations above agreeing, (ii) analytic known answers (tests/test_oracle_ops.py), (iii) fp64
gradcheck of the gradient parameters, and -- since round 3 -- (iv) PINNED to the reference's own statements: tests/golden/
ref_ops_golden.npz holds the results of EXECUTING dnnlib/tflib/ops/upfirdn_2d.py (all of the functions above, plus the
gradient definition of _upfirdn_2d_cuda to second order) with a NumPy stand-in for the TF primitives (tests/golden/np_tf.py);
tests/test_ref_ops_golden.py requires this file to reproduce them to 1e-11.  Unpinned remainder: the semantics of tf.nn.conv2d /
conv2d_transpose / tf.pad themselves (restated in np_tf.py from TensorFlow's documentation).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _floor_div(a, b):
    return a // b  # Python floor division == upfirdn_2d.cu:22-28 floorDiv for b > 0


def upfirdn_2d_loops(x, k, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0):
    """x: ndarray [majorDim, inH, inW, minorDim]; k: [kH, kW].  upfirdn_2d.cu:64-117."""
    x = np.asarray(x)
    k = np.asarray(k, dtype=x.dtype)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_w = (in_w * upx + padx0 + padx1 - kw + downx) // downx   # upfirdn_2d.cu:254
    out_h = (in_h * upy + pady0 + pady1 - kh + downy) // downy   # upfirdn_2d.cu:255
    assert out_w >= 1 and out_h >= 1
    y = np.zeros((major, out_h, out_w, minor), dtype=x.dtype)
    for out_y in range(out_h):
        mid_y = out_y * downy + upy - 1 - pady0
        in_y = min(max(_floor_div(mid_y, upy), 0), in_h)
        h = min(max(_floor_div(mid_y + kh, upy), 0), in_h) - in_y
        kernel_y = mid_y + kh - (in_y + 1) * upy
        for out_x in range(out_w):
            mid_x = out_x * downx + upx - 1 - padx0
            in_x = min(max(_floor_div(mid_x, upx), 0), in_w)
            w = min(max(_floor_div(mid_x + kw, upx), 0), in_w) - in_x
            kernel_x = mid_x + kw - (in_x + 1) * upx
            v = np.zeros((major, minor), dtype=x.dtype)
            for yy in range(h):
                for xx in range(w):
                    v += x[:, in_y + yy, in_x + xx, :] * k[kernel_y - yy * upy, kernel_x - xx * upx]
            y[:, out_y, out_x, :] = v
    return y


def upfirdn_2d_ref(x, k, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0):
    """x: torch tensor [majorDim, inH, inW, minorDim] (differentiable).  upfirdn_2d.py:66-101."""
    k = torch.as_tensor(np.asarray(k), dtype=x.dtype)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    # Upsample (insert zeros) :84-86
    x = x.reshape(major, in_h, 1, in_w, 1, minor)
    x = F.pad(x, [0, 0, 0, upx - 1, 0, 0, 0, upy - 1])
    x = x.reshape(major, in_h * upy, in_w * upx, minor)
    # Pad (crop if negative) :89-90
    x = F.pad(x, [0, 0, max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)])
    x = x[:, max(-pady0, 0): x.shape[1] - max(-pady1, 0), max(-padx0, 0): x.shape[2] - max(-padx1, 0), :]
    # Convolve with filter :93-98
    x = x.permute(0, 3, 1, 2)
    x = x.reshape(-1, 1, in_h * upy + pady0 + pady1, in_w * upx + padx0 + padx1)
    w = torch.flip(k, [0, 1])[None, None]
    x = F.conv2d(x, w)
    x = x.reshape(major, minor, in_h * upy + pady0 + pady1 - kh + 1, in_w * upx + padx0 + padx1 - kw + 1)
    x = x.permute(0, 2, 3, 1)
    # Downsample :101
    return x[:, ::downy, ::downx, :]


def upfirdn_2d_grad_params(in_h, in_w, k, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
    """Parameters of the op that computes d/dx (upfirdn_2d.py:119-128,136)."""
    k = np.asarray(k)
    kh, kw = k.shape
    out_w = (in_w * upx + padx0 + padx1 - kw) // downx + 1
    out_h = (in_h * upy + pady0 + pady1 - kh) // downy + 1
    return dict(k=k[::-1, ::-1].copy(), upx=downx, upy=downy, downx=upx, downy=upy,
                padx0=kw - padx0 - 1, pady0=kh - pady0 - 1,
                padx1=in_w * upx - out_w * downx + padx0 - upx + 1,
                pady1=in_h * upy - out_h * downy + pady0 - upy + 1)


# ---------------------------------------------------------------------------- :144-362

def setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    assert k.ndim == 2 and k.shape[0] == k.shape[1]
    return k


def simple_upfirdn_2d(x, k, up=1, down=1, pad0=0, pad1=0):
    """NCHW in/out (upfirdn_2d.py:353-362: reshape to [N*C, H, W, 1])."""
    n, c, h, w = x.shape
    y = x.reshape(n * c, h, w, 1)
    y = upfirdn_2d_ref(y, k, upx=up, upy=up, downx=down, downy=down, padx0=pad0, padx1=pad1, pady0=pad0, pady1=pad1)
    return y.reshape(n, c, y.shape[1], y.shape[2])


def filter_2d(x, k, gain=1):
    k = setup_kernel(k) * gain
    p = k.shape[0] - 1
    return simple_upfirdn_2d(x, k, pad0=(p + 1) // 2, pad1=p // 2)


def upsample_2d(x, k=None, factor=2, gain=1):
    if k is None:
        k = [1] * factor
    k = setup_kernel(k) * (gain * (factor ** 2))
    p = k.shape[0] - factor
    return simple_upfirdn_2d(x, k, up=factor, pad0=(p + 1) // 2 + factor - 1, pad1=p // 2)


def downsample_2d(x, k=None, factor=2, gain=1):
    if k is None:
        k = [1] * factor
    k = setup_kernel(k) * gain
    p = k.shape[0] - factor
    return simple_upfirdn_2d(x, k, down=factor, pad0=(p + 1) // 2, pad1=p // 2)


def upsample_conv_2d(x, w, k=None, factor=2, gain=1):
    """x NCHW, w HWIO.  upfirdn_2d.py:258-292 (num_groups handled as in the reference)."""
    conv_h, conv_w, in_c, out_c = w.shape
    assert conv_w == conv_h
    if k is None:
        k = [1] * factor
    k = setup_kernel(k) * (gain * (factor ** 2))
    p = (k.shape[0] - factor) - (conv_w - 1)
    num_groups = x.shape[1] // in_c
    # Transpose weights :286-288 -> TF conv2d_transpose filter [kh, kw, out, in]
    wt = w.reshape(conv_h, conv_w, in_c, num_groups, -1)
    wt = torch.flip(wt, [0, 1]).permute(0, 1, 4, 3, 2)
    wt = wt.reshape(conv_h, conv_w, -1, num_groups * in_c)
    # tf.nn.conv2d_transpose(x, filter[kh,kw,out,in], strides, VALID) == gradient of conv2d w.r.t. its
    # input == torch.conv_transpose2d with weight[in, out/groups, kh, kw] = filter[kh,kw,out,in] permuted.
    w_t = wt.permute(3, 2, 0, 1)                      # [groups*in_c, out_per_group, kh, kw]
    y = F.conv_transpose2d(x, w_t, stride=factor, groups=num_groups)
    assert y.shape[2] == (x.shape[2] - 1) * factor + conv_h          # output_shape :278
    return simple_upfirdn_2d(y, k, pad0=(p + 1) // 2 + factor - 1, pad1=p // 2 + 1)


def conv_downsample_2d(x, w, k=None, factor=2, gain=1):
    """x NCHW, w HWIO.  upfirdn_2d.py:319-332."""
    conv_h, conv_w, _in_c, _out_c = w.shape
    assert conv_w == conv_h
    if k is None:
        k = [1] * factor
    k = setup_kernel(k) * gain
    p = (k.shape[0] - factor) + (conv_w - 1)
    x = simple_upfirdn_2d(x, k, pad0=(p + 1) // 2, pad1=p // 2)
    return F.conv2d(x, w.permute(3, 2, 0, 1), stride=factor, groups=x.shape[1] // w.shape[2])


def conv2d_same(x, w):
    """tf.nn.conv2d(x, w, strides 1, padding SAME, NCHW) for odd kernels (networks_stylegan2.py:60,120)."""
    kh = w.shape[0]
    return F.conv2d(x, w.permute(3, 2, 0, 1), padding=(kh - 1) // 2, groups=x.shape[1] // w.shape[2])
