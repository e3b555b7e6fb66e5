"""Resampling ops on the HIP path, same public surface as the reference's
`dnnlib/tflib/ops/upfirdn_2d.py` (upfirdn_2d :19-62, filter_2d :144-165, upsample_2d :169-198,
downsample_2d :202-230, upsample_conv_2d :234-292, conv_downsample_2d :296-332).

Differences that are deliberate:
  * tensors are torch (ROCm) tensors; `impl` is 'hip' (there is no in-package 'ref': the CPU
    restatement lives under oracle/ and is test infrastructure only);
  * NCHW inputs are not reshaped to [N*C, H, W, 1] (upfirdn_2d.py:357-361): activations are
    channels_last, so the NCHW view is handed to the kernel as [N, H, W, C] (minorDim = C),
    which is what makes the kernel a coalesced HBM stream on MI355X;
  * conv2d / conv2d_transpose come from the package's own MFMA implicit-GEMM kernel, and the
    transposed conv is stated directly (zero-insert x, correlate with w) instead of through the
    flip + regroup of upfirdn_2d.py:286-288 -- the two are identical (see oracle/ and tests);
  * grouped convolution (the fused-modconv trick) is not offered: modulation is done with
    per-sample operand scales instead (networks_stylegan2.modulated_conv2d_layer).
"""
import numpy as np
import torch

from .... import hip_ops

_IMPLS = ('hip', 'cuda')    # 'cuda' is the reference's name for "the device kernel" (upfirdn_2d.py:19,57-60): same path here


def _check_impl(impl):
    if impl not in _IMPLS:
        raise ValueError("impl must be 'hip' (or the reference's 'cuda'; got %r); the CPU reference lives in oracle/ and is not a product path" % (impl,))

#----------------------------------------------------------------------------

def upfirdn_2d(x, k, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, impl='hip'):
    """Pad, upsample, FIR filter, and downsample a batch of 2D images `[majorDim, inH, inW, minorDim]`.
    Same semantics and argument meaning as upfirdn_2d.py:19-62; gradients of arbitrary order."""
    _check_impl(impl)
    k = np.asarray(k, dtype=np.float32)
    if x.dim() != 4:
        raise ValueError('input must have rank 4')
    for v in (upx, upy, downx, downy, padx0, padx1, pady0, pady1):
        assert isinstance(v, (int, np.integer))
    return hip_ops.UpFirDn2dFn.apply(x, k, int(upx), int(upy), int(downx), int(downy), int(padx0), int(padx1), int(pady0), int(pady1))

#----------------------------------------------------------------------------

def filter_2d(x, k, gain=1, data_format='NCHW', impl='hip'):
    k = _setup_kernel(k) * gain
    p = k.shape[0] - 1
    return _simple_upfirdn_2d(x, k, pad0=(p+1)//2, pad1=p//2, data_format=data_format, impl=impl)

def upsample_2d(x, k=None, factor=2, gain=1, data_format='NCHW', impl='hip'):
    assert isinstance(factor, int) and factor >= 1
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * (gain * (factor ** 2))
    p = k.shape[0] - factor
    return _simple_upfirdn_2d(x, k, up=factor, pad0=(p+1)//2+factor-1, pad1=p//2, data_format=data_format, impl=impl)

def downsample_2d(x, k=None, factor=2, gain=1, data_format='NCHW', impl='hip'):
    assert isinstance(factor, int) and factor >= 1
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * gain
    p = k.shape[0] - factor
    return _simple_upfirdn_2d(x, k, down=factor, pad0=(p+1)//2, pad1=p//2, data_format=data_format, impl=impl)

#----------------------------------------------------------------------------

def _to_nchw(x, data_format):
    assert data_format in ['NCHW', 'NHWC']
    return x if data_format == 'NCHW' else x.permute(0, 3, 1, 2)

def _from_nchw(y, data_format):
    return y if data_format == 'NCHW' else y.permute(0, 2, 3, 1)

def upsample_conv_2d(x, w, k=None, factor=2, gain=1, data_format='NCHW', impl='hip'):
    """Fused `upsample_2d()` followed by conv2d (upfirdn_2d.py:234-292): stride-`factor`
    transposed convolution with `w` [kh, kw, inC, outC], then the FIR with
    pad0=(p+1)//2+factor-1, pad1=p//2+1, p=(firN-factor)-(convW-1)."""
    _check_impl(impl)
    assert isinstance(factor, int) and factor >= 1
    assert w.dim() == 4
    convH, convW, inC, outC = [int(d) for d in w.shape]
    assert convW == convH
    if factor != 2:
        raise NotImplementedError('upsample_conv_2d: only factor=2 is built (the only one the configs use)')
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * (gain * (factor ** 2))
    p = (k.shape[0] - factor) - (convW - 1)
    xn = _to_nchw(x, data_format)
    if xn.shape[1] != inC:
        raise NotImplementedError('upsample_conv_2d: grouped convolution is not offered on the hip path')
    H, W = int(xn.shape[2]), int(xn.shape[3])
    # conv2d_transpose(VALID, stride f, output (H-1)*f + k) of the flipped/regrouped filter (:278,286-291)
    # == zero-insert x by f, pad k-1, cross-correlate with w.
    geom = hip_ops.ConvGeom(convH, convW, 1, factor, convH - 1, convW - 1)
    y = hip_ops.conv2d(xn, w, geom, ((H - 1) * factor + convH, (W - 1) * factor + convW))
    y = _simple_upfirdn_2d(y, k, pad0=(p+1)//2+factor-1, pad1=p//2+1, data_format='NCHW', impl=impl)
    return _from_nchw(y, data_format)

def conv_downsample_2d(x, w, k=None, factor=2, gain=1, data_format='NCHW', impl='hip'):
    """Fused conv2d followed by `downsample_2d()` (upfirdn_2d.py:296-332): FIR with
    pad0=(p+1)//2, pad1=p//2, p=(firN-factor)+(convW-1), then VALID conv with stride `factor`."""
    _check_impl(impl)
    assert isinstance(factor, int) and factor >= 1
    convH, convW, inC, outC = [int(d) for d in w.shape]
    assert convW == convH
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * gain
    p = (k.shape[0] - factor) + (convW - 1)
    xn = _to_nchw(x, data_format)
    xn = _simple_upfirdn_2d(xn, k, pad0=(p+1)//2, pad1=p//2, data_format='NCHW', impl=impl)
    H, W = int(xn.shape[2]), int(xn.shape[3])
    geom = hip_ops.ConvGeom(convH, convW, factor, 1, 0, 0)
    y = hip_ops.conv2d(xn, w, geom, ((H - convH) // factor + 1, (W - convW) // factor + 1))
    return _from_nchw(y, data_format)

#----------------------------------------------------------------------------
# Internal helper funcs.

def _setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    assert k.ndim == 2
    assert k.shape[0] == k.shape[1]
    return k

def _simple_upfirdn_2d(x, k, up=1, down=1, pad0=0, pad1=0, data_format='NCHW', impl='hip'):
    assert data_format in ['NCHW', 'NHWC']
    assert x.dim() == 4
    y = x
    if data_format == 'NCHW':
        y = hip_ops.nhwc(y).permute(0, 2, 3, 1)     # [N,H,W,C] dense view, no copy for channels_last input
    y = upfirdn_2d(y, k, upx=up, upy=up, downx=down, downy=down, padx0=pad0, padx1=pad1, pady0=pad0, pady1=pad1, impl=impl)
    if data_format == 'NCHW':
        y = y.permute(0, 3, 1, 2)                   # logical NCHW, channels_last strides
    return y

#----------------------------------------------------------------------------
