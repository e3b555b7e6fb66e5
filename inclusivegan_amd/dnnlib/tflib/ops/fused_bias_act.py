"""Fused bias + activation on the HIP path; same call surface as the reference's
`dnnlib/tflib/ops/fused_bias_act.py:34-68` (`fused_bias_act(x, b, axis, act, alpha, gain, impl)`).

The activation table carries what the kernel dispatch needs: the kernel's activation index
(fused_bias_act.cu:64-111), default alpha / gain, which tensor the derivative kernels take as
`ref`, and whether the second derivative vanishes (fused_bias_act.py:20-30).
"""
import numpy as np
import torch

from .... import hip_ops
from ...util import EasyDict

activation_funcs = {
    'linear':   EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=1, ref='y', zero_2nd_grad=True),
    'relu':     EasyDict(def_alpha=None, def_gain=np.sqrt(2), hip_idx=2, ref='y', zero_2nd_grad=True),
    'lrelu':    EasyDict(def_alpha=0.2,  def_gain=np.sqrt(2), hip_idx=3, ref='y', zero_2nd_grad=True),
    'tanh':     EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=4, ref='y', zero_2nd_grad=False),
    'sigmoid':  EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=5, ref='y', zero_2nd_grad=False),
    'elu':      EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=6, ref='y', zero_2nd_grad=False),
    'selu':     EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=7, ref='y', zero_2nd_grad=False),
    'softplus': EasyDict(def_alpha=None, def_gain=1.0,        hip_idx=8, ref='y', zero_2nd_grad=False),
    'swish':    EasyDict(def_alpha=None, def_gain=np.sqrt(2), hip_idx=9, ref='x', zero_2nd_grad=False),
}


def fused_bias_act(x, b=None, axis=1, act='linear', alpha=None, gain=None, impl='hip'):
    """y = act(x + b) * gain with `b` broadcast along `axis`.  Twice differentiable for all nine activations, like the
    reference's op (fused_bias_act.py:149-189): the piecewise-linear ones (linear / relu / lrelu -- what R1 and the path-length
    regulariser need) through the closed grad = 1 kernel, the smooth ones through the grad = 2 kernel (hip_ops.FusedBiasActSmoothFn)."""
    if impl not in ('hip', 'cuda'):     # 'cuda' = the reference's name for the device kernel (fused_bias_act.py:34,61-64)
        raise ValueError("impl must be 'hip' (or the reference's 'cuda'; got %r); the CPU reference lives in oracle/ and is not a product path" % (impl,))
    spec = activation_funcs[act]
    if b is not None:
        if b.dim() != 1:
            raise ValueError('b must have rank 1')
        if not (0 <= axis < x.dim()):
            raise ValueError('axis out of bounds')
        if b.shape[0] != x.shape[axis]:
            raise ValueError('b has wrong number of elements')
    if alpha is None:
        alpha = spec.def_alpha
    if gain is None:
        gain = spec.def_gain
    # fused_bias_act.py:116-117
    if act == 'linear' and b is None and gain == 1.0:
        return x
    # piecewise-linear activations on channel-minor data: one-pass forward and one-pass backward
    # (dx and db together) -- csrc/bias_act_noise.hip
    if axis == 1 and spec.hip_idx in (1, 2, 3) and gain > 0 and x.dim() in (2, 4) and x.shape[1] % 4 == 0 and x.dtype == torch.float32:
        return hip_ops.bias_act_noise(x, b, None, None, spec.hip_idx, 0.0 if alpha is None else float(alpha), float(gain))
    if not spec.zero_2nd_grad:      # func_nonzero_2nd_grad (:174-189)
        return hip_ops.FusedBiasActSmoothFn.apply(x, b, axis, spec.hip_idx, 0.0 if alpha is None else float(alpha), float(gain), spec.ref)
    return hip_ops.FusedBiasActFn.apply(x, b, axis, spec.hip_idx, 0.0 if alpha is None else float(alpha), float(gain),
                                        spec.ref, spec.zero_2nd_grad)
