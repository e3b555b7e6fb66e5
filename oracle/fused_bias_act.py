"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

fused_bias_act restated from dnnlib/tflib/ops/fused_bias_act.py:20-30,72-96 (forward, any
activation) and the per-element derivative table of dnnlib/tflib/ops/fused_bias_act.cu:42-116
(`fused_bias_act_kernel_ref`: what the CUDA op returns for grad = 0/1/2 given x, b, ref).
PINNED (round 3) for the table and the forward: tests/golden/ref_ops_golden.npz holds the reference's own activation_funcs
fields and the outputs of its _fused_bias_act_ref executed with a NumPy stand-in for TF (tests/test_ref_ops_golden.py).  The
.cu derivative table cannot be executed here: it stays pinned only by autograd of the forward agreeing with the
grad=1 / grad=2 restatement (tests/test_oracle_ops.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

# name -> (func, def_alpha, def_gain, cuda_idx, ref, zero_2nd_grad)   fused_bias_act.py:20-30
activation_funcs = {
    'linear':   (lambda x, alpha: x,                          None, 1.0,        1, 'y', True),
    'relu':     (lambda x, alpha: F.relu(x),                  None, np.sqrt(2), 2, 'y', True),
    'lrelu':    (lambda x, alpha: F.leaky_relu(x, alpha),     0.2,  np.sqrt(2), 3, 'y', True),
    'tanh':     (lambda x, alpha: torch.tanh(x),              None, 1.0,        4, 'y', False),
    'sigmoid':  (lambda x, alpha: torch.sigmoid(x),           None, 1.0,        5, 'y', False),
    'elu':      (lambda x, alpha: F.elu(x),                   None, 1.0,        6, 'y', False),
    'selu':     (lambda x, alpha: F.selu(x),                  None, 1.0,        7, 'y', False),
    'softplus': (lambda x, alpha: F.softplus(x),              None, 1.0,        8, 'y', False),
    'swish':    (lambda x, alpha: torch.sigmoid(x) * x,       None, np.sqrt(2), 9, 'x', False),
}


def fused_bias_act(x, b=None, axis=1, act='linear', alpha=None, gain=None):
    """fused_bias_act.py:72-96."""
    func, def_alpha, def_gain, _idx, _ref, _z = activation_funcs[act]
    if alpha is None:
        alpha = def_alpha
    if gain is None:
        gain = def_gain
    if b is not None:
        x = x + b.reshape([-1 if i == axis else 1 for i in range(x.dim())])
    x = func(x, alpha)
    if gain != 1:
        x = x * gain
    return x


def fused_bias_act_kernel_ref(x, b, ref, grad, act_idx, alpha, gain, step_b):
    """Element-wise restatement of FusedBiasActKernel (fused_bias_act.cu:50-115) on flat tensors."""
    x = x.reshape(-1).clone()
    n = x.numel()
    if b is not None:
        idx = (torch.arange(n) // step_b) % b.numel()
        x = x + b[idx]
    r = ref.reshape(-1).clone() if ref is not None else torch.zeros_like(x)
    if gain != 0.0 and act_idx != 9:
        r = r / gain
    sel = act_idx * 10 + grad
    seluScale = 1.0507009873554804934193349852946
    seluAlpha = 1.6732632423543772848170429916717
    if sel in (10, 11): y = x
    elif sel in (12, 22, 32): y = torch.zeros_like(x)
    elif sel == 20: y = torch.where(x > 0, x, torch.zeros_like(x))
    elif sel == 21: y = torch.where(r > 0, x, torch.zeros_like(x))
    elif sel == 30: y = torch.where(x > 0, x, x * alpha)
    elif sel == 31: y = torch.where(r > 0, x, x * alpha)
    elif sel == 40: y = torch.tanh(x)
    elif sel == 41: y = x * (1 - r * r)
    elif sel == 42: y = x * (1 - r * r) * (-2 * r)
    elif sel == 50: y = torch.sigmoid(x)
    elif sel == 51: y = x * r * (1 - r)
    elif sel == 52: y = x * r * (1 - r) * (1 - 2 * r)
    elif sel == 60: y = torch.where(x >= 0, x, torch.exp(x) - 1)
    elif sel == 61: y = torch.where(r >= 0, x, x * (r + 1))
    elif sel == 62: y = torch.where(r >= 0, torch.zeros_like(x), x * (r + 1))
    elif sel == 70: y = torch.where(x >= 0, seluScale * x, (seluScale * seluAlpha) * (torch.exp(x) - 1))
    elif sel == 71: y = torch.where(r >= 0, x * seluScale, x * (r + seluScale * seluAlpha))
    elif sel == 72: y = torch.where(r >= 0, torch.zeros_like(x), x * (r + seluScale * seluAlpha))
    elif sel == 80: y = F.softplus(x)
    elif sel == 81: y = x * (1 - torch.exp(-r))
    elif sel == 82: c = torch.exp(-r); y = x * c * (1 - c)
    elif sel == 90: y = x * torch.sigmoid(x)
    elif sel == 91: c = torch.exp(r); d = c + 1; y = x * c * (r + d) / (d * d)
    elif sel == 92: c = torch.exp(r); d = c + 1; y = x * c * (r * (2 - d) + 2 * d) / (d * d * d)
    else: y = x
    return y * gain
