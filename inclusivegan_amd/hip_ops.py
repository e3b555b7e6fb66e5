"""Tensor-level entry to the C ABI plus the autograd closures around it.

Two layers:

* ``*_raw`` functions: take torch (ROCm) tensors, allocate outputs / workspaces
  from torch's caching allocator, pass raw pointers + the current HIP stream to
  ``libigan_hip.so``.  No autograd, no host synchronisation (graph-capturable).
* ``torch.autograd.Function`` closures with the same differentiation structure as
  the reference's ``tf.custom_gradient`` closures:
    - upfirdn_2d: backward is the same op with (up<->down), flipped taps and
      ``gpad*`` (dnnlib/tflib/ops/upfirdn_2d.py:123-139) -> closed under
      differentiation, arbitrary order;
    - fused_bias_act: ``grad=1`` kernel with ``ref`` + bias reduction, second
      order through the same ``grad=1`` kernel for piecewise-linear activations
      (dnnlib/tflib/ops/fused_bias_act.py:126-172);
    - conv2d / its data gradient / its weight gradient form a closed triple
      (conv is bilinear), which is what ``tf.gradients`` of ``tf.nn.conv2d`` /
      ``conv2d_transpose`` gives the reference for R1 and path-length
      regularisation (training/loss.py:64-66,107-111).

All activations are fp32.  4-D activations are logical NCHW with
``torch.channels_last`` strides, i.e. physically [N, H, W, C].
"""
import ctypes
import os
from collections import namedtuple

import numpy as np
import torch

from . import _abi

CL = torch.channels_last


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_cuda_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('inclusivegan_amd kernels need tensors on a ROCm device (got %s); there is no CPU path' % t.device)
        if t.dtype != torch.float32:
            raise TypeError('inclusivegan_amd kernels are fp32 (got %s)' % t.dtype)


def _require_cuda_io(*ts):
    """Like _require_cuda_f32, for the two ops the reference registers for float AND half (UpFirDn2D, FusedBiasAct): all tensors
    of one call share one of the two types."""
    kinds = {t.dtype for t in ts if t is not None}
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('inclusivegan_amd kernels need tensors on a ROCm device (got %s); there is no CPU path' % t.device)
    if len(kinds) != 1 or next(iter(kinds)) not in (torch.float32, torch.float16):
        raise TypeError('upfirdn_2d / fused_bias_act take float32 or float16 tensors of one type (got %s)' % sorted(str(k) for k in kinds))
    return next(iter(kinds)) == torch.float16


def _is_meta(t):
    """Shape-only dry runs (Network template pass) use the meta device and never reach a kernel."""
    return t.device.type == 'meta'


def _needed(ctx, i):
    """Is the gradient w.r.t. input i actually consumed by the running backward pass?
    `ctx.needs_input_grad` is static (requires_grad of the input); when a backward is restricted to some
    inputs -- torch.autograd.grad(scores, [reals]) for R1 (loss.py:108), grad(..., [dlatents]) for the
    path-length term (loss.py:65), backward(inputs=one network's trainables) -- the engine prunes the
    other branches, and computing e.g. a filter gradient for them is wasted MFMA time."""
    if not ctx.needs_input_grad[i]:
        return False
    try:
        # next_functions has one entry per TENSOR input of forward (None / non-tensor arguments are skipped), in order
        mask = getattr(ctx, 'input_is_tensor', None)
        j = i if mask is None else sum(1 for m in mask[:i] if m)
        node = ctx.next_functions[j][0]
        return node is not None and torch._C._will_engine_execute_node(node)
    except Exception:
        return True


def _mark_inputs(ctx, *inputs):
    """Record which forward arguments are tensors, for Functions whose tensor arguments are not a prefix of the argument list
    (optional tensors that may be None, leading configuration arguments): `_needed` needs it to find an input's graph edge."""
    ctx.input_is_tensor = [isinstance(t, torch.Tensor) for t in inputs]


def nhwc(x):
    """Logical NCHW tensor -> same tensor, physically [N,H,W,C] dense."""
    return x.contiguous(memory_format=CL)


def empty_nchw(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32, memory_format=CL)


# ----------------------------------------------------------------------------
# upfirdn2d

def upfirdn2d_raw(x, k, upx, upy, downx, downy, padx0, padx1, pady0, pady1, epilogue=None):
    """x: [majorDim, inH, inW, minorDim] dense.  k: host float32 [kH, kW].
    epilogue = (noise [majorDim or 1, outH, outW] or None, strength, bias [minorDim] or None, act_idx, alpha, gain): the layer epilogue
    fused into the FIR's store (igan_upfirdn2d_ban; FIR fast path only)."""
    k = np.ascontiguousarray(k, dtype=np.float32)
    if k.ndim != 2:
        raise ValueError('kernel must have rank 2')
    if x.dim() != 4:
        raise ValueError('input must have rank 4')
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_w = (in_w * upx + padx0 + padx1 - kw + downx) // downx
    out_h = (in_h * upy + pady0 + pady1 - kh + downy) // downy
    if out_w < 1 or out_h < 1:
        raise ValueError('output must be at least 1x1')
    if _is_meta(x):
        return torch.empty((major, out_h, out_w, minor), device='meta')
    lib = _abi.get_plugin()
    half = _require_cuda_io(x)
    if half and epilogue is not None:
        raise TypeError('upfirdn2d: the fused layer epilogue is a float32 path')
    x = x.contiguous()
    y = torch.empty((major, out_h, out_w, minor), device=x.device, dtype=x.dtype)
    p = _abi.UpFirDn2DParams(
        x=x.data_ptr(), k=k.ctypes.data, y=y.data_ptr(),
        upx=upx, upy=upy, downx=downx, downy=downy,
        padx0=padx0, padx1=padx1, pady0=pady0, pady1=pady1,
        majorDim=major, inH=in_h, inW=in_w, minorDim=minor,
        kernelH=kh, kernelW=kw, outH=out_h, outW=out_w)
    if epilogue is not None:
        noise, strength, bias, act_idx, alpha, gain = epilogue
        _require_cuda_f32(noise, strength, bias)
        if noise is not None:
            noise = noise.contiguous()
            if noise.numel() not in (out_h * out_w, major * out_h * out_w):
                raise ValueError('upfirdn2d epilogue: noise must hold [majorDim or 1, outH, outW] values')
        bcast = 1 if (noise is not None and noise.numel() == out_h * out_w and major > 1) else 0
        _abi.check(lib.igan_upfirdn2d_ban(_stream(), ctypes.byref(p), _ptr(noise), _ptr(strength if noise is not None else None), bcast,
                                          _ptr(bias.contiguous() if bias is not None else None), int(act_idx), float(alpha), float(gain)))
        return y
    _abi.check((lib.igan_upfirdn2d_f16 if half else lib.igan_upfirdn2d)(_stream(), ctypes.byref(p)))
    return y


def fir_ban_fusable(x, k, act_idx):
    """FIR (pad only, <= 4x4 taps) + noise + bias + activation in one pass: first-order path, channel-minor data with C % 4 == 0."""
    k = np.asarray(k)
    return (_FIR_BAN and _second_order_depth == 0 and not _is_meta(x) and x.is_cuda and x.dim() == 4 and x.shape[1] % 4 == 0
            and act_idx in (1, 2, 3) and k.ndim == 2 and k.shape[0] <= 4 and k.shape[1] <= 4)


_FIR_BAN = os.environ.get('IGAN_FIR_BAN', '1') != '0'      # A/B switch


class FirBanFn(torch.autograd.Function):
    """y = act(fir(x; k, pad) + noise * strength + b) * gain on logical-NCHW channels_last tensors: the FIR that follows an
    up-convolution together with the layer's epilogue (networks_stylegan2.py:349-357 with up=True) as one pass -- the filtered,
    pre-activation tensor is never written.  Backward: the one-pass epilogue gradient (dx_pre, db, dstrength), then the FIR's own
    gradient (the same op with the flipped filter, upfirdn_2d.py:123-128).  First order only (see fir_ban_fusable)."""

    @staticmethod
    def forward(ctx, x, k, pad0, pad1, b, noise, strength, act_idx, alpha, gain):
        _mark_inputs(ctx, x, k, pad0, pad1, b, noise, strength, act_idx, alpha, gain)
        k = np.asarray(k, dtype=np.float32)
        xh = nhwc(x).permute(0, 2, 3, 1)                                  # [N, H, W, C] dense view
        y = upfirdn2d_raw(xh, k, 1, 1, 1, 1, pad0, pad1, pad0, pad1, epilogue=(noise, strength, b, act_idx, alpha, gain))
        in_h, in_w = xh.shape[1], xh.shape[2]
        out_h, out_w = y.shape[1], y.shape[2]
        kh, kw = k.shape
        ctx.gargs = (np.ascontiguousarray(k[::-1, ::-1]), 1, 1, 1, 1, kw - pad0 - 1, in_w - out_w + pad0, kh - pad0 - 1, in_h - out_h + pad0)
        y = y.permute(0, 3, 1, 2)                                         # logical NCHW, channels_last strides
        ctx.save_for_backward(y, noise)
        ctx.cfg = (act_idx, alpha, gain)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():
            raise NotImplementedError('fused FIR + epilogue: second-order gradients go through UpFirDn2dFn + BiasActNoiseFn (hip_ops.second_order())')
        y, noise = ctx.saved_tensors
        act_idx, alpha, gain = ctx.cfg
        need_b = ctx.has_b and _needed(ctx, 4)
        need_s = noise is not None and _needed(ctx, 6)
        nz = noise.expand(_noise_view(y)).contiguous() if noise is not None else None
        dxp, db, ds = bias_act_noise_bwd_raw(dy, y, nz, act_idx, alpha, gain, need_b)
        dx = None
        if _needed(ctx, 0):
            dx = upfirdn2d_raw(nhwc(dxp).permute(0, 2, 3, 1), *ctx.gargs).permute(0, 3, 1, 2)
        return dx, None, None, None, (db if need_b else None), None, (ds if need_s else None), None, None, None


class UpFirDn2dFn(torch.autograd.Function):
    """upfirdn_2d.py:130-140: y = op(x; k, up, down, pad); grad = op(dy; flip(k), down, up, gpad)."""

    @staticmethod
    def forward(ctx, x, k, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
        k = np.asarray(k, dtype=np.float32)
        in_h, in_w = x.shape[1], x.shape[2]
        y = upfirdn2d_raw(x, k, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
        out_h, out_w = y.shape[1], y.shape[2]
        kh, kw = k.shape
        # upfirdn_2d.py:125-128
        gpadx0 = kw - padx0 - 1
        gpady0 = kh - pady0 - 1
        gpadx1 = in_w * upx - out_w * downx + padx0 - upx + 1
        gpady1 = in_h * upy - out_h * downy + pady0 - upy + 1
        ctx.gargs = (np.ascontiguousarray(k[::-1, ::-1]), downx, downy, upx, upy, gpadx0, gpadx1, gpady0, gpady1)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = UpFirDn2dFn.apply(dy, *ctx.gargs)
        return (dx,) + (None,) * 9


# ----------------------------------------------------------------------------
# fused_bias_act

def _bias_layout(x, axis):
    """Returns (x_dense, sizeB, stepB) for a bias along `axis` given x's physical layout."""
    if x.dim() == 4 and axis == 1 and x.is_contiguous(memory_format=CL):
        return x, x.shape[1], 1
    xc = x.contiguous()
    step = 1
    for d in xc.shape[axis + 1:]:
        step *= int(d)
    return xc, int(xc.shape[axis]), step


def fused_bias_act_raw(x, b, ref, grad, act_idx, alpha, gain, size_b, step_b):
    """x (and ref) dense in the layout described by (size_b, step_b); returns empty_like(x)."""
    if _is_meta(x):
        return torch.empty_like(x)
    lib = _abi.get_plugin()
    half = _require_cuda_io(x, b, ref)
    y = torch.empty_like(x)
    p = _abi.FusedBiasActParams(
        x=x.data_ptr(), b=(b.data_ptr() if b is not None else None),
        ref=(ref.data_ptr() if ref is not None else None), y=y.data_ptr(),
        grad=grad, act=act_idx, alpha=float(alpha), gain=float(gain),
        sizeX=x.numel(), sizeB=size_b, stepB=step_b)
    _abi.check((lib.igan_fused_bias_act_f16 if half else lib.igan_fused_bias_act)(_stream(), ctypes.byref(p)))
    return y


def bias_grad_raw(dx, size_b, step_b):
    lib = _abi.get_plugin()
    if dx.dtype == torch.float16:      # half instantiation: db = reduce_sum(dx) (fused_bias_act.py:137-146), accumulated in float
        return dx.reshape(-1, size_b, step_b).float().sum(dim=(0, 2)).to(torch.float16)
    _require_cuda_f32(dx)
    n = dx.numel()
    ws_floats = lib.igan_bias_grad_workspace_floats(n, size_b, step_b)
    ws = torch.empty((max(int(ws_floats), 1),), device=dx.device, dtype=torch.float32)
    db = torch.empty((size_b,), device=dx.device, dtype=torch.float32)
    _abi.check(lib.igan_bias_grad(_stream(), _ptr(dx), _ptr(db), _ptr(ws), n, size_b, step_b))
    return db


def _match_layout(t, ref):
    """t with exactly ref's (dense) physical layout."""
    if t.shape == ref.shape and t.stride() == ref.stride():
        return t
    out = torch.empty_like(ref)
    out.copy_(t)
    return out


class _FbaGradFn(torch.autograd.Function):
    """dx = kernel(grad=1)(dy [+ d_db], ref): the first-order gradient as a differentiable op (fused_bias_act.py:161-189).
    Piecewise-linear activations (zero second derivative): closed under differentiation, nothing depends on x.  Smooth activations
    (tanh, sigmoid, elu, selu, softplus, swish): dx also depends on x through f'(x), and its gradient w.r.t. x is
    kernel(grad=2)(d_dx, ref) * dy -- the grad=2 table entry is gain * f''(x) (expressed through x or y, whichever `ref` is), as in
    fused_bias_act.cu.  `xrecv` is the forward input (graph-connected): it only RECEIVES that gradient; `ref` stays a constant, like
    the closed-over y of the reference's grad_impl(dy, x) (:176-188).  Deviation from the reference, on purpose: its grad2_d_x (:165-168)
    hands only d_dx to the grad=2 kernel, so its second-order term lacks the factor dy (its PyTorch successor passes dy); the two
    agree for dy = 1, and what is computed here is the derivative (checked against fp64 autograd in tests/test_gpu_ops.py)."""

    @staticmethod
    def forward(ctx, dy, bias_term, ref, axis, act_idx, alpha, gain, size_b, step_b, zero_2nd, xrecv=None):
        ctx.cfg = (axis, act_idx, alpha, gain, size_b, step_b, zero_2nd)
        ctx.has_bias_term = bias_term is not None
        dyl = _match_layout(dy, ref)
        ctx.save_for_backward(ref, *(() if zero_2nd else (dyl, bias_term) if bias_term is not None else (dyl,)))
        return fused_bias_act_raw(dyl, bias_term, ref, 1, act_idx, alpha, gain, size_b, step_b)

    @staticmethod
    def backward(ctx, d_dx):
        ref = ctx.saved_tensors[0]
        axis, act_idx, alpha, gain, size_b, step_b, zero_2nd = ctx.cfg
        d_dy = _FbaGradFn.apply(d_dx, None, ref, axis, act_idx, alpha, gain, size_b, step_b, True) if zero_2nd else None
        d_x = None
        if not zero_2nd:
            with torch.no_grad():       # third order is not offered
                g = _match_layout(d_dx, ref)
                d_dy = fused_bias_act_raw(g, None, ref, 1, act_idx, alpha, gain, size_b, step_b)
                if ctx.needs_input_grad[10]:
                    dy = ctx.saved_tensors[1]
                    if ctx.has_bias_term:
                        view = [1] * dy.dim(); view[axis] = -1
                        dy = dy + ctx.saved_tensors[2].view(*view)
                    d_x = fused_bias_act_raw(g, None, ref, 2, act_idx, alpha, gain, size_b, step_b) * dy
        d_bias_term = None
        if ctx.has_bias_term:
            d_bias_term = _BiasGradFn.apply(d_dy, axis, size_b, step_b)
        return (d_dy, d_bias_term) + (None,) * 8 + (d_x,)


class _BiasGradFn(torch.autograd.Function):
    """db = reduce_sum(dx) over all axes but the bias axis (fused_bias_act.py:137-146)."""

    @staticmethod
    def forward(ctx, dx, axis, size_b, step_b):
        ctx.shape = dx.shape
        ctx.axis = axis
        return bias_grad_raw(dx, size_b, step_b)

    @staticmethod
    def backward(ctx, d_db):
        view = [1] * len(ctx.shape)
        view[ctx.axis] = -1
        return d_db.view(*view).expand(ctx.shape), None, None, None


class FusedBiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, axis, act_idx, alpha, gain, ref_kind, zero_2nd):
        xd, size_b, step_b = _bias_layout(x, axis)
        y = fused_bias_act_raw(xd, b, None, 0, act_idx, alpha, gain, size_b, step_b)
        ctx.save_for_backward(xd if ref_kind == 'x' else y)
        ctx.cfg = (axis, act_idx, alpha, gain, size_b, step_b, zero_2nd)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (ref,) = ctx.saved_tensors
        axis, act_idx, alpha, gain, size_b, step_b, zero_2nd = ctx.cfg
        dx = _FbaGradFn.apply(dy, None, ref, axis, act_idx, alpha, gain, size_b, step_b, zero_2nd)
        db = _BiasGradFn.apply(dx, axis, size_b, step_b) if (ctx.has_b and _needed(ctx, 1)) else None
        return (dx, db) + (None,) * 6


class FusedBiasActSmoothFn(torch.autograd.Function):
    """FusedBiasActFn for the activations with a non-zero second derivative (the reference's func_nonzero_2nd_grad,
    fused_bias_act.py:174-189): under create_graph the first-order gradient is taken by `_FbaGradFn` with a receiver for the
    second-order term, so the op is twice differentiable w.r.t. x, b and dy.  The receiver is what `ref` is a function of: the
    pre-activation x + b for the activations whose ref is y (tanh, sigmoid, elu, selu, softplus), and the op input x alone for swish --
    the reference hands its kernel ref = x WITHOUT the bias there (fused_bias_act.py:135, fused_bias_act.cu:57-58), and the first-order
    gradient here is that literal value, so its derivative is taken literally too."""

    @staticmethod
    def forward(ctx, x, b, axis, act_idx, alpha, gain, ref_kind):
        xd, size_b, step_b = _bias_layout(x, axis)
        y = fused_bias_act_raw(xd, b, None, 0, act_idx, alpha, gain, size_b, step_b)
        ctx.cfg = (axis, act_idx, alpha, gain, size_b, step_b, ref_kind)
        ctx.has_b = b is not None
        ctx.save_for_backward(xd, y, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, y, b = ctx.saved_tensors
        axis, act_idx, alpha, gain, size_b, step_b, ref_kind = ctx.cfg
        ref = (xd if ref_kind == 'x' else y).detach()
        if torch.is_grad_enabled():
            recv = xd
            if ref_kind != 'x' and b is not None:
                view = [1] * xd.dim(); view[axis] = -1
                recv = xd + b.view(*view)
            dx = _FbaGradFn.apply(dy, None, ref, axis, act_idx, alpha, gain, size_b, step_b, False, recv)
        else:
            dx = fused_bias_act_raw(_match_layout(dy, ref), None, ref, 1, act_idx, alpha, gain, size_b, step_b)
        db = _BiasGradFn.apply(dx, axis, size_b, step_b) if (ctx.has_b and _needed(ctx, 1)) else None
        return (dx, db) + (None,) * 5


# ----------------------------------------------------------------------------
# fused layer epilogue: noise + bias + activation (one pass forward, one pass backward)

def _rows_c(x):
    """(rows, C) of a channel-minor tensor: channels_last 4-D or plain 2-D; None if not eligible."""
    if x.dim() == 4 and x.is_contiguous(memory_format=CL) and x.shape[1] % 4 == 0:
        return x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    if x.dim() == 2 and x.is_contiguous() and x.shape[1] % 4 == 0:
        return x.shape[0], x.shape[1]
    return None


def bias_act_noise_fwd_raw(x, noise, strength, b, act_idx, alpha, gain):
    if _is_meta(x):
        return torch.empty_like(x)
    lib = _abi.get_plugin()
    _require_cuda_f32(x, noise, strength, b)
    rows, c = _rows_c(x)
    y = torch.empty_like(x)
    _abi.check(lib.igan_bias_act_noise_fwd(_stream(), _ptr(x), _ptr(noise), _ptr(strength), _ptr(b), _ptr(y),
                                           rows, c, act_idx, float(alpha), float(gain)))
    return y


def bias_act_noise_bwd_raw(dy, y, noise, act_idx, alpha, gain, want_db):
    lib = _abi.get_plugin()
    _require_cuda_f32(dy, y, noise)
    rows, c = _rows_c(y)
    dy = _match_layout(dy, y)
    dx = torch.empty_like(y)
    ws = torch.empty((int(lib.igan_bias_act_noise_workspace_floats(rows, c)),), device=y.device, dtype=torch.float32)
    db = torch.empty((c,), device=y.device, dtype=torch.float32) if want_db else None
    ds = torch.empty((), device=y.device, dtype=torch.float32) if noise is not None else None
    _abi.check(lib.igan_bias_act_noise_bwd(_stream(), _ptr(dy), _ptr(y), _ptr(noise), _ptr(dx), _ptr(db), _ptr(ds), _ptr(ws),
                                           rows, c, act_idx, float(alpha), float(gain)))
    return dx, db, ds


class BiasActNoiseFn(torch.autograd.Function):
    """y = act(x + noise * strength + b) * gain for act in {linear, relu, lrelu}; noise/strength may be
    None.  First-order backward is the fused one-pass kernel; under create_graph it is written with the
    closed `_FbaGradFn` / `_BiasGradFn` pair (piecewise-linear activations: nothing depends on x)."""

    @staticmethod
    def forward(ctx, x, b, noise, strength, act_idx, alpha, gain):
        _mark_inputs(ctx, x, b, noise, strength, act_idx, alpha, gain)
        y = bias_act_noise_fwd_raw(x, noise, strength, b, act_idx, alpha, gain)
        ctx.save_for_backward(y, noise)
        ctx.cfg = (act_idx, alpha, gain)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, noise = ctx.saved_tensors
        act_idx, alpha, gain = ctx.cfg
        need_b = ctx.has_b and _needed(ctx, 1)
        need_s = noise is not None and _needed(ctx, 3)
        if torch.is_grad_enabled():
            c = y.shape[1]
            dx = _FbaGradFn.apply(dy, None, y, 1, act_idx, alpha, gain, c, 1, True)
            db = _BiasGradFn.apply(dx, 1, c, 1) if need_b else None
            ds = (dx * noise.reshape(_noise_view(y))).sum() if need_s else None
            return dx, db, None, ds, None, None, None
        dx, db, ds = bias_act_noise_bwd_raw(dy, y, noise, act_idx, alpha, gain, need_b)
        return dx, db, None, (ds if need_s else None), None, None, None


def _noise_view(y):
    return (y.shape[0], 1, y.shape[2], y.shape[3]) if y.dim() == 4 else (y.shape[0], 1)


def bias_act_noise(x, b, noise, strength, act_idx, alpha, gain):
    """Dispatch: the fused kernels when x is 2-D / 4-D with C % 4 == 0 (made channel-minor if it is
    not), otherwise the general ops."""
    if _is_meta(x):
        return torch.empty_like(x)
    if x.dim() in (2, 4) and x.shape[1] % 4 == 0:
        x = nhwc(x) if x.dim() == 4 else x.contiguous()
        if noise is not None:
            noise = noise.expand(_noise_view(x)).contiguous()
        return BiasActNoiseFn.apply(x, b, noise, strength, act_idx, alpha, gain)
    if noise is not None:
        x = x + noise * strength
    return FusedBiasActFn.apply(x, b, 1, act_idx, alpha, gain, 'y', True)


# ----------------------------------------------------------------------------
# conv2d family

# Geometry of y = conv(x, w): see include/igan_hip.h (igan_conv2d_params).
ConvGeom = namedtuple('ConvGeom', 'kh kw stride up pad_y pad_x alpha', defaults=(1.0,))   # alpha: output multiplier (runtime weight scale)

_plan_cache = {}
_wplan_cache = {}

# Profiling hook (bench.py): when set to a list, every conv2d_raw launch is bracketed by HIP events on the
# launch stream and logged as (kernel name, algorithmic FLOPs, start event, end event).
launch_log = None
stamp_log = None      # a StampLog: every conv2d_raw launch is bracketed by device-side stamps (works inside captured graphs)
shape_log = None      # tools/conv_shapes.py: (kind, shape key, flops, splits, start event, end event) per conv-family launch


class StampLog:
    """Per-launch device time of the conv family, measured ON the device (include/igan_hip.h igan_stamp): every logged launch
    gets a slot; a pair of one-wave stamp kernels around it reads the 100 MHz counter in stream order, and `fold()` -- called
    at the end of each captured graph / eager op -- adds the slot's duration into an accumulator.  Because stamps and folds
    are ordinary kernels they are captured into the training ops' hipGraphs: replaying a graph K times yields K samples per
    launch, with the device running exactly as in the timed region (no host-side gaps, full clocks)."""

    def __init__(self, device, capacity=16384):
        self.stamps = torch.zeros((2 * capacity,), device=device, dtype=torch.int64)
        self.acc = torch.zeros((capacity,), device=device, dtype=torch.int64)
        self.entries = []           # (kernel name, algorithmic flops) per slot
        self.shapes = []            # free-form shape description per slot (for per-shape tables)
        self.capacity = capacity
        self._folded = 0            # slots below this index already have their fold kernel
        self.folds = {}             # graph tag -> number of times its fold ran is tracked by the caller (replays)

    def bracket(self, name, flops, launch, shape=None):
        i = len(self.entries)
        if i >= self.capacity:
            raise RuntimeError('StampLog: capacity exceeded')
        lib = _abi.get_plugin()
        self.entries.append((name, flops))
        self.shapes.append(shape)
        _abi.check(lib.igan_stamp(_stream(), ctypes.c_void_p(self.stamps.data_ptr() + 16 * i)))
        launch()
        _abi.check(lib.igan_stamp(_stream(), ctypes.c_void_p(self.stamps.data_ptr() + 16 * i + 8)))

    def fold(self):
        """Accumulate the slots created since the last fold (one captured op's launches). Returns (first, count)."""
        first, count = self._folded, len(self.entries) - self._folded
        if count > 0:
            _abi.check(_abi.get_plugin().igan_stamp_accumulate(_stream(), _ptr(self.stamps), _ptr(self.acc), first, count))
            self._folded = len(self.entries)
        return first, count

    def totals_us(self):
        """-> list of (name, flops, accumulated microseconds) per slot (10 ns ticks)."""
        acc = self.acc[:len(self.entries)].cpu().tolist()
        return [(n, f, a * 0.01) for (n, f), a in zip(self.entries, acc)]

    def shape_table(self, replays):
        """Rows (shape, launches per replayed iteration set, total us, TFLOP/s), grouped by shape, sorted by time; `replays`
        maps slot index -> number of replays its graph saw."""
        rows = {}
        for i, ((name, flops, us), shape) in enumerate(zip(self.totals_us(), self.shapes)):
            n = replays.get(i, 0)
            if n == 0:
                continue
            r = rows.setdefault((shape, name), [0, 0.0, 0.0])
            r[0] += n; r[1] += flops * n; r[2] += us
        return sorted(((k[0], k[1], v[0], v[2], v[1] / max(v[2], 1e-9) / 1e6) for k, v in rows.items()), key=lambda r: -r[3])


def _shape_logged(kind, key, flops, splits, launch):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    shape_log.append((kind, key, flops, splits, e0, e1))


def conv_flops(n, h, w, cin, oh, ow, cout, geom):
    """Algorithmic FLOPs of one conv2d call: 2 * (output pixel, tap) pairs that exist * Cin * Cout
    (a transposed conv only counts the taps of each output parity class)."""
    up = geom.up
    pairs = 0
    for py in range(up):
        for px in range(up):
            qh = (oh - py + up - 1) // up
            qw = (ow - px + up - 1) // up
            ky0 = (geom.pad_y - py * geom.stride) % up
            kx0 = (geom.pad_x - px * geom.stride) % up
            nky = (geom.kh - ky0 + up - 1) // up if ky0 < geom.kh else 0
            nkx = (geom.kw - kx0 + up - 1) // up if kx0 < geom.kw else 0
            pairs += qh * qw * nky * nkx
    return 2.0 * n * pairs * cin * cout


PIECES_SHARE = os.environ.get('IGAN_PIECES_SHARE', '1') != '0'
# ^ the piece forms of the large 3x3 convolutions (csrc/conv2d_mfma.hip; two fp16 pieces by default, IGAN_CONV_PLANES=1 three bf16 pieces, =0 switches them off inside
#   the library) keeps its piece images shared between the calls of a layer (IGAN_PIECES_SHARE=0: every convolution call writes its own, A/B switch)

_pieces_rule = {}


def pieces_wanted(geom, cin, cout):
    """Will the convolution calls of a layer with this filter take the piece form (so that writing ONE shared image pays)?  The LIBRARY answers
    (igan_conv_pieces_wanted: the batch-independent part of planes_shape_ok / wgrad_planes_shape_ok), cached per filter shape -- the rules are
    not restated here.  (A call the library then runs in fp32 after all simply ignores the image.)"""
    if not PIECES_SHARE:
        return False
    key = (geom.kh, geom.kw, int(cin), int(cout))
    hit = _pieces_rule.get(key)
    if hit is None:
        hit = _pieces_rule[key] = bool(_abi.get_plugin().igan_conv_pieces_wanted(*key))
    return hit


class PieceImage:
    """A piece image with its byte size: the library checks the size against the tensor it is claimed to be the image of (ABI v6)."""
    __slots__ = ('buf', 'nbytes')

    def __init__(self, buf, nbytes):
        self.buf, self.nbytes = buf, nbytes

    def data_ptr(self):
        return self.buf.data_ptr()


def to_pieces(x, scale=None):
    """Piece image of a channels-last tensor (times scale[n, c]) for the bf16-piece form: written once when several convolution
    calls consume the same tensor (dy in the data and the weight gradient, x in the forward pass and the weight gradient).
    Returns None when the form is off or the tensor is not of a kind the piece kernels take (igan_pieces_image_ok; they then make their
    own image, or run fp32)."""
    if not PIECES_SHARE or _is_meta(x) or x.dim() != 4:
        return None
    n, c, h, w = x.shape
    if not _abi.get_plugin().igan_pieces_image_ok(int(n), int(h * w), int(c)):
        return None
    x = nhwc(x)
    if scale is not None:
        scale = scale.contiguous()
    out = torch.empty((n * h * w * c * 6 // 4,), device=x.device, dtype=torch.float32)
    if (x.data_ptr() | out.data_ptr() | (scale.data_ptr() if scale is not None else 0)) & 15:
        return None
    launch = lambda: _abi.check(_abi.get_plugin().igan_to_pieces(_stream(), _ptr(x), (_ptr(scale) if scale is not None else None), _ptr(out), n, h * w, c))
    if stamp_log is not None:       # part of the piece form's cost: counted in the conv family's time (no FLOPs of its own)
        stamp_log.bracket('to_planes_kernel (shared image)', 0.0, launch, shape='%-12s N%-3d %3dx%-3d C%-4d' % ('pieces', n, h, w, c))
    else:
        launch()
    return PieceImage(out, n * h * w * c * 6)


_colmax_floats = {}
COLMAX_SHARE = os.environ.get('IGAN_COLMAX_SHARE', '1') != '0'      # A/B switch: 0 = the weight gradient finds its operands' channel maxima by passes of its own


def colmax_buffer(x, want=True):
    """fp16 form: the buffer a convolution call fills with the per-channel maxima of the tensor it images per pixel (igan_conv2d_params.x_colmax), to be handed to
    the weight gradient of the same tensor (x_colmax / dy_colmax) -- its column image then needs no pass of its own for them.  None when the library takes none
    for this tensor (igan_colmax_floats() == 0: another form, another channel count) or `want` is false."""
    if not (want and COLMAX_SHARE) or _is_meta(x) or x.dim() != 4:
        return None
    n, c, h, w = x.shape
    key = (int(n), int(h * w), int(c))
    fl = _colmax_floats.get(key)
    if fl is None:
        fl = _colmax_floats[key] = int(_abi.get_plugin().igan_colmax_floats(*key))
    if fl == 0:
        return None
    return torch.empty((fl,), device=x.device, dtype=torch.float32)


_filter_bytes = {}
FILTER_CACHE = os.environ.get('IGAN_FILTER_CACHE', '1') != '0'      # A/B switch: 0 = every call images its filter itself, constant or not


def mark_constant(w):
    """Declare a weight tensor a constant of the run (the LPIPS network's filters): the convolution calls on it keep ONE filter image per orientation on the
    tensor object instead of writing one per call (ABI v9 w_pieces).  An in-place change of the tensor (its version counter) drops the images."""
    w._igan_constant = True
    return w


def _constant_filter_image(lib, w, geom, cin, cout, w_transposed):
    """The cached image of a constant filter for a call with this (Cin, Cout, orientation), or None (not a constant, no piece form for this filter, or the
    first use falls inside a stream capture: an allocation made there belongs to the graph's pool and cannot be kept)."""
    if not (FILTER_CACHE and getattr(w, '_igan_constant', False)):
        return None
    key = (geom.kh, geom.kw, int(cin), int(cout))
    nbytes = _filter_bytes.get(key)
    if nbytes is None:
        nbytes = _filter_bytes[key] = int(lib.igan_filter_image_bytes(*key))
    if nbytes == 0 or (w.data_ptr() & 15):
        return None
    cache = getattr(w, '_igan_filter_images', None)
    if cache is None or cache[0] != w._version:
        if cache is not None:        # a captured hipGraph may still read the old images: they stay allocated (a constant changes a handful of times per run at most)
            w._igan_filter_keepalive = getattr(w, '_igan_filter_keepalive', []) + list(cache[1].values())
        cache = (w._version, {})
        w._igan_filter_images = cache
    img = cache[1].get((key, bool(w_transposed)))
    if img is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        img = torch.empty(((nbytes + 3) // 4,), device=w.device, dtype=torch.float32)
        _abi.check(lib.igan_filter_image(_stream(), _ptr(w), _ptr(img), geom.kh, geom.kw, int(cin), int(cout), 1 if w_transposed else 0))
        cache[1][(key, bool(w_transposed))] = img
    return PieceImage(img, nbytes)


def conv2d_raw(x, w, geom, out_hw, cout, w_transposed=False, in_scale=None, out_scale=None, bias=None, act=None, noise=None, strength=None, x_pieces=None, colmax=None):
    """x: logical [N,Cin,H,W] (channels_last).  w: HWIO [KH,KW,Cin,Cout] (or the forward layer's
    [KH,KW,Cout,Cin] when w_transposed).  Returns logical [N,Cout,OH,OW] channels_last.
    act = (act_idx, alpha, gain) fuses y = act(y + bias) * gain into the kernel's epilogue (bias may be None)."""
    if _is_meta(x):
        return torch.empty((x.shape[0], cout, out_hw[0], out_hw[1]), device='meta')
    lib = _abi.get_plugin()
    _require_cuda_f32(x, w, in_scale, out_scale, bias)
    x = nhwc(x)
    w = w.contiguous()
    n, cin, h, wd = x.shape
    oh, ow = out_hw
    y = empty_nchw(n, cout, oh, ow, x)
    if bias is not None:
        bias = bias.contiguous()
    if in_scale is not None:
        in_scale = in_scale.contiguous()
    if out_scale is not None:
        out_scale = out_scale.contiguous()
    p = _abi.Conv2DParams(
        x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(),
        in_scale=(in_scale.data_ptr() if in_scale is not None else None),
        out_scale=(out_scale.data_ptr() if out_scale is not None else None),
        workspace=None, workspace_floats=0,
        N=n, H=h, W=wd, Cin=cin, OH=oh, OW=ow, Cout=cout,
        KH=geom.kh, KW=geom.kw, stride=geom.stride, up=geom.up,
        pad_y=geom.pad_y, pad_x=geom.pad_x,
        w_transposed=1 if w_transposed else 0, splits=1, alpha=float(geom.alpha),
        bias=(bias.data_ptr() if bias is not None else None), act=(int(act[0]) if act is not None else 0),
        act_alpha=(float(act[1]) if act is not None else 0.0), act_gain=(float(act[2]) if act is not None else 1.0))
    if x_pieces is not None:    # bf16-piece form: the image of x * in_scale, written once by to_pieces() for several consumers
        p.x_pieces = x_pieces.data_ptr()
        p.x_pieces_bytes = x_pieces.nbytes
    if colmax is not None:      # fp16 form: the channel maxima of x * in_scale as a by-product (colmax_buffer)
        p.x_colmax = colmax.data_ptr()
    wimg = _constant_filter_image(lib, w, geom, cin, cout, w_transposed)
    if wimg is not None:        # a constant filter's image, written once (mark_constant)
        p.w_pieces = wimg.data_ptr()
        p.w_pieces_bytes = wimg.nbytes
    if noise is not None:       # epilogue noise: [N or 1, 1, OH, OW] contiguous + device scalar strength (needs act)
        noise = noise.contiguous()
        _require_cuda_f32(noise, strength)
        if act is None or strength is None or noise.numel() not in (oh * ow, n * oh * ow):
            raise ValueError('conv2d: noise needs the fused epilogue, a strength scalar and [N or 1, 1, OH, OW] values')
        p.noise = noise.data_ptr(); p.noise_strength = strength.data_ptr(); p.noise_bcast = 1 if (noise.numel() == oh * ow and n > 1) else 0
    key = (n, h, wd, cin, oh, ow, cout, geom, act is not None, bool(w_transposed), in_scale is not None, out_scale is not None)
    plan = _plan_cache.get(key)
    if plan is None:
        splits = ctypes.c_int(1)
        sliced = ctypes.c_int(0)
        wsf = ctypes.c_size_t(0)
        _abi.check(lib.igan_conv2d_plan(ctypes.byref(p), ctypes.byref(splits), ctypes.byref(sliced), ctypes.byref(wsf)))
        plan = (splits.value, wsf.value, sliced.value)
        _plan_cache[key] = plan
    ws = None
    if plan[1] > 0:     # partial tiles of the sliced tail and / or the piece images of the bf16-piece form
        ws = torch.empty((plan[1],), device=x.device, dtype=torch.float32)
        p.workspace = ws.data_ptr()
        p.workspace_floats = plan[1]
    if plan[0] > 1:
        p.splits = plan[0]
        p.sliced_tiles = plan[2]
    if stamp_log is not None:
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
        kind = ('dgrad' if w_transposed else 'fwd') + ('+scale' if (in_scale is not None or out_scale is not None) else '') + ('+act' if act is not None else '')
        stamp_log.bracket(buf.value.decode(), conv_flops(n, h, wd, cin, oh, ow, cout, geom),
                          lambda: _abi.check(lib.igan_conv2d(_stream(), ctypes.byref(p))),
                          shape='%-12s N%-3d %3dx%-3d C%-4d -> %3dx%-3d C%-4d k%d s%d u%d splits %d' % (kind, n, h, wd, cin, oh, ow, cout, geom.kh, geom.stride, geom.up, plan[0]))
        return y
    if launch_log is not None:
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _abi.check(lib.igan_conv2d(_stream(), ctypes.byref(p)))
        e1.record()
        launch_log.append((buf.value.decode(), conv_flops(n, h, wd, cin, oh, ow, cout, geom), plan[0], e0, e1))
        return y
    if shape_log is not None:
        kind = ('dgrad' if w_transposed else 'fwd') + ('+s' if in_scale is not None else '')
        _shape_logged(kind, key, conv_flops(n, h, wd, cin, oh, ow, cout, geom), plan[0],
                      lambda: _abi.check(lib.igan_conv2d(_stream(), ctypes.byref(p))))
        return y
    _abi.check(lib.igan_conv2d(_stream(), ctypes.byref(p)))
    return y


def conv2d_wgrad_raw(x, dy, geom, in_scale=None, out_scale=None, x_pieces=None, dy_pieces=None, x_colmax=None, dy_colmax=None):
    """dw[KH,KW,Cin,Cout] for y = conv(x, w) with geometry `geom`."""
    lib = _abi.get_plugin()
    _require_cuda_f32(x, dy, in_scale, out_scale)
    x = nhwc(x)
    dy = nhwc(dy)
    n, cin, h, wd = x.shape
    n2, cout, oh, ow = dy.shape
    if n2 != n:
        raise ValueError('conv2d_wgrad: batch mismatch')
    dw = torch.empty((geom.kh, geom.kw, cin, cout), device=x.device, dtype=torch.float32)
    if in_scale is not None:
        in_scale = in_scale.contiguous()
    if out_scale is not None:
        out_scale = out_scale.contiguous()
    p = _abi.Conv2DWgradParams(
        x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(),
        in_scale=(in_scale.data_ptr() if in_scale is not None else None),
        out_scale=(out_scale.data_ptr() if out_scale is not None else None),
        workspace=None, workspace_floats=0,
        N=n, H=h, W=wd, Cin=cin, OH=oh, OW=ow, Cout=cout,
        KH=geom.kh, KW=geom.kw, stride=geom.stride, up=geom.up,
        pad_y=geom.pad_y, pad_x=geom.pad_x, splits=1, alpha=float(geom.alpha))
    if x_pieces is not None:
        p.x_pieces = x_pieces.data_ptr()
        p.x_pieces_bytes = x_pieces.nbytes
    if dy_pieces is not None:
        p.dy_pieces = dy_pieces.data_ptr()
        p.dy_pieces_bytes = dy_pieces.nbytes
    if x_colmax is not None:    # fp16 form: channel maxima written by the calls that imaged the same x * in_scale / dy * out_scale per pixel
        p.x_colmax = x_colmax.data_ptr()
    if dy_colmax is not None:
        p.dy_colmax = dy_colmax.data_ptr()
    key = (n, h, wd, cin, oh, ow, cout, geom, in_scale is not None, out_scale is not None)
    plan = _wplan_cache.get(key)
    if plan is None:
        splits = ctypes.c_int(1)
        wsf = ctypes.c_size_t(0)
        _abi.check(lib.igan_conv2d_wgrad_plan(ctypes.byref(p), ctypes.byref(splits), ctypes.byref(wsf)))
        plan = (splits.value, wsf.value)
        _wplan_cache[key] = plan
    ws = None
    if plan[1] > 0:     # partial filters of the pixel slices and / or the piece images of the bf16-piece form
        ws = torch.empty((plan[1],), device=x.device, dtype=torch.float32)
        p.workspace = ws.data_ptr()
        p.workspace_floats = plan[1]
    if plan[0] > 1:
        p.splits = plan[0]
    if stamp_log is not None:
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_wgrad_kernel_name(ctypes.byref(p), buf, 128))
        stamp_log.bracket(buf.value.decode() + ' (+ reduce)', conv_flops(n, h, wd, cin, oh, ow, cout, geom),
                          lambda: _abi.check(lib.igan_conv2d_wgrad(_stream(), ctypes.byref(p))),
                          shape='%-12s N%-3d %3dx%-3d C%-4d -> %3dx%-3d C%-4d k%d s%d u%d splits %d' % (
                              'wgrad' + ('+scale' if (in_scale is not None or out_scale is not None) else ''), n, h, wd, cin, oh, ow, cout, geom.kh, geom.stride, geom.up, plan[0]))
        return dw
    if shape_log is not None:
        _shape_logged('wgrad' + ('+s' if in_scale is not None or out_scale is not None else ''), key,
                      conv_flops(n, h, wd, cin, oh, ow, cout, geom), plan[0],
                      lambda: _abi.check(lib.igan_conv2d_wgrad(_stream(), ctypes.byref(p))))
        return dw
    _abi.check(lib.igan_conv2d_wgrad(_stream(), ctypes.byref(p)))
    return dw


def dgrad_geom(g):
    """Data-gradient geometry of a forward geometry (include/igan_hip.h)."""
    return ConvGeom(g.kh, g.kw, g.up, g.stride, g.kh - 1 - g.pad_y, g.kw - 1 - g.pad_x, g.alpha)


class Conv2dFn(torch.autograd.Function):
    """y = conv(x, w; geom).  x: [N,Cin,H,W], w: HWIO."""

    @staticmethod
    def forward(ctx, x, w, geom, out_hw):
        ctx.save_for_backward(x, w)
        ctx.geom = geom
        ctx.in_hw = (x.shape[2], x.shape[3])
        return conv2d_raw(x, w, geom, out_hw, w.shape[3])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        if _needed(ctx, 0):
            dx = ConvDgradFn.apply(dy, w, ctx.geom, ctx.in_hw)
        if _needed(ctx, 1):
            dw = ConvWgradFn.apply(x, dy, ctx.geom)
        return dx, dw, None, None


class ConvDgradFn(torch.autograd.Function):
    """dx = conv(dy, flip/transpose(w); mirrored geom): the data gradient of Conv2dFn."""

    @staticmethod
    def forward(ctx, dy, w, geom, in_hw):
        ctx.save_for_backward(dy, w)
        ctx.geom = geom
        ctx.out_hw = (dy.shape[2], dy.shape[3])
        return conv2d_raw(dy, w, dgrad_geom(geom), in_hw, w.shape[2], w_transposed=True)

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        d_dy = d_w = None
        if _needed(ctx, 0):
            d_dy = Conv2dFn.apply(ddx, w, ctx.geom, ctx.out_hw)
        if _needed(ctx, 1):
            d_w = ConvWgradFn.apply(ddx, dy, ctx.geom)
        return d_dy, d_w, None, None


class ConvWgradFn(torch.autograd.Function):
    """dw = wgrad(x, dy; geom): the filter gradient of Conv2dFn."""

    @staticmethod
    def forward(ctx, x, dy, geom):
        ctx.save_for_backward(x, dy)
        ctx.geom = geom
        return conv2d_wgrad_raw(x, dy, geom)

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        d_x = d_dy = None
        if _needed(ctx, 0):
            d_x = ConvDgradFn.apply(dy, ddw, ctx.geom, (x.shape[2], x.shape[3]))
        if _needed(ctx, 1):
            d_dy = Conv2dFn.apply(x, ddw, ctx.geom, (dy.shape[2], dy.shape[3]))
        return d_x, d_dy, None


class ConvBiasActFn(torch.autograd.Function):
    """y = act(conv(x, w; geom) + b) * gain with the bias / activation in the convolution's epilogue (the
    apply_bias_act that follows conv2d_layer, networks_stylegan2.py:66-68): the pre-activation tensor is never written.
    act_idx in {1 linear, 2 relu, 3 lrelu}.  Backward: the one-pass epilogue gradient (dx_pre, db), then the data /
    weight gradient kernels; under create_graph the same through the closed Function pairs."""

    @staticmethod
    def forward(ctx, x, w, b, geom, out_hw, act_idx, alpha, gain):
        share = pieces_wanted(geom, x.shape[1], w.shape[3]) and ctx.needs_input_grad[1]
        xp = to_pieces(x) if share else None      # bf16 form: one piece image for the forward pass and the weight gradient
        xcm = colmax_buffer(x, share)             # fp16 form: the forward call's row image leaves the channel maxima the weight gradient's column image needs
        y = conv2d_raw(x, w, geom, out_hw, w.shape[3], bias=b, act=(act_idx, alpha, gain), x_pieces=xp, colmax=xcm)
        ctx.save_for_backward(x, w, y)
        ctx.xp, ctx.xcm = xp, xcm
        ctx.geom, ctx.cfg = geom, (act_idx, alpha, gain)
        ctx.in_hw = (x.shape[2], x.shape[3])
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act_idx, alpha, gain = ctx.cfg
        need_x, need_w = _needed(ctx, 0), _needed(ctx, 1)
        need_b = ctx.has_b and _needed(ctx, 2)
        dx = dw = db = None
        if torch.is_grad_enabled():
            c = y.shape[1]
            dxp = _FbaGradFn.apply(dy, None, y, 1, act_idx, alpha, gain, c, 1, True)
            if need_b:
                db = _BiasGradFn.apply(dxp, 1, c, 1)
            if need_x:
                dx = ConvDgradFn.apply(dxp, w, ctx.geom, ctx.in_hw)
            if need_w:
                dw = ConvWgradFn.apply(x, dxp, ctx.geom)
            return dx, dw, db, None, None, None, None, None
        dxp, db, _ = bias_act_noise_bwd_raw(dy, y, None, act_idx, alpha, gain, need_b)
        share = pieces_wanted(ctx.geom, x.shape[1], y.shape[1]) and need_x and need_w
        dyp = to_pieces(dxp) if share else None                # bf16 form: one image of dy for both gradients
        dycm = colmax_buffer(dxp, share)
        if need_x:
            dx = conv2d_raw(dxp, w, dgrad_geom(ctx.geom), ctx.in_hw, w.shape[2], w_transposed=True, x_pieces=dyp, colmax=dycm)
        if need_w:
            dw = conv2d_wgrad_raw(x, dxp, ctx.geom, x_pieces=ctx.xp, dy_pieces=dyp, x_colmax=ctx.xcm, dy_colmax=dycm)
        return dx, dw, (db if need_b else None), None, None, None, None, None


def conv_bias_act_fusable(x, cout, act_idx):
    """The fused epilogue runs on the MFMA tiles; thin-channel layers (Cin <= 4) and meta dry runs keep the two-step path."""
    return (not _is_meta(x)) and x.is_cuda and x.dim() == 4 and x.shape[1] > 4 and cout % 4 == 0 and act_idx in (1, 2, 3) and _CONV_EPILOGUE


_CONV_EPILOGUE = os.environ.get('IGAN_CONV_EPILOGUE', '1') != '0'      # A/B switch


def conv2d(x, w, geom, out_hw):
    return Conv2dFn.apply(x, w, geom, out_hw)


def matmul(x, w, alpha=1.0):
    """alpha * tf.matmul(x[N,in], w[in,out]) (networks_stylegan2.py:46) as a 1x1 conv on [N,in,1,1]."""
    n, cin = x.shape
    y = Conv2dFn.apply(x.reshape(n, cin, 1, 1), w.reshape(1, 1, cin, w.shape[1]), ConvGeom(1, 1, 1, 1, 0, 0, float(alpha)), (1, 1))
    return y.reshape(n, w.shape[1])


# ----------------------------------------------------------------------------
# modulated conv (fused scales on the first-order path)

def scale_dot_raw(a, b, s=None, want_scaled=False):
    """dot[n,c] = sum_hw a*b for channels_last a, b [N,C,H,W]; optionally also b*s[n,c] (written over b)."""
    lib = _abi.get_plugin()
    _require_cuda_f32(a, b, s)
    a = nhwc(a)
    b = nhwc(b)
    n, c, h, w = a.shape
    dot = torch.empty((n, c), device=a.device, dtype=torch.float32)
    ws = torch.empty((int(lib.igan_scale_dot_workspace_floats(n, h * w, c)),), device=a.device, dtype=torch.float32)
    if s is not None:
        s = s.contiguous()
    _abi.check(lib.igan_scale_dot(_stream(), _ptr(a), _ptr(b), _ptr(s), _ptr(b) if want_scaled else ctypes.c_void_p(0),
                                  _ptr(dot), _ptr(ws), n, h * w, c))
    return dot, (b if want_scaled else None)


class ModConv2dFn(torch.autograd.Function):
    """y = d * conv(x * s, w): modulated_conv2d_layer in its non-fused form
    (networks_stylegan2.py:112,126) with the two scalings folded into the MFMA kernel's operand
    load / epilogue.  d may be None (demodulate=False, ToRGB).

    First-order backward uses the same fused kernels.  When the backward itself has to be
    differentiated (create_graph=True: path-length regulariser, loss.py:64-66) it is re-expressed
    through the closed Conv2dFn / ConvDgradFn / ConvWgradFn triple and differentiable elementwise
    ops, so gradients of any order are exact."""

    @staticmethod
    def forward(ctx, x, w, s, d, geom, out_hw):
        share = pieces_wanted(geom, x.shape[1], w.shape[3]) and ctx.needs_input_grad[1]
        xp = to_pieces(x, s) if share else None   # bf16 form: one piece image of x * s for the forward pass and the weight gradient
        xcm = colmax_buffer(x, share)             # fp16 form: channel maxima of x * s for the weight gradient's column image
        y = conv2d_raw(x, w, geom, out_hw, w.shape[3], in_scale=s, out_scale=d, x_pieces=xp, colmax=xcm)
        ctx.save_for_backward(x, w, s, d, y)
        ctx.xp, ctx.xcm = xp, xcm
        ctx.geom = geom
        ctx.out_hw = out_hw
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s, d, y = ctx.saved_tensors
        geom = ctx.geom
        need_x, need_w, need_s, need_d = [_needed(ctx, i) for i in range(4)]
        in_hw = (x.shape[2], x.shape[3])
        if torch.is_grad_enabled():
            # create_graph=True: the partial derivatives w.r.t. (x, w, s, d) -- each treated as an
            # independent input, although d is itself a function of (s, w) upstream -- written with
            # differentiable pieces (the closed conv triple + elementwise ops), so that the result can
            # be differentiated again.  (A nested autograd.grad over a recomputed forward would follow
            # d's history into s and w and count that path twice.)
            dx = dw = ds = dd = None
            if (not need_w) and d is not None and _MODCONV_GRAD_FN and x.shape[1] % 4 == 0 and y.shape[1] % 4 == 0:
                # path-length regulariser (loss.py:64-66): only (dx, ds, dd) are wanted and must be differentiable once more --
                # the fused first-order kernels as a Function whose own backward is written out with the same kernels
                dx, ds, dd = ModConvGradFn.apply(dy, x, w, s, d, y, geom, in_hw, ctx.out_hw)
                return (dx if need_x else None), None, (ds if need_s else None), (dd if need_d else None), None, None
            dyd = dy * d[:, :, None, None] if d is not None else dy
            if need_x or need_s:
                dxs = ConvDgradFn.apply(dyd, w, geom, in_hw)
                if need_x:
                    dx = dxs * s[:, :, None, None]
                if need_s:
                    ds = (dxs * x).sum(dim=(2, 3))
            if need_w or (need_d and d is not None):
                xs = x * s[:, :, None, None]
                if need_w:
                    dw = ConvWgradFn.apply(xs, dyd, geom)
                if need_d and d is not None:
                    dd = (dy * Conv2dFn.apply(xs, w, geom, ctx.out_hw)).sum(dim=(2, 3))
            return dx, dw, ds, dd, None, None
        dx = dw = ds = dd = None
        dy = nhwc(dy)
        share = pieces_wanted(geom, x.shape[1], y.shape[1]) and (need_x or need_s) and need_w
        dyp = to_pieces(dy, d) if share else None  # bf16 form: one image of dy * d for both gradients
        dycm = colmax_buffer(dy, share)
        if need_x or need_s:
            # dxs = dgrad(dy * d, w)   (un-modulated input gradient)
            dxs = conv2d_raw(dy, w, dgrad_geom(geom), in_hw, w.shape[2], w_transposed=True, in_scale=d, x_pieces=dyp, colmax=dycm)
            if x.shape[1] % 4 == 0:
                # one pass: ds = sum_hw x * dxs, and dx = dxs * s written in place over dxs
                ds, dx = scale_dot_raw(x, dxs, s, want_scaled=need_x)
                if not need_s:
                    ds = None
            else:
                if need_s:
                    ds = (x * dxs).sum(dim=(2, 3))
                if need_x:
                    dx = dxs * s[:, :, None, None]
        if need_w:
            dw = conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d, x_pieces=ctx.xp, dy_pieces=dyp, x_colmax=ctx.xcm, dy_colmax=dycm)
        if need_d and d is not None:
            if y.shape[1] % 4 == 0:
                dd = scale_dot_raw(dy, y)[0] / d
            else:
                dd = (dy * y).sum(dim=(2, 3)) / d
        return dx, dw, ds, dd, None, None


_MODCONV_GRAD_FN = os.environ.get('IGAN_MODCONV_GRAD_FN', '1') != '0'      # A/B switch


class ModConvGradFn(torch.autograd.Function):
    """(dx, ds, dd) = gradients of y = d * conv(x * s, w) w.r.t. (x, s, d) for an upstream dy, as ONE differentiable op: the backward
    of ModConv2dFn when it has to be differentiated again (create_graph=True: path-length regulariser, loss.py:64-66).
        u = dy * d,   t = dgrad(u, w),   dx = t * s,   ds = sum_hw t * x,   dd = sum_hw dy * z,   z = conv(x * s, w) = y / d
    forward = the fused first-order kernels.  backward (cotangents g_dx, g_ds, g_dd; not differentiable again):
        g_t  = g_dx * s + g_ds * x
        g_dy = d * conv(g_t, w) + g_dd * z                      g_d = sum_hw conv(g_t, w) * dy
        g_w  = wgrad(g_t, u) + wgrad(x * s, g_dd * dy)
        g_x  = g_ds * t + s * dgrad(g_dd * dy, w)               g_s = sum_hw g_dx * t + sum_hw x * dgrad(g_dd * dy, w)
    -- every convolution and every reduction over pixels is one of the library's kernels with the scales folded in, instead of
    the dozens of element-wise passes per layer the generic composite needs."""

    @staticmethod
    def forward(ctx, dy, x, w, s, d, y, geom, in_hw, out_hw):
        dy = nhwc(dy)
        t = conv2d_raw(dy, w, dgrad_geom(geom), in_hw, w.shape[2], w_transposed=True, in_scale=d)
        ds, _ = scale_dot_raw(x, t, None)
        dx = t * s[:, :, None, None]
        dd = scale_dot_raw(dy, y)[0] / d
        ctx.save_for_backward(dy, x, w, s, d, y, t)
        ctx.geom, ctx.in_hw, ctx.out_hw = geom, in_hw, out_hw
        return dx, ds, dd

    @staticmethod
    def backward(ctx, g_dx, g_ds, g_dd):
        if torch.is_grad_enabled():
            raise NotImplementedError('modulated conv: third-order gradients are not built')
        dy, x, w, s, d, y, t = ctx.saved_tensors
        geom = ctx.geom
        need_dy, need_x, need_w, need_s, need_d = [_needed(ctx, i) for i in range(5)]
        g_dy = g_x = g_w = g_s = g_d = None
        if g_dx is not None or g_ds is not None:
            if g_dx is not None:
                g_t = nhwc(g_dx) * s[:, :, None, None]
                if g_ds is not None:
                    g_t = torch.addcmul(g_t, x, g_ds[:, :, None, None])
            else:
                g_t = x * g_ds[:, :, None, None]
            if need_dy or need_d:
                g_dy = conv2d_raw(g_t, w, geom, ctx.out_hw, w.shape[3], out_scale=d)          # d * conv(g_t, w)
                if need_d:
                    g_d = scale_dot_raw(g_dy, dy)[0] / d
            if need_w:
                g_w = conv2d_wgrad_raw(g_t, dy, geom, out_scale=d)
            if need_s and g_dx is not None:
                g_s = scale_dot_raw(g_dx, t)[0]
            if need_x and g_ds is not None:
                g_x = t * g_ds[:, :, None, None]
        if g_dd is not None:
            g_dd = g_dd.contiguous()
            if need_x or need_s:
                g_xs = conv2d_raw(dy, w, dgrad_geom(geom), ctx.in_hw, w.shape[2], w_transposed=True, in_scale=g_dd)
                gs2, gx2 = scale_dot_raw(x, g_xs, s, want_scaled=need_x)
                if need_s:
                    g_s = gs2 if g_s is None else g_s + gs2
                if need_x:
                    g_x = gx2 if g_x is None else g_x + gx2
            if need_w:
                gw2 = conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=g_dd)
                g_w = gw2 if g_w is None else g_w + gw2
            if need_dy:
                gdy2 = y * (g_dd / d)[:, :, None, None]
                g_dy = gdy2 if g_dy is None else g_dy + gdy2
        return g_dy, g_x, g_w, g_s, g_d, None, None, None, None


def bias_act_noise_bwd_dd_raw(dy, y, noise, strength, b, d, act_idx, alpha, gain):
    """include/igan_hip.h igan_bias_act_noise_bwd_dd: (dx, db, dstrength, dd) of the epilogue fused into a modulated convolution."""
    lib = _abi.get_plugin()
    _require_cuda_f32(dy, y, noise, strength, b, d)
    y = nhwc(y)
    dy = nhwc(dy)
    n, c, h, w = y.shape
    dx = torch.empty_like(y)
    ws = torch.empty((int(lib.igan_bias_act_noise_dd_workspace_floats(n, h * w, c)),), device=y.device, dtype=torch.float32)
    db = torch.empty((c,), device=y.device, dtype=torch.float32)
    ds = torch.empty((), device=y.device, dtype=torch.float32) if noise is not None else None
    dd = torch.empty((n, c), device=y.device, dtype=torch.float32)
    bcast = 1 if (noise is not None and noise.numel() == h * w and n > 1) else 0
    _abi.check(lib.igan_bias_act_noise_bwd_dd(_stream(), _ptr(dy), _ptr(y), _ptr(noise), _ptr(strength), _ptr(b), _ptr(d.contiguous()), _ptr(dx), _ptr(db),
                                              _ptr(ds), _ptr(dd), _ptr(ws), bcast, n, h * w, c, act_idx, float(alpha), float(gain)))
    return dx, db, ds, dd


class ModConvBanFn(torch.autograd.Function):
    """y = act(d * conv(x * s, w) + noise * strength + b) * gain: a whole synthesis layer that has no FIR between the convolution and
    its epilogue (`layer()` with up=False, networks_stylegan2.py:349-357) as one kernel forward -- noise, bias and activation run in
    the convolution's epilogue, the pre-activation tensor is never written -- and, backward, one pass that turns dy into the
    gradient w.r.t. the convolution output together with db, dstrength AND the demodulation gradient dd (the pre-activation value
    is recovered from y: linear / lrelu only), followed by the data / style / weight gradient kernels of ModConv2dFn.
    First order only: under hip_ops.second_order() the layer is built from ModConv2dFn + BiasActNoiseFn instead."""

    @staticmethod
    def forward(ctx, x, w, s, d, b, noise, strength, geom, out_hw, act_idx, alpha, gain):
        _mark_inputs(ctx, x, w, s, d, b, noise, strength, geom, out_hw, act_idx, alpha, gain)
        share = pieces_wanted(geom, x.shape[1], w.shape[3]) and ctx.needs_input_grad[1]
        xp = to_pieces(x, s) if share else None   # as ModConv2dFn
        xcm = colmax_buffer(x, share)
        y = conv2d_raw(x, w, geom, out_hw, w.shape[3], in_scale=s, out_scale=d, bias=b, act=(act_idx, alpha, gain), noise=noise, strength=strength, x_pieces=xp, colmax=xcm)
        ctx.save_for_backward(x, w, s, d, b, noise, strength, y)
        ctx.xp, ctx.xcm = xp, xcm
        ctx.geom, ctx.out_hw, ctx.cfg = geom, out_hw, (act_idx, alpha, gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():
            raise NotImplementedError('fused synthesis layer: second-order gradients go through ModConv2dFn + BiasActNoiseFn (hip_ops.second_order())')
        x, w, s, d, b, noise, strength, y = ctx.saved_tensors
        act_idx, alpha, gain = ctx.cfg
        geom = ctx.geom
        need_x, need_w, need_s, need_d, need_b = [_needed(ctx, i) for i in range(5)]
        need_st = noise is not None and _needed(ctx, 6)
        dxp, db, dst, dd = bias_act_noise_bwd_dd_raw(dy, y, noise, strength, b, d, act_idx, alpha, gain)
        dx = dw = ds = None
        in_hw = (x.shape[2], x.shape[3])
        share = pieces_wanted(geom, x.shape[1], y.shape[1]) and (need_x or need_s) and need_w
        dyp = to_pieces(dxp, d) if share else None
        dycm = colmax_buffer(dxp, share)
        if need_x or need_s:
            dxs = conv2d_raw(dxp, w, dgrad_geom(geom), in_hw, w.shape[2], w_transposed=True, in_scale=d, x_pieces=dyp, colmax=dycm)
            ds, dx = scale_dot_raw(x, dxs, s, want_scaled=need_x)
            if not need_s:
                ds = None
        if need_w:
            dw = conv2d_wgrad_raw(x, dxp, geom, in_scale=s, out_scale=d, x_pieces=ctx.xp, dy_pieces=dyp, x_colmax=ctx.xcm, dy_colmax=dycm)
        return dx, dw, ds, (dd if need_d else None), (db if need_b else None), None, (dst if need_st else None), None, None, None, None, None


def modconv_ban_fusable(x, w, d, b, noise, act_idx):
    """The fused layer applies on the first-order path, with demodulation, a piecewise-linear invertible activation, MFMA-sized
    channel counts and 16 B paths; everything else takes the two-Function form."""
    return (_MODCONV_BAN and _second_order_depth == 0 and not _is_meta(x) and x.is_cuda and d is not None and b is not None
            and act_idx in (1, 3) and x.shape[1] % 32 == 0 and w.shape[3] % 32 == 0 and x.shape[1] >= 32 and 32 <= w.shape[3] <= 1024)    # <= 1024: the backward's noise-strength partial is one block column wide (igan_bias_act_noise_bwd_dd)


_MODCONV_BAN = os.environ.get('IGAN_MODCONV_BAN', '1') != '0'      # A/B switch


def modconv_composite(x, w, s, d, geom, out_hw):
    y = Conv2dFn.apply(x * s[:, :, None, None], w, geom, out_hw)
    if d is not None:
        y = y * d[:, :, None, None]
    return y


# ----------------------------------------------------------------------------
# style path of modulated_conv2d_layer: s = A(w_lat) + b + 1, d = rsqrt(s^2 . sum_taps(w^2) + 1e-8)

def _row_major(t):
    """[M,K] with unit inner stride and a 4-float-aligned row stride (a dlatents[:, i] slice qualifies)."""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.stride(0) < t.shape[1] or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t


def dense_small_raw(x, w, w_transposed=False, alpha=1.0, prologue=_abi.DENSE_PRO_NONE, x2=None, pro_scale=0.0,
                    epilogue=_abi.DENSE_EPI_SCALE, bias=None, bias_scale=1.0, add_const=0.0, eps=0.0, e1=None, e2=None,
                    want_colsum=False):
    """include/igan_hip.h igan_dense_small: y = epi(alpha * pro(x) . W), M <= 64 rows.  Returns y or (y, colsum)."""
    lib = _abi.get_plugin()
    _require_cuda_f32(x, w, x2, bias, e1, e2)
    x = _row_major(x)
    w = w.contiguous()
    m, k = x.shape
    n = w.shape[0] if w_transposed else w.shape[1]
    y = torch.empty((m, n), device=x.device, dtype=torch.float32)
    colsum = torch.empty((n,), device=x.device, dtype=torch.float32) if want_colsum else None
    x2 = x2.contiguous() if x2 is not None else None
    e1 = e1.contiguous() if e1 is not None else None
    e2 = e2.contiguous() if e2 is not None else None
    bias = bias.contiguous() if bias is not None else None
    p = _abi.DenseParams(x=x.data_ptr(), x2=_ptr(x2), w=w.data_ptr(), y=y.data_ptr(), bias=_ptr(bias), e1=_ptr(e1), e2=_ptr(e2),
                         colsum=_ptr(colsum), ldx=x.stride(0), ldy=n, M=m, K=k, N=n, w_transposed=1 if w_transposed else 0,
                         prologue=prologue, epilogue=epilogue, alpha=float(alpha), pro_scale=float(pro_scale),
                         bias_scale=float(bias_scale), add_const=float(add_const), eps=float(eps))
    _abi.check(lib.igan_dense_small(_stream(), ctypes.byref(p)))
    return (y, colsum) if want_colsum else y


def dense_small_wgrad_raw(a, b, alpha=1.0, pro_a=_abi.DENSE_PRO_NONE, pro_b=_abi.DENSE_PRO_NONE, b2=None, pro_scale=0.0):
    """dw[K,N] = alpha * pro_a(a)^T . pro_b(b)   (igan_dense_small_wgrad)."""
    lib = _abi.get_plugin()
    _require_cuda_f32(a, b, b2)
    a = _row_major(a)
    b = b.contiguous()
    b2 = b2.contiguous() if b2 is not None else None
    m, k = a.shape
    n = b.shape[1]
    dw = torch.empty((k, n), device=a.device, dtype=torch.float32)
    p = _abi.DenseWgradParams(a=a.data_ptr(), b=b.data_ptr(), b2=_ptr(b2), dw=dw.data_ptr(), lda=a.stride(0), M=m, K=k, N=n,
                              pro_a=pro_a, pro_b=pro_b, alpha=float(alpha), pro_scale=float(pro_scale))
    _abi.check(lib.igan_dense_small_wgrad(_stream(), ctypes.byref(p)))
    return dw


def sumsq_taps_raw(w):
    """[KH,KW,Cin,Cout] -> [Cin,Cout] sum over the taps of w^2."""
    lib = _abi.get_plugin()
    _require_cuda_f32(w)
    w = w.contiguous()
    out = torch.empty(tuple(w.shape[2:]), device=w.device, dtype=torch.float32)
    _abi.check(lib.igan_sumsq_taps(_stream(), _ptr(w), _ptr(out), w.shape[0] * w.shape[1], out.numel()))
    return out


def bcast_mul_taps_raw(w, v, scale):
    """scale * w[KH,KW,Cin,Cout] * v[Cin,Cout] (broadcast over the taps)."""
    lib = _abi.get_plugin()
    _require_cuda_f32(w, v)
    w = w.contiguous()
    v = v.contiguous()
    out = torch.empty_like(w)
    _abi.check(lib.igan_bcast_mul_taps(_stream(), _ptr(w), _ptr(v), _ptr(out), w.shape[0] * w.shape[1], v.numel(), float(scale)))
    return out


class SumsqTapsFn(torch.autograd.Function):
    """wsq[Cin, Cout] = sum over the taps of w^2 (the weights-only factor of the demodulation, networks_stylegan2.py:105-107) as ONE kernel with ONE kernel
    for its gradient (2 w g, spread over the taps) -- as `(w * w).sum((0, 1))` the second-order path paid two passes over every [3, 3, Cin, Cout] filter
    forward and three more backward, per layer and path-length step (the largest torch kernels of that step).  The gradient is linear in g and w: when the
    backward is itself being recorded (create_graph) it is taken through differentiable torch operations."""

    @staticmethod
    def forward(ctx, w):
        ctx.save_for_backward(w)
        return sumsq_taps_raw(w)

    @staticmethod
    def backward(ctx, g):
        w, = ctx.saved_tensors
        if torch.is_grad_enabled():
            return 2.0 * w * g[None, None]
        return bcast_mul_taps_raw(w, g.contiguous(), 2.0)


_SUMSQ_FN = os.environ.get('IGAN_SUMSQ_FN', '1') != '0'      # A/B switch: 0 = the torch composite (w * w).sum((0, 1))


def _sumsq_taps(w):
    if _SUMSQ_FN and w.is_cuda and w.dim() == 4 and (w.shape[2] * w.shape[3]) % 4 == 0 and not (w.data_ptr() & 15):
        return SumsqTapsFn.apply(w)
    return (w * w).sum(dim=(0, 1))


def style_mod_composite(y, a_w, a_b, w, c_a, c_w, demodulate, b1=None):
    """The same (s, d) from differentiable pieces (any order of differentiation; any sizes).  b1: a_b + 1 when the caller has it (style_bias_plus_one)."""
    s = matmul(y, a_w, alpha=c_a) + (b1 if b1 is not None else a_b + 1.0)
    d = None
    if demodulate:
        d = torch.rsqrt(matmul(s * s, _sumsq_taps(w), alpha=c_w * c_w) + 1e-8)
    return s, d


def style_bias_plus_one(biases):
    """[a_b + 1 for every layer] from ONE concatenation and ONE add (views of the result): the second-order path's per-layer `+ a_b + 1` pairs."""
    flat = torch.cat([b.reshape(-1) for b in biases]) + 1.0
    return list(flat.split([b.numel() for b in biases]))


_STYLE_FUSION = os.environ.get('IGAN_STYLE_FUSION', '1') != '0'      # A/B switch for profiling
_second_order_depth = 0


class second_order:
    """Context: the forward pass inside will be differentiated twice (path-length regulariser, loss.py:60-66).
    Functions whose fused backward is first-order only take their differentiable composite form directly,
    instead of recomputing it inside a create_graph backward."""

    def __enter__(self):
        global _second_order_depth
        _second_order_depth += 1

    def __exit__(self, *exc):
        global _second_order_depth
        _second_order_depth -= 1


def style_mod_fusable(y, a_w, w, demodulate):
    cin = a_w.shape[1]
    ok = _STYLE_FUSION and y.is_cuda and y.dim() == 2 and y.shape[0] <= 64 and y.shape[1] % 4 == 0 and cin % 4 == 0
    if demodulate:
        ok = ok and w.shape[3] % 4 == 0
    return ok


class StyleModFn(torch.autograd.Function):
    """(s, d) of modulated_conv2d_layer (networks_stylegan2.py:99-107) for a batch of at most 32 latents:
        s = c_a * y . A + b + 1                      [N, Cin]
        d = rsqrt(c_w^2 * (s^2 . wsq) + 1e-8)        [N, Cout],  wsq = sum_taps w^2 (passed in: cached per training op)
    two launches forward; backward five (style gradient + bias gradient, d wsq, its spread onto the filter taps,
    d latents, d A) instead of ~40 element-wise / GEMM launches.  When the backward has to be differentiated
    itself (create_graph: path-length regulariser) it is taken through `style_mod_composite`."""

    @staticmethod
    def forward(ctx, y, a_w, a_b, w, wsq, c_a, c_w):
        s = dense_small_raw(y, a_w, alpha=c_a, epilogue=_abi.DENSE_EPI_BIAS, bias=a_b, bias_scale=1.0, add_const=1.0)
        d = None
        if wsq is not None:
            d = dense_small_raw(s, wsq, alpha=c_w * c_w, prologue=_abi.DENSE_PRO_SQUARE, epilogue=_abi.DENSE_EPI_RSQRT, eps=1e-8)
        ctx.save_for_backward(y, a_w, a_b, w, wsq, s, d)
        ctx.c_a, ctx.c_w = c_a, c_w
        if d is None:
            return s
        return s, d

    @staticmethod
    def backward(ctx, gs, gd=None):
        y, a_w, a_b, w, wsq, s, d = ctx.saved_tensors
        c_a, c_w = ctx.c_a, ctx.c_w
        need = [_needed(ctx, i) for i in range(4)]
        if torch.is_grad_enabled():
            with torch.enable_grad():
                sc, dc = style_mod_composite(y, a_w, a_b, w, c_a, c_w, d is not None)
                outs, gouts = [sc], [gs if gs is not None else torch.zeros_like(sc)]
                if dc is not None and gd is not None:
                    outs.append(dc); gouts.append(gd)
                inputs = [t for t, nd in zip((y, a_w, a_b, w), need) if nd]
                grads = list(torch.autograd.grad(outs, inputs, gouts, create_graph=True, allow_unused=True)) if inputs else []
            res = [grads.pop(0) if nd else None for nd in need]
            return res[0], res[1], res[2], res[3], None, None, None
        dy = da_w = da_b = dw = None
        if d is not None and gd is not None:
            ps = -0.5 * c_w * c_w
            ds, da_b = dense_small_raw(gd, wsq, w_transposed=True, prologue=_abi.DENSE_PRO_DEMOD_GRAD, x2=d, pro_scale=ps,
                                       epilogue=_abi.DENSE_EPI_STYLE_GRAD, e1=gs, e2=s, bias_scale=1.0, want_colsum=True)
            if need[3]:
                dwsq = dense_small_wgrad_raw(s, gd, pro_a=_abi.DENSE_PRO_SQUARE, pro_b=_abi.DENSE_PRO_DEMOD_GRAD, b2=d, pro_scale=ps)
                dw = bcast_mul_taps_raw(w, dwsq, 2.0)
        else:
            ds = gs
            da_b = gs.sum(dim=0) if need[2] else None
        if not need[2]:
            da_b = None
        if need[0]:
            dy = dense_small_raw(ds, a_w, w_transposed=True, alpha=c_a)
        if need[1]:
            da_w = dense_small_wgrad_raw(y, ds, alpha=c_a)
        return dy, da_w, da_b, dw, None, None, None


def _dense_group(x, w, y, w_transposed=False, alpha=1.0, prologue=_abi.DENSE_PRO_NONE, x2=None, pro_scale=0.0,
                 epilogue=_abi.DENSE_EPI_SCALE, bias=None, bias_scale=1.0, add_const=0.0, eps=0.0, e1=None, e2=None, colsum=None):
    m, k = x.shape
    n = y.shape[1]
    return _abi.DenseParams(x=x.data_ptr(), x2=_ptr(x2), w=w.data_ptr(), y=y.data_ptr(), bias=_ptr(bias), e1=_ptr(e1), e2=_ptr(e2),
                            colsum=_ptr(colsum), ldx=x.stride(0), ldy=n, M=m, K=k, N=n, w_transposed=1 if w_transposed else 0,
                            prologue=prologue, epilogue=epilogue, alpha=float(alpha), pro_scale=float(pro_scale),
                            bias_scale=float(bias_scale), add_const=float(add_const), eps=float(eps))


def dense_small_grouped_raw(groups):
    """One launch for a list of igan_dense_params (include/igan_hip.h igan_dense_small_grouped)."""
    lib = _abi.get_plugin()
    arr = (_abi.DenseParams * len(groups))(*groups)
    _abi.check(lib.igan_dense_small_grouped(_stream(), arr, len(groups)))


def _wgrad_group(a, b, dw, alpha=1.0, pro_a=_abi.DENSE_PRO_NONE, pro_b=_abi.DENSE_PRO_NONE, b2=None, pro_scale=0.0):
    m, k = a.shape
    return _abi.DenseWgradParams(a=a.data_ptr(), b=b.data_ptr(), b2=_ptr(b2), dw=dw.data_ptr(), lda=a.stride(0), M=m, K=k, N=b.shape[1],
                                 pro_a=pro_a, pro_b=pro_b, alpha=float(alpha), pro_scale=float(pro_scale))


def dense_small_wgrad_grouped_raw(groups):
    lib = _abi.get_plugin()
    arr = (_abi.DenseWgradParams * len(groups))(*groups)
    _abi.check(lib.igan_dense_small_wgrad_grouped(_stream(), arr, len(groups)))


def sumsq_taps_grouped_raw(ws):
    """[KH,KW,Cin,Cout] -> [Cin,Cout] for a list of filters, one launch."""
    lib = _abi.get_plugin()
    ws = [w.contiguous() for w in ws]
    outs = [torch.empty(tuple(w.shape[2:]), device=w.device, dtype=torch.float32) for w in ws]
    arr = (_abi.TapsParams * len(ws))(*[_abi.TapsParams(w=w.data_ptr(), v=None, out=o.data_ptr(), taps=w.shape[0] * w.shape[1], n=o.numel(), scale=1.0)
                                         for w, o in zip(ws, outs)])
    _abi.check(lib.igan_sumsq_taps_grouped(_stream(), arr, len(ws)))
    return outs


def bcast_mul_taps_grouped_raw(ws, vs, scale):
    lib = _abi.get_plugin()
    ws = [w.contiguous() for w in ws]
    outs = [torch.empty_like(w) for w in ws]
    arr = (_abi.TapsParams * len(ws))(*[_abi.TapsParams(w=w.data_ptr(), v=v.data_ptr(), out=o.data_ptr(), taps=w.shape[0] * w.shape[1], n=v.numel(), scale=float(scale))
                                         for w, v, o in zip(ws, vs, outs)])
    _abi.check(lib.igan_bcast_mul_taps_grouped(_stream(), arr, len(ws)))
    return outs


class StyleModAllFn(torch.autograd.Function):
    """StyleModFn for ALL modulated layers of one synthesis pass at once (they depend on the latents and the weights only,
    not on the activations): forward = one grouped launch for the style affines + one for the demodulations, backward = five
    grouped launches, instead of 2 and 5 per layer.  `cfg` is a list of (c_a, c_w, demodulate) per layer; the tensor
    arguments are, per layer, (y, a_w, a_b, w, wsq) with w / wsq None where there is no demodulation.  Returns the
    styles of all layers followed by the demodulation coefficients of the demodulated ones."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        _mark_inputs(ctx, cfg, *tensors)
        L = len(cfg)
        layers = [tensors[5 * i:5 * i + 5] for i in range(L)]
        ys = [_row_major(t[0]) for t in layers]
        dev = ys[0].device
        ss = [torch.empty((ys[i].shape[0], layers[i][1].shape[1]), device=dev, dtype=torch.float32) for i in range(L)]
        a_ws = [t[1].contiguous() for t in layers]
        a_bs = [t[2].contiguous() for t in layers]
        dense_small_grouped_raw([_dense_group(ys[i], a_ws[i], ss[i], alpha=cfg[i][0], epilogue=_abi.DENSE_EPI_BIAS, bias=a_bs[i], add_const=1.0)
                                 for i in range(L)])
        dem = [i for i in range(L) if cfg[i][2]]
        ds = {i: torch.empty((ys[i].shape[0], layers[i][3].shape[3]), device=dev, dtype=torch.float32) for i in dem}
        if dem:
            dense_small_grouped_raw([_dense_group(ss[i], layers[i][4], ds[i], alpha=cfg[i][1] ** 2, prologue=_abi.DENSE_PRO_SQUARE,
                                                  epilogue=_abi.DENSE_EPI_RSQRT, eps=1e-8) for i in dem])
        ctx.cfg, ctx.dem = cfg, dem
        ctx.save_for_backward(*tensors, *ss, *[ds[i] for i in dem])
        return tuple(ss) + tuple(ds[i] for i in dem)

    @staticmethod
    def backward(ctx, *grads):
        cfg, dem = ctx.cfg, ctx.dem
        L = len(cfg)
        saved = ctx.saved_tensors
        layers = [saved[5 * i:5 * i + 5] for i in range(L)]
        ss = saved[5 * L:6 * L]
        dmap = {i: saved[6 * L + j] for j, i in enumerate(dem)}
        gss = list(grads[:L])
        gds = {i: grads[L + j] for j, i in enumerate(dem)}
        out = [None] * (5 * L)
        if torch.is_grad_enabled():
            # differentiable per-layer composite (second-order gradients)
            for i in range(L):
                y, a_w, a_b, w, _ = layers[i]
                need = [_needed(ctx, 1 + 5 * i + j) for j in range(4)]
                need[3] = need[3] and cfg[i][2]
                if not any(need) or (gss[i] is None and gds.get(i) is None):
                    continue
                with torch.enable_grad():
                    sc, dc = style_mod_composite(y, a_w, a_b, w, cfg[i][0], cfg[i][1], cfg[i][2])
                    outs, gouts = [sc], [gss[i] if gss[i] is not None else torch.zeros_like(sc)]
                    if dc is not None and gds.get(i) is not None:
                        outs.append(dc); gouts.append(gds[i])
                    ins = [t for t, nd in zip((y, a_w, a_b, w), need) if nd]
                    gr = list(torch.autograd.grad(outs, ins, gouts, create_graph=True, allow_unused=True))
                for j in range(4):
                    if need[j]:
                        out[5 * i + j] = gr.pop(0)
            return (None, *out)
        dev = ss[0].device
        m = ss[0].shape[0]
        gss = [g.contiguous() if g is not None else torch.zeros_like(ss[i]) for i, g in enumerate(gss)]
        # 1. style gradients through the demodulation (+ bias gradients as column sums)
        dss = list(gss)
        dbs = [None] * L
        act = [i for i in dem if gds.get(i) is not None]
        if act:
            groups = []
            for i in act:
                dss[i] = torch.empty_like(ss[i])
                dbs[i] = torch.empty((ss[i].shape[1],), device=dev, dtype=torch.float32)
                groups.append(_dense_group(gds[i].contiguous(), layers[i][4], dss[i], w_transposed=True, prologue=_abi.DENSE_PRO_DEMOD_GRAD,
                                           x2=dmap[i], pro_scale=-0.5 * cfg[i][1] ** 2, epilogue=_abi.DENSE_EPI_STYLE_GRAD, e1=gss[i], e2=ss[i],
                                           bias_scale=1.0, colsum=dbs[i]))
            dense_small_grouped_raw(groups)
            # 2./3. d wsq and its spread onto the filter taps
            dwsqs = [torch.empty_like(layers[i][4]) for i in act]
            dense_small_wgrad_grouped_raw([_wgrad_group(ss[i], gds[i].contiguous(), dwsqs[j], pro_a=_abi.DENSE_PRO_SQUARE, pro_b=_abi.DENSE_PRO_DEMOD_GRAD,
                                                        b2=dmap[i], pro_scale=-0.5 * cfg[i][1] ** 2) for j, i in enumerate(act)])
            dws = bcast_mul_taps_grouped_raw([layers[i][3] for i in act], dwsqs, 2.0)
            for j, i in enumerate(act):
                out[5 * i + 3] = dws[j]
        for i in range(L):
            if dbs[i] is None:
                dbs[i] = dss[i].sum(dim=0)
            out[5 * i + 2] = dbs[i]
        # 4. latent gradients, 5. affine weight gradients
        ys = [_row_major(layers[i][0]) for i in range(L)]
        dys = [torch.empty((m, ys[i].shape[1]), device=dev, dtype=torch.float32) for i in range(L)]
        a_ws = [layers[i][1].contiguous() for i in range(L)]
        dense_small_grouped_raw([_dense_group(dss[i], a_ws[i], dys[i], w_transposed=True, alpha=cfg[i][0]) for i in range(L)])
        das = [torch.empty_like(a_ws[i]) for i in range(L)]
        dense_small_wgrad_grouped_raw([_wgrad_group(ys[i], dss[i], das[i], alpha=cfg[i][0]) for i in range(L)])
        for i in range(L):
            out[5 * i + 0] = dys[i]
            out[5 * i + 1] = das[i]
        return (None, *out)


_STYLE_GROUPED = os.environ.get('IGAN_STYLE_GROUPED', '1') != '0'      # A/B switch


def style_mod_all(layers):
    """layers: list of dicts (y, a_w, a_b, w, wsq, c_a, c_w, demodulate).  Returns a list of (s, d) or None when the grouped
    kernels do not apply (sizes, second-order context, more than IGAN_DENSE_MAX_GROUPS layers)."""
    if not _STYLE_GROUPED or _second_order_depth > 0 or not layers or len(layers) > _abi.DENSE_MAX_GROUPS:
        return None
    for l in layers:
        if _is_meta(l['y']) or not style_mod_fusable(l['y'], l['a_w'], l['w'], l['demodulate']):
            return None
    cfg = [(l['c_a'], l['c_w'], bool(l['demodulate'])) for l in layers]
    flat = []
    for l in layers:
        flat += [l['y'], l['a_w'], l['a_b'], l['w'] if l['demodulate'] else None, l['wsq'] if l['demodulate'] else None]
    res = StyleModAllFn.apply(cfg, *flat)
    L = len(layers)
    dem = [i for i in range(L) if cfg[i][2]]
    dmap = {i: res[L + j] for j, i in enumerate(dem)}
    return [(res[i], dmap.get(i)) for i in range(L)]


def style_mod(y, a_w, a_b, w, wsq, c_a, c_w, demodulate):
    """Dispatch: fused kernels when the sizes allow, the differentiable composite otherwise."""
    if _is_meta(y):
        s = torch.empty((y.shape[0], a_w.shape[1]), device='meta')
        return s, (torch.empty((y.shape[0], w.shape[3]), device='meta') if demodulate else None)
    if _second_order_depth > 0 or not style_mod_fusable(y, a_w, w, demodulate):
        return style_mod_composite(y, a_w, a_b, w, c_a, c_w, demodulate)
    if demodulate:
        return StyleModFn.apply(y, a_w, a_b, w, wsq, c_a, c_w)
    return StyleModFn.apply(y, a_w, a_b, w, None, c_a, c_w), None


# ----------------------------------------------------------------------------
# LPIPS per-layer distance (fused normalise / diff / lin / spatial sum)

class LpipsLayerFn(torch.autograd.Function):
    """sum_{h,w} sum_c lin_c (fa_c/|fa| - fb_c/|fb|)^2 per sample, on raw VGG features (channels_last)."""

    @staticmethod
    def forward(ctx, fa, fb, lin):
        if _is_meta(fa):
            return torch.empty((fa.shape[0],), device='meta')
        lib = _abi.get_plugin()
        _require_cuda_f32(fa, fb, lin)
        fa = nhwc(fa)
        fb = nhwc(fb)
        lin = lin.contiguous()
        n, c, h, w = fa.shape
        blocks = lib.igan_lpips_layer_blocks(n, h * w)
        partial = torch.empty((n, blocks), device=fa.device, dtype=torch.float32)
        _abi.check(lib.igan_lpips_layer_fwd(_stream(), _ptr(fa), _ptr(fb), _ptr(lin), _ptr(partial), n, h * w, c))
        ctx.save_for_backward(fa, fb, lin)
        return partial.sum(dim=1)

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():
            raise NotImplementedError('LPIPS distance: second-order gradients are not built (the reconstruction term is first-order)')
        fa, fb, lin = ctx.saved_tensors
        lib = _abi.get_plugin()
        n, c, h, w = fa.shape
        g = g.contiguous()
        outs = [None, None]
        for i, (p, q) in enumerate(((fa, fb), (fb, fa))):     # the distance is symmetric in (fa, fb)
            if _needed(ctx, i):
                d = torch.empty_like(p)
                _abi.check(lib.igan_lpips_layer_bwd(_stream(), _ptr(p), _ptr(q), _ptr(lin), _ptr(g), _ptr(d), n, h * w, c))
                outs[i] = d
        return outs[0], outs[1], None


_pair_tables = {}


def _lpips_pair_tables(n, device):
    """int32 sample tables of the G loss's four LPIPS distances (see LpipsPairsFn), created once per (n, device)."""
    key = (n, str(device))
    if key not in _pair_tables:
        r = np.arange(n)
        ia = np.concatenate([np.arange(3 * n), 2 * n + r]).astype(np.int32)
        ib = np.concatenate([np.arange(2 * n), n + r, r]).astype(np.int32)
        _pair_tables[key] = (torch.from_numpy(ia).to(device), torch.from_numpy(ib).to(device))
    return _pair_tables[key]


class LpipsPairsFn(torch.autograd.Function):
    """All LPIPS distances of the G loss (training/loss.py:31,41) in one Function, over every feature layer:
        rows [0, 2n)  : d(f_gen[i], f_real[i])            (rec_1 vs real_1, rec_2 vs real_2)
        rows [2n, 3n) : d(f_gen[2n + i], f_real[n + i])   (interp vs real_2)
        rows [3n, 4n) : d(f_gen[2n + i], f_real[i])       (interp vs real_1)
    f_gen: per layer [3n, C, H, W] (rec_1, rec_2, interp stacked on the batch axis), f_real: per layer [2n, C, H, W];
    lins: per layer [C] = |lin| / (C * H * W) (the spatial mean folded into the weights: H * W is a power of two, so the
    values are those of dividing afterwards).  One launch per layer (igan_lpips_pairs_fwd: sample-pair tables), the layers'
    block sums side by side in one [4n, width] array and ONE row sum; no slices, no concatenation.  Backward: per layer one
    launch for rows [0, 3n) writing the whole gradient w.r.t. f_gen and one for rows [3n, 4n) adding the interpolated
    images' second contribution.   apply(*lins, *f_gen, *f_real, n, num_layers) -> [4n].  (Tensors first: _needed() indexes
    ctx.next_functions, which counts tensor inputs only.)"""

    @staticmethod
    def forward(ctx, *args):
        n, L = args[-2], args[-1]
        lins, fg, fr = args[:L], args[L:2 * L], args[2 * L:3 * L]
        lib = _abi.get_plugin()
        fg = [nhwc(t) for t in fg]
        fr = [nhwc(t) for t in fr]
        lins = [t.contiguous() for t in lins]
        _require_cuda_f32(*fg, *fr, *lins)
        ia, ib = _lpips_pair_tables(n, fg[0].device)
        geo = []
        for a, b in zip(fg, fr):
            assert a.shape[0] == 3 * n and b.shape[0] == 2 * n and a.shape[1:] == b.shape[1:]
            c, hw = a.shape[1], a.shape[2] * a.shape[3]
            geo.append((c, hw, lib.igan_lpips_layer_blocks(4 * n, hw)))
        width = sum(g[2] for g in geo)
        partial = torch.empty((4 * n, width), device=fg[0].device, dtype=torch.float32)
        col = 0
        st = _stream()
        for a, b, lin, (c, hw, blocks) in zip(fg, fr, lins, geo):
            _abi.check(lib.igan_lpips_pairs_fwd(st, _ptr(a), _ptr(b), _ptr(lin), _ptr(ia), _ptr(ib), partial.data_ptr() + 4 * col,
                                                width, blocks, 4 * n, hw, c))
            col += blocks
        ctx.save_for_backward(*lins, *fg, *fr)
        ctx.n, ctx.L, ctx.geo = n, L, geo
        return partial.sum(dim=1)

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():
            raise NotImplementedError('LPIPS distance: second-order gradients are not built (the reconstruction term is first-order)')
        n, L = ctx.n, ctx.L
        t = ctx.saved_tensors
        lins, fg, fr = t[:L], t[L:2 * L], t[2 * L:3 * L]
        lib = _abi.get_plugin()
        g = g.contiguous()
        ia, ib = _lpips_pair_tables(n, g.device)
        st = _stream()
        outs = []
        for i, (a, b, lin, (c, hw, _)) in enumerate(zip(fg, fr, lins, ctx.geo)):
            if not _needed(ctx, L + i):
                outs.append(None)
                continue
            d = torch.empty_like(a)
            _abi.check(lib.igan_lpips_pairs_bwd(st, _ptr(a), _ptr(b), _ptr(lin), _ptr(ia), _ptr(ib), _ptr(g), _ptr(d), 0, 3 * n, hw, c))
            _abi.check(lib.igan_lpips_pairs_bwd(st, _ptr(a), _ptr(b), _ptr(lin), ia.data_ptr() + 4 * 3 * n, ib.data_ptr() + 4 * 3 * n,
                                                g.data_ptr() + 4 * 3 * n, _ptr(d), 1, n, hw, c))
            outs.append(d)
        return (None,) * L + tuple(outs) + (None,) * L + (None, None)


class PoolTapFn(torch.autograd.Function):
    """x -> (x, maxpool2x2(x)): a VGG feature map that is both an LPIPS tap and the input of the next block.  One Function
    owns both consumers so that the backward is one pass, dx = d_tap + route(d_pooled) (igan_maxpool2x2_bwd), instead of a
    pooling gradient at full resolution followed by the autograd engine's add."""

    @staticmethod
    def forward(ctx, x):
        lib = _abi.get_plugin()
        _require_cuda_f32(x)
        x = nhwc(x)
        n, c, h, w = x.shape
        y = empty_nchw(n, c, h // 2, w // 2, x)
        _abi.check(lib.igan_maxpool2x2_fwd(_stream(), _ptr(x), _ptr(y), n, h, w, c))
        ctx.save_for_backward(x)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_tap, g_pool):
        if torch.is_grad_enabled():
            raise NotImplementedError('VGG max-pool: second-order gradients are not built (the reconstruction term is first-order)')
        x, = ctx.saved_tensors
        if g_pool is None:
            return g_tap
        lib = _abi.get_plugin()
        n, c, h, w = x.shape
        g_pool = nhwc(g_pool)
        skip = nhwc(g_tap) if g_tap is not None else None
        dx = torch.empty_like(x)
        _abi.check(lib.igan_maxpool2x2_bwd(_stream(), _ptr(x), _ptr(g_pool), _ptr(skip), _ptr(dx), n, h, w, c))
        return dx


# ----------------------------------------------------------------------------
# minibatch stddev

def mbstd_composite(x, group_size):
    """networks_stylegan2.py:132-144 with torch ops (used only to differentiate the backward)."""
    n, c, h, w = x.shape
    g = min(group_size, n)
    y = x.reshape(g, -1, 1, c, h, w)
    y = y - y.mean(dim=0, keepdim=True)
    y = (y * y).mean(dim=0)
    y = torch.sqrt(y + 1e-8)
    y = y.mean(dim=(2, 3, 4), keepdim=True)
    y = y.mean(dim=2)
    y = y.repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)


def mbstd_fwd_raw(x, g):
    if _is_meta(x):
        return torch.empty((x.shape[0], x.shape[1] + 1, x.shape[2], x.shape[3]), device='meta')
    lib = _abi.get_plugin()
    _require_cuda_f32(x)
    x = nhwc(x)
    n, c, h, w = x.shape
    y = empty_nchw(n, c + 1, h, w, x)
    stat = torch.empty((int(lib.igan_mbstd_workspace_floats(n, h, w, c, g)),), device=x.device, dtype=torch.float32)
    _abi.check(lib.igan_mbstd_fwd(_stream(), _ptr(x), _ptr(y), _ptr(stat), n, h, w, c, g))
    return y


def mbstd_bwd_raw(x, dy, g):
    lib = _abi.get_plugin()
    _require_cuda_f32(x, dy)
    x = nhwc(x)
    dy = nhwc(dy)
    n, c, h, w = x.shape
    dx = empty_nchw(n, c, h, w, x)
    _abi.check(lib.igan_mbstd_bwd(_stream(), _ptr(x), _ptr(dy), _ptr(dx), n, h, w, c, g))
    return dx


class MbStdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group_size):
        g = min(group_size, x.shape[0])
        ctx.save_for_backward(x)
        ctx.g = g
        return mbstd_fwd_raw(x, g)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            with torch.enable_grad():
                y2 = mbstd_composite(x, ctx.g)
                (dx,) = torch.autograd.grad(y2, [x], dy, create_graph=True)
            return dx, None
        return mbstd_bwd_raw(x, dy, ctx.g), None


# ----------------------------------------------------------------------------
# nearest neighbour, optimizer

def row_sqnorm_raw(a):
    lib = _abi.get_plugin()
    _require_cuda_f32(a)
    a = a.contiguous()
    out = torch.empty((a.shape[0],), device=a.device, dtype=torch.float32)
    _abi.check(lib.igan_row_sqnorm(_stream(), _ptr(a), _ptr(out), a.shape[0], a.shape[1]))
    return out


def nn1_state(nq, device):
    """Running minimum of igan_nn1_update for nq queries: (squared distance fp64 = +inf, candidate index int32 = INT32_MAX)."""
    return (torch.full((nq,), float('inf'), device=device, dtype=torch.float64),
            torch.full((nq,), 2 ** 31 - 1, device=device, dtype=torch.int32))


def nn1_update_raw(query, qnorm, cand, cnorm, best_d2, best_idx, idx_base):
    """Fold one candidate batch into the running exact minimum (best_d2 fp64 [nq], best_idx int32 [nq]; contiguous views)."""
    lib = _abi.get_plugin()
    _require_cuda_f32(query, qnorm, cand, cnorm)
    if best_d2.dtype != torch.float64 or best_idx.dtype != torch.int32 or not (best_d2.is_contiguous() and best_idx.is_contiguous()):
        raise TypeError('nn1_update: best_d2 must be contiguous float64 and best_idx contiguous int32')
    query = query.contiguous()
    cand = cand.contiguous()
    nq, dim = query.shape
    nc = cand.shape[0]
    dots = torch.empty((nq, nc), device=query.device, dtype=torch.float32)
    _abi.check(lib.igan_nn1_update(_stream(), _ptr(query), _ptr(qnorm), _ptr(cand), _ptr(cnorm), _ptr(best_d2), _ptr(best_idx), _ptr(dots),
                                   nq, nc, dim, idx_base))


def finite_check_raw(g, flag):
    lib = _abi.get_plugin()
    _abi.check(lib.igan_finite_check(_stream(), _ptr(g), g.numel(), _ptr(flag)))


def adam_step_raw(w, g, m, v, lr, beta1, beta2, eps, pow_state, skip_flag):
    lib = _abi.get_plugin()
    _abi.check(lib.igan_adam_step(_stream(), _ptr(w), _ptr(g), _ptr(m), _ptr(v), w.numel(),
                                  float(lr), float(beta1), float(beta2), float(eps), _ptr(pow_state), _ptr(skip_flag)))


def ema_raw(dst, src, beta):
    lib = _abi.get_plugin()
    _abi.check(lib.igan_ema(_stream(), _ptr(dst), _ptr(src), dst.numel(), float(beta)))
