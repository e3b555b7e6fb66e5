"""GPU parity tests of the HIP kernels (through the C ABI) against the CPU oracle.

Tolerances (fp32 kernels vs an fp64 oracle evaluation of the same fp32 inputs), as relative
max-abs error  max|hip - oracle| / max|oracle|:
    streaming kernels (upfirdn2d, fused_bias_act, mbstd, Adam, EMA): <= 2e-6 .. 1e-5
    MFMA convolutions (K up to ~4600 fp32 FMAs per output):           <= 3e-5
"""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_err, to_nhwc_cuda

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- upfirdn2d
UPFIRDN_CASES = [
    # (major, H, W, C, kshape, up, down, pad0, pad1)   call sites of config-e-Gskip-Dresnet + edge cases
    (2, 17, 17, 8, 4, 1, 1, 1, 1),      # G Conv0_up post-filter
    (2, 16, 16, 8, 4, 1, 1, 2, 2),      # D Conv1_down pre-filter
    (2, 16, 16, 8, 4, 1, 1, 1, 1),      # D Skip pre-filter
    (2, 8, 8, 3, 4, 2, 1, 2, 1),        # G RGB upsample_2d
    (2, 16, 16, 3, 4, 1, 2, 1, 2),      # its gradient-like downsample
    (1, 5, 7, 4, 3, 1, 1, 1, 1),        # 3x3 taps, ragged size (fast path, zero-extended taps)
    (3, 9, 6, 5, 4, 1, 1, 0, 3),        # C % 4 != 0 -> generic path
    (1, 6, 6, 4, 2, 2, 1, 1, 0),        # 2-tap upsample
    (1, 12, 10, 4, 4, 1, 1, -1, 2),     # negative pad = crop
    (2, 4, 4, 12, 4, 1, 1, 2, 2),       # outH < 8 -> small-strip variant
    (1, 33, 31, 16, 4, 1, 1, 1, 1),     # strip tail (outH % 8 != 0)
    (1, 7, 7, 4, 1, 1, 1, 0, 0),        # 1x1 filter
    (1, 9, 9, 2, 6, 3, 2, 2, 3),        # up 3 / down 2 generic
]


@pytest.mark.parametrize('case', UPFIRDN_CASES)
def test_upfirdn2d_forward_and_grads(case, cuda_device):
    from oracle import upfirdn_2d as O
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    major, H, W, C, ks, up, down, p0, p1 = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = torch.from_numpy(rng.randn(major, H, W, C))
    k = rng.rand(ks, ks).astype(np.float32) + 0.1
    kw = dict(upx=up, upy=up, downx=down, downy=down, padx0=p0, padx1=p1, pady0=p0, pady1=p1)
    xo = x.clone().requires_grad_(True)
    yo = O.upfirdn_2d_ref(xo, k, **kw)
    xg = x.to(torch.float32).to(cuda_device).requires_grad_(True)
    yg = upfirdn_2d(xg, k, **kw)
    assert tuple(yg.shape) == tuple(yo.shape)
    assert rel_err(yg, yo) < 2e-6
    # first and second order gradients (the op is linear: d/dx <dy, y> = op^T dy)
    dy = torch.from_numpy(rng.randn(*yo.shape))
    (gxo,) = torch.autograd.grad(yo, xo, dy, create_graph=True)
    (gxg,) = torch.autograd.grad(yg, xg, dy.to(torch.float32).to(cuda_device), create_graph=True)
    assert rel_err(gxg, gxo) < 2e-6
    # differentiate the gradient w.r.t. dy direction: pass a second cotangent through the backward op
    dyo = dy.clone().requires_grad_(True)
    dyg = dy.to(torch.float32).to(cuda_device).requires_grad_(True)
    (g2o,) = torch.autograd.grad(O.upfirdn_2d_ref(xo, k, **kw), xo, dyo, create_graph=True)
    (g2g,) = torch.autograd.grad(upfirdn_2d(xg, k, **kw), xg, dyg, create_graph=True)
    v = torch.from_numpy(rng.randn(*x.shape))
    (ddo,) = torch.autograd.grad(g2o, dyo, v)
    (ddg,) = torch.autograd.grad(g2g, dyg, v.to(torch.float32).to(cuda_device))
    assert rel_err(ddg, ddo) < 2e-6


def test_upfirdn2d_matches_loop_oracle_small(cuda_device):
    from oracle import upfirdn_2d as O
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    rng = np.random.RandomState(5)
    x = rng.randn(2, 6, 5, 4).astype(np.float32)
    k = rng.rand(4, 4).astype(np.float32)
    for kw in [dict(padx0=1, padx1=1, pady0=1, pady1=1), dict(upx=2, upy=2, padx0=2, padx1=1, pady0=2, pady1=1),
               dict(downx=2, downy=2, padx0=1, padx1=1, pady0=1, pady1=1)]:
        ref = O.upfirdn_2d_loops(x.astype(np.float64), k, **kw)
        got = upfirdn_2d(torch.from_numpy(x).to(cuda_device), k, **kw)
        assert rel_err(got, ref) < 2e-6


def test_upfirdn2d_argument_errors(cuda_device):
    from inclusivegan_amd import hip_ops
    x = torch.zeros(1, 4, 4, 4, device=cuda_device)
    k = np.ones((4, 4), np.float32)
    with pytest.raises(ValueError):
        hip_ops.upfirdn2d_raw(x, k, 1, 1, 1, 1, -10, 0, 0, 0)   # output must be at least 1x1
    with pytest.raises(ValueError):
        hip_ops.upfirdn2d_raw(x, np.ones((4,), np.float32), 1, 1, 1, 1, 0, 0, 0, 0)   # kernel must have rank 2


# ----------------------------------------------------------------------------- fused_bias_act
ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']


@pytest.mark.parametrize('act', ACTS)
@pytest.mark.parametrize('layout', ['nhwc4d', 'nchw4d', '2d', 'odd'])
def test_fused_bias_act_forward_backward(act, layout, cuda_device):
    from oracle import fused_bias_act as O
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act
    rng = np.random.RandomState(ACTS.index(act) * 7 + len(layout))
    shape = {'nhwc4d': (3, 8, 5, 6), 'nchw4d': (3, 8, 5, 6), '2d': (5, 12), 'odd': (2, 3, 5, 7)}[layout]
    x = torch.from_numpy(rng.randn(*shape))
    b = torch.from_numpy(rng.randn(shape[1]))
    xo = x.clone().requires_grad_(True); bo = b.clone().requires_grad_(True)
    yo = O.fused_bias_act(xo, bo, act=act)
    xg = x.to(torch.float32).to(cuda_device)
    if layout == 'nhwc4d':
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg = xg.requires_grad_(True)
    bg = b.to(torch.float32).to(cuda_device).requires_grad_(True)
    yg = fused_bias_act(xg, bg, act=act)
    assert rel_err(yg, yo) < 1e-5
    dy = torch.from_numpy(rng.randn(*shape))
    gxo, gbo = torch.autograd.grad(yo, [xo, bo], dy)
    if act == 'swish':
        # Reference behaviour, not calculus: swish differentiates at ref = x, the op's *pre-bias* input
        # (fused_bias_act.py:29,133; fused_bias_act.cu:109), so with a bias the reference's gradient is
        # swish'(x), not swish'(x + b).  The product reproduces the reference.
        gxo = O.fused_bias_act_kernel_ref(dy, None, x, 1, 9, 0.0, float(np.sqrt(2)), 1).reshape(shape)
        gbo = gxo.sum(dim=[d for d in range(len(shape)) if d != 1])
    gxg, gbg = torch.autograd.grad(yg, [xg, bg], dy.to(torch.float32).to(cuda_device))
    assert rel_err(gxg, gxo) < 1e-5
    assert rel_err(gbg, gbo) < 1e-5


@pytest.mark.parametrize('act', ACTS)
def test_fused_bias_act_second_order(act, cuda_device):
    """Twice differentiable for all nine activations (fused_bias_act.py:149-189): an R1-style penalty on the gradient w.r.t. x, its
    gradient w.r.t. x, b and the first cotangent, against fp64 autograd of the oracle.  The smooth activations go through the grad = 2
    kernel (hip_ops.FusedBiasActSmoothFn: d(dx)/dx = kernel2(d_dx, ref) * dy).  swish runs without a bias here: with one, the
    reference evaluates swish' at the pre-bias input (see the first-order test), which is not the derivative autograd would take."""
    from oracle import fused_bias_act as O
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(4, 8, 6, 6)); b = torch.from_numpy(rng.randn(8)) if act != 'swish' else None
    w = torch.from_numpy(rng.randn(4, 8, 6, 6))
    def penalty(fba, x, b, w):
        y = fba(x, b, act=act)
        (gx,) = torch.autograd.grad((y * w).sum(), x, create_graph=True)   # R1-style: gradient norm penalty
        return (gx * gx).sum() + (y * y).sum()
    xo = x.clone().requires_grad_(True); bo = b.clone().requires_grad_(True) if b is not None else None
    wo = w.clone().requires_grad_(True)
    po = penalty(O.fused_bias_act, xo, bo, wo)
    gxo, gbo, gwo = torch.autograd.grad(po, [xo, bo, wo] if bo is not None else [xo, wo, wo], allow_unused=True)
    xg = x.float().to(cuda_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.float().to(cuda_device).requires_grad_(True) if b is not None else None
    wg = w.float().to(cuda_device).requires_grad_(True)
    pg = penalty(fused_bias_act, xg, bg, wg)
    gxg, gbg, gwg = torch.autograd.grad(pg, [xg, bg, wg] if bg is not None else [xg, wg, wg], allow_unused=True)
    tol = 1e-5 if act in ('linear', 'relu', 'lrelu') else 1e-4
    assert rel_err(pg, po) < tol
    assert rel_err(gxg, gxo) < tol
    assert rel_err(gwg, gwo) < tol
    if bo is not None and gbo is not None:
        assert rel_err(gbg, gbo) < tol
    if act == 'tanh':       # the reference's literal second-order term (kernel grad = 2 on d_dx alone, fused_bias_act.py:165-168) is this one at dy = 1
        from inclusivegan_amd import hip_ops
        y = fused_bias_act(xg.detach(), bg.detach(), act=act)
        d_dx = torch.randn_like(y)
        lit = hip_ops.fused_bias_act_raw(d_dx, None, y, 2, 4, 0.0, 1.0, 8, 1)
        xs = xg.detach().clone().requires_grad_(True)
        ys = fused_bias_act(xs, bg.detach(), act=act)
        (dx,) = torch.autograd.grad(ys, xs, torch.ones_like(ys), create_graph=True)
        (d_x,) = torch.autograd.grad(dx, xs, d_dx)
        assert rel_err(d_x, lit) < 1e-5


def test_fused_bias_act_kernel_table(cuda_device):
    """grad = 0/1/2 kernels against the element-wise restatement of fused_bias_act.cu:64-111."""
    from oracle import fused_bias_act as O
    from inclusivegan_amd import hip_ops
    rng = np.random.RandomState(11)
    x = torch.from_numpy(rng.randn(6, 8).astype(np.float32)); ref = torch.from_numpy(rng.randn(6, 8).astype(np.float32) * 0.7)
    b = torch.from_numpy(rng.randn(8).astype(np.float32))
    for act_idx in range(1, 10):
        for grad in (0, 1, 2):
            want = O.fused_bias_act_kernel_ref(x.double(), b.double(), ref.double() if grad else None, grad, act_idx, 0.2, 1.3, 1)
            got = hip_ops.fused_bias_act_raw(x.to(cuda_device), b.to(cuda_device), ref.to(cuda_device) if grad else None,
                                             grad, act_idx, 0.2, 1.3, 8, 1)
            assert rel_err(got.reshape(-1), want) < 1e-5, (act_idx, grad)


@pytest.mark.parametrize('shape', [(2, 8, 5, 6), (6, 512, 32, 32), (3, 128, 17, 9), (4, 1024, 4, 4), (5, 12, 3, 3), (7, 512)])
@pytest.mark.parametrize('act', ['lrelu', 'linear', 'relu'])
def test_bias_act_noise_fused_epilogue(shape, act, cuda_device):
    """x + noise*strength -> bias -> act (networks_stylegan2.py:351-357): forward, dx, db, dstrength, and
    the create_graph path, against the oracle composition."""
    from oracle import fused_bias_act as O
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import activation_funcs
    rng = np.random.RandomState(len(shape) * 31 + shape[1])
    spec = activation_funcs[act]
    x = torch.from_numpy(rng.randn(*shape)); b = torch.from_numpy(rng.randn(shape[1]))
    has_noise = len(shape) == 4
    noise = torch.from_numpy(rng.randn(shape[0], 1, shape[2], shape[3])) if has_noise else None
    st = torch.tensor(0.37, dtype=torch.float64)
    xo = x.clone().requires_grad_(True); bo = b.clone().requires_grad_(True); so = st.clone().requires_grad_(True)
    yo = O.fused_bias_act(xo + noise * so if has_noise else xo, bo, act=act)
    xg = x.float().to(cuda_device)
    xg = (xg.contiguous(memory_format=torch.channels_last) if has_noise else xg).requires_grad_(True)
    bg = b.float().to(cuda_device).requires_grad_(True); sg = st.float().to(cuda_device).requires_grad_(True)
    ng = noise.float().to(cuda_device) if has_noise else None
    yg = hip_ops.bias_act_noise(xg, bg, ng, sg if has_noise else None, spec.hip_idx, spec.def_alpha or 0.0, spec.def_gain)
    assert rel_err(yg, yo) < 1e-5
    dy = torch.from_numpy(rng.randn(*shape))
    ins_o = [xo, bo] + ([so] if has_noise else []); ins_g = [xg, bg] + ([sg] if has_noise else [])
    go = torch.autograd.grad(yo, ins_o, dy, create_graph=True)
    gg = torch.autograd.grad(yg, ins_g, dy.float().to(cuda_device), retain_graph=True)
    gg2 = torch.autograd.grad(yg, ins_g, dy.float().to(cuda_device), create_graph=True)
    for a_, b_, c_ in zip(gg, go, gg2):
        assert rel_err(a_, b_) < 2e-5
        assert rel_err(c_, b_) < 2e-5


# ----------------------------------------------------------------------------- conv2d family
def _conv_oracle(x, w, stride, up, pad, out_hw):
    """Direct statement of include/igan_hip.h's conv formula with torch ops (fp64)."""
    import torch.nn.functional as F
    n, c, h, wd = x.shape
    if up > 1:
        xu = torch.zeros(n, c, (h - 1) * up + 1, (wd - 1) * up + 1, dtype=x.dtype)
        xu[:, :, ::up, ::up] = x
    else:
        xu = x
    kh, kw = w.shape[0], w.shape[1]
    oh, ow = out_hw
    need_h = (oh - 1) * stride + kh
    need_w = (ow - 1) * stride + kw
    pad_b = max(need_h - pad - xu.shape[2], 0)
    pad_r = max(need_w - pad - xu.shape[3], 0)
    xp = F.pad(xu, [pad, pad_r, pad, pad_b])
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride)
    return y[:, :, :oh, :ow]


CONV_CASES = [
    # (N, Cin, H, W, Cout, K, stride, up, pad, out)   out=None -> derived
    (2, 32, 8, 8, 64, 3, 1, 1, 1, None),     # SAME 3x3
    (2, 16, 9, 7, 40, 3, 1, 1, 1, None),     # ragged sizes, Cout % 32 != 0
    (3, 512, 4, 4, 512, 3, 1, 1, 1, None),   # 4x4 layer shape (split-K path)
    (2, 48, 6, 6, 3, 1, 1, 1, 0, None),      # ToRGB 1x1, Cout = 3
    (2, 3, 16, 16, 32, 1, 1, 1, 0, None),    # FromRGB 1x1, Cin = 3
    (2, 24, 5, 5, 36, 3, 1, 2, 2, 11),       # transposed conv (up 2): (H-1)*2+3
    (1, 64, 8, 8, 128, 3, 1, 2, 2, 17),
    (2, 20, 11, 11, 28, 3, 2, 1, 0, 5),      # VALID stride 2
    (2, 16, 9, 9, 24, 1, 2, 1, 0, 5),        # 1x1 stride 2 (D Skip)
    (6, 513, 4, 4, 512, 3, 1, 1, 1, None),   # after mbstd: Cin = 513
    (7, 64, 1, 1, 40, 1, 1, 1, 0, None),     # dense layer as 1x1 conv, M = 7 (small-batch dense kernels)
    (24, 512, 1, 1, 512, 1, 1, 1, 0, None),  # mapping / style dense at the G batch
    (3, 512, 1, 1, 512, 1, 1, 1, 0, None),   # ... at the path-length batch
    (12, 520, 1, 1, 36, 1, 1, 1, 0, None),   # K spans three LDS tiles with a ragged last one
    (32, 20, 1, 1, 3, 1, 1, 1, 0, None),     # 32 rows, Cout = 3 (weight gradient falls back to the MFMA path)
    (33, 64, 1, 1, 64, 1, 1, 1, 0, None),    # 33 rows: the 48-row form of the dense kernels
    (48, 512, 1, 1, 512, 1, 1, 1, 0, None),  # both latent sets of a generator pass through the mapping network at once
    (64, 96, 1, 1, 40, 1, 1, 1, 0, None),    # 64 rows
    (65, 64, 1, 1, 64, 1, 1, 1, 0, None),    # 65 rows: too many for the dense kernels -> MFMA tiles
    (5, 30, 1, 1, 16, 1, 1, 1, 0, None),     # Cin % 4 != 0: MFMA path
    (2, 3, 12, 12, 64, 3, 1, 1, 1, None),    # VGG conv1_1 (thin-input kernel; its data gradient: thin-output 3x3)
    (3, 128, 9, 7, 3, 1, 1, 1, 0, None),     # ToRGB at a thin-kernel size: 32 lanes per pixel, ragged pixel count
    (2, 512, 4, 4, 3, 1, 1, 1, 0, None),     # ToRGB of the 512-channel layers: two channel blocks per lane
    (2, 64, 5, 5, 4, 3, 1, 1, 1, None),      # thin output with 4 channels, 3x3
    (2, 4, 6, 6, 16, 1, 1, 1, 0, None),      # thin input with 4 channels
    (5, 3, 16, 16, 512, 1, 1, 1, 0, None),   # FromRGB into 512 channels: two channel blocks
    (1, 128, 32, 32, 128, 3, 1, 1, 1, None), # bigger tile grid
    (24, 512, 8, 8, 512, 3, 1, 1, 1, None),  # 48 sliced tiles: fix-up row groups of 5 overhang the 128-row tile
    (10, 64, 64, 64, 128, 3, 1, 1, 1, None), # 320 tiles: one whole round + a sliced tail of 64
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_forward_dgrad_wgrad(case, cuda_device):
    from inclusivegan_amd import hip_ops
    N, Cin, H, W, Cout, K, stride, up, pad, out = case
    rng = np.random.RandomState(abs(hash(case)) % 2**31)
    if out is None:
        oh, ow = H, W
    else:
        oh = out
        ow = out if H == W else None
    if ow is None:
        ow = W
    x = torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32))
    w = torch.from_numpy((rng.randn(K, K, Cin, Cout) / np.sqrt(K * K * Cin)).astype(np.float32))
    xo = x.double().requires_grad_(True); wo = w.double().requires_grad_(True)
    alpha = 1.0 if N % 2 else 0.73          # the kernels' output multiplier (runtime weight scale)
    yo = _conv_oracle(xo, wo, stride, up, pad, (oh, ow)) * alpha
    xg = to_nhwc_cuda(x, cuda_device).requires_grad_(True)
    wg = w.to(cuda_device).requires_grad_(True)
    geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad, alpha)
    yg = hip_ops.conv2d(xg, wg, geom, (oh, ow))
    assert tuple(yg.shape) == tuple(yo.shape)
    assert rel_err(yg, yo) < 3e-5
    dy = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    gxo, gwo = torch.autograd.grad(yo, [xo, wo], dy.double())
    gxg, gwg = torch.autograd.grad(yg, [xg, wg], to_nhwc_cuda(dy, cuda_device))
    assert rel_err(gxg, gxo) < 3e-5
    assert rel_err(gwg, gwo) < 3e-5


@pytest.mark.parametrize('case', [(6, 128, 32, 32, 128, 3, 1, 1, 1, 32), (18, 64, 32, 32, 64, 3, 1, 1, 1, 32), (6, 512, 8, 8, 512, 3, 1, 1, 1, 8),
                                  (4, 64, 16, 16, 64, 3, 1, 2, 2, 33), (6, 128, 64, 64, 3, 1, 1, 1, 0, 64), (6, 3, 64, 64, 128, 1, 1, 1, 0, 64),
                                  (24, 512, 1, 1, 512, 1, 1, 1, 0, 1)])
def test_conv_family_is_bit_reproducible(case, cuda_device):
    """Every reduction in the conv family has a fixed order (sliced tails, split weight gradients, thin-channel and
    dense kernels): two runs on the same inputs must agree bit for bit -- this is also the race detector."""
    from inclusivegan_amd import hip_ops
    N, Cin, H, W, Cout, K, stride, up, pad, out = case
    g = torch.Generator(device='cpu').manual_seed(N * 131 + Cin)
    x = to_nhwc_cuda(torch.randn(N, Cin, H, W, generator=g), cuda_device)
    w = (torch.randn(K, K, Cin, Cout, generator=g) / (K * K * Cin) ** 0.5).to(cuda_device)
    dy = to_nhwc_cuda(torch.randn(N, Cout, out, out, generator=g), cuda_device)
    s = (torch.rand(N, Cin, generator=g) + 0.5).to(cuda_device)
    geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad, 0.5)
    runs = []
    for _ in range(3):
        y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s)
        dx = hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, W), Cin, w_transposed=True)
        dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s)
        runs.append((y, dx, dw))
        torch.empty(1 << 22, device=cuda_device).normal_()     # churn the allocator / caches between runs
    for r in runs[1:]:
        for a_, b_ in zip(r, runs[0]):
            assert torch.equal(a_, b_)


@pytest.mark.parametrize('case', [(2, 16, 9, 9, 24, 3, 1, 1, 3), (12, 128, 32, 32, 128, 3, 1, 1, 3), (3, 20, 11, 11, 28, 3, 2, 0, 3),
                                  (6, 513, 4, 4, 512, 3, 1, 1, 3), (4, 64, 16, 16, 64, 3, 1, 1, 2), (2, 32, 8, 8, 32, 1, 1, 0, 1)])
def test_conv_bias_act_epilogue(case, cuda_device):
    """ConvBiasActFn (bias + activation in the convolution's epilogue, direct and sliced-tail tiles) == convolution followed by
    the stand-alone epilogue kernel: values, first-order gradients, and an R1-style second-order gradient."""
    from inclusivegan_amd import hip_ops
    N, Cin, H, W, Cout, K, stride, pad, act = case
    rng = np.random.RandomState(N * 7 + Cin)
    oh = (H + 2 * pad - K) // stride + 1
    x = to_nhwc_cuda(torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32)), cuda_device)
    w = torch.from_numpy((rng.randn(K, K, Cin, Cout) / np.sqrt(K * K * Cin)).astype(np.float32)).to(cuda_device)
    b = torch.from_numpy((rng.randn(Cout) * 0.3).astype(np.float32)).to(cuda_device)
    dy = to_nhwc_cuda(torch.from_numpy(rng.randn(N, Cout, oh, oh).astype(np.float32)), cuda_device)
    geom = hip_ops.ConvGeom(K, K, stride, 1, pad, pad, 0.8)
    alpha, gain = 0.2, float(np.sqrt(2))
    def run(fused):
        xs = x.clone().requires_grad_(True); ws = w.clone().requires_grad_(True); bs = b.clone().requires_grad_(True)
        if fused:
            y = hip_ops.ConvBiasActFn.apply(xs, ws, bs, geom, (oh, oh), act, alpha, gain)
        else:
            y = hip_ops.bias_act_noise(hip_ops.conv2d(xs, ws, geom, (oh, oh)), bs, None, None, act, alpha, gain)
        g1 = torch.autograd.grad(y, [xs, ws, bs], dy, create_graph=True)
        pen = (g1[0] * g1[0]).sum()
        g2 = torch.autograd.grad(pen, [ws], allow_unused=True)
        return [y] + list(g1) + [g2[0]]
    assert hip_ops.conv_bias_act_fusable(x, Cout, act)
    for a_, b_ in zip(run(True), run(False)):
        assert rel_err(a_, b_) < 3e-5


def test_conv2d_double_backward(cuda_device):
    """R1-style penalty through conv (needs d(dgrad)/dw and d(dgrad)/d(dy))."""
    from inclusivegan_amd import hip_ops
    rng = np.random.RandomState(2)
    for (stride, up, pad, H, out) in [(1, 1, 1, 8, 8), (2, 1, 0, 9, 4), (1, 2, 2, 5, 11)]:
        x = torch.from_numpy(rng.randn(2, 16, H, H).astype(np.float32))
        w = torch.from_numpy((rng.randn(3, 3, 16, 24) / 12).astype(np.float32))
        def pen(conv, x, w):
            y = conv(x, w)
            (gx,) = torch.autograd.grad((y * y).sum() * 0.5, x, create_graph=True)
            return (gx * gx).sum()
        xo = x.double().requires_grad_(True); wo = w.double().requires_grad_(True)
        po = pen(lambda a, b: _conv_oracle(a, b, stride, up, pad, (out, out)), xo, wo)
        gxo, gwo = torch.autograd.grad(po, [xo, wo])
        xg = to_nhwc_cuda(x, cuda_device).requires_grad_(True); wg = w.to(cuda_device).requires_grad_(True)
        geom = hip_ops.ConvGeom(3, 3, stride, up, pad, pad)
        pg = pen(lambda a, b: hip_ops.conv2d(a, b, geom, (out, out)), xg, wg)
        gxg, gwg = torch.autograd.grad(pg, [xg, wg])
        assert rel_err(pg, po) < 5e-5
        assert rel_err(gxg, gxo) < 5e-5
        assert rel_err(gwg, gwo) < 5e-5


@pytest.mark.parametrize('mode', ['plain', 'up', 'down'])
@pytest.mark.parametrize('demod', [True, False])
def test_modconv_fused_scales(mode, demod, cuda_device):
    """ModConv2dFn (scales folded into the MFMA kernel) == composite, values and all four gradients,
    first order and through create_graph."""
    from inclusivegan_amd import hip_ops
    rng = np.random.RandomState(9)
    N, Cin, Cout, H = 3, 20, 28, 6
    geom, out = {'plain': (hip_ops.ConvGeom(3, 3, 1, 1, 1, 1), 6), 'up': (hip_ops.ConvGeom(3, 3, 1, 2, 2, 2), 13),
                 'down': (hip_ops.ConvGeom(3, 3, 2, 1, 0, 0), 2)}[mode]
    x = torch.from_numpy(rng.randn(N, Cin, H, H).astype(np.float32))
    w = torch.from_numpy((rng.randn(3, 3, Cin, Cout) / 13).astype(np.float32))
    s = torch.from_numpy((rng.randn(N, Cin) * 0.3 + 1).astype(np.float32))
    d = torch.from_numpy((rng.rand(N, Cout) + 0.5).astype(np.float32)) if demod else None
    def run(fn, create_graph):
        xs = to_nhwc_cuda(x, cuda_device).requires_grad_(True); ws = w.to(cuda_device).requires_grad_(True)
        ss = s.to(cuda_device).requires_grad_(True); ds = d.to(cuda_device).requires_grad_(True) if demod else None
        y = fn(xs, ws, ss, ds)
        ins = [xs, ws, ss] + ([ds] if demod else [])
        dy = torch.from_numpy(np.random.RandomState(1).randn(*y.shape).astype(np.float32)).to(cuda_device)
        gs = torch.autograd.grad(y, ins, dy, create_graph=create_graph)
        return y, gs
    y1, g1 = run(lambda a, b, c, e: hip_ops.ModConv2dFn.apply(a, b, c, e, geom, (out, out)), False)
    y2, g2 = run(lambda a, b, c, e: hip_ops.modconv_composite(a, b, c, e, geom, (out, out)), False)
    y3, g3 = run(lambda a, b, c, e: hip_ops.ModConv2dFn.apply(a, b, c, e, geom, (out, out)), True)
    assert rel_err(y1, y2) < 3e-5
    for a, b, c in zip(g1, g2, g3):
        assert rel_err(a, b) < 5e-5
        assert rel_err(c, b) < 5e-5


@pytest.mark.parametrize('cin', [128, 512])
def test_modconv_torgb_thin_kernels(cin, cuda_device):
    """ToRGB (1x1, Cout = 3, modulated, no demodulation, alpha != 1) runs on the thin-channel kernels: forward,
    data gradient (thin input, transposed weights), weight gradient (per-sample scale) and style gradient vs the composite."""
    from inclusivegan_amd import hip_ops
    rng = np.random.RandomState(cin)
    N, H = 3, 10
    geom = hip_ops.ConvGeom(1, 1, 1, 1, 0, 0, 0.37)
    x = torch.from_numpy(rng.randn(N, cin, H, H).astype(np.float32))
    w = torch.from_numpy(rng.randn(1, 1, cin, 3).astype(np.float32))
    s = torch.from_numpy((rng.randn(N, cin) * 0.3 + 1).astype(np.float32))
    dy = torch.from_numpy(rng.randn(N, 3, H, H).astype(np.float32))
    outs = []
    for fn in (hip_ops.ModConv2dFn.apply, hip_ops.modconv_composite):
        xs = to_nhwc_cuda(x, cuda_device).requires_grad_(True); ws = w.to(cuda_device).requires_grad_(True)
        ss = s.to(cuda_device).requires_grad_(True)
        y = fn(xs, ws, ss, None, geom, (H, H))
        outs.append((y,) + torch.autograd.grad(y, [xs, ws, ss], to_nhwc_cuda(dy, cuda_device)))
    yo = torch.einsum('nchw,co->nohw', x.double() * s.double()[:, :, None, None], w.double()[0, 0]) * 0.37
    assert rel_err(outs[0][0], yo) < 2e-5
    for a_, b_ in zip(outs[0], outs[1]):
        assert rel_err(a_, b_) < 3e-5


# ----------------------------------------------------------------------------- style path (s, d)
@pytest.mark.parametrize('case', [(24, 512, 512, 512, 3, True), (3, 512, 256, 128, 3, True), (6, 512, 128, 3, 1, False),
                                  (32, 64, 36, 20, 3, True), (12, 512, 512, 256, 3, True)])
def test_style_mod_fused_vs_oracle(case, cuda_device):
    """s = c_a y.A + b + 1, d = rsqrt(c_w^2 s^2.sum_taps(w^2) + 1e-8) (networks_stylegan2.py:99-107): fused kernels vs
    an fp64 evaluation, forward, first-order gradients of all four inputs, and a second-order check through the
    differentiable backward (create_graph)."""
    from inclusivegan_amd import hip_ops
    N, L, Cin, Cout, K, demod = case
    rng = np.random.RandomState(N + Cin)
    y = rng.randn(N, 3, L)[:, 1]                      # a strided [N, L] slice like dlatents[:, i]
    A = rng.randn(L, Cin); b = rng.randn(Cin) * 0.1; w = rng.randn(K, K, Cin, Cout)
    c_a, c_w = 1.0 / np.sqrt(L), 1.0 / np.sqrt(K * K * Cin)
    gs = rng.randn(N, Cin); gd = rng.randn(N, Cout)

    def oracle(y, A, b, w):
        s = c_a * (y @ A) + b + 1.0
        d = torch.rsqrt(c_w * c_w * ((s * s) @ (w * w).sum(dim=(0, 1))) + 1e-8) if demod else None
        return s, d

    to = [torch.from_numpy(np.ascontiguousarray(t)).requires_grad_(True) for t in (y, A, b, w)]
    so, do = oracle(*to)
    yfull = torch.from_numpy(rng.randn(N, 3, L)).float().to(cuda_device)
    yfull[:, 1] = torch.from_numpy(y).float().to(cuda_device)
    yg = yfull.requires_grad_(True)
    tg = [torch.from_numpy(np.ascontiguousarray(t)).float().to(cuda_device).requires_grad_(True) for t in (A, b, w)]
    wsq = hip_ops.sumsq_taps_raw(tg[2].detach()) if demod else None
    assert hip_ops.style_mod_fusable(yg[:, 1], tg[0], tg[2], demod)
    sg, dg = hip_ops.style_mod(yg.unbind(1)[1], tg[0], tg[1], tg[2], wsq, c_a, c_w, demod)
    assert rel_err(sg, so) < 1e-5
    outs_o, outs_g = [so], [sg]
    go, gg = [torch.from_numpy(gs)], [torch.from_numpy(gs).float().to(cuda_device)]
    if demod:
        assert rel_err(dg, do) < 1e-5
        outs_o.append(do); outs_g.append(dg)
        go.append(torch.from_numpy(gd)); gg.append(torch.from_numpy(gd).float().to(cuda_device))
    ins_o = to if demod else to[:3]
    ins_g = [yg] + (tg if demod else tg[:2])
    grads_o = torch.autograd.grad(outs_o, ins_o, go, create_graph=True)
    grads_g = torch.autograd.grad(outs_g, ins_g, gg, retain_graph=True)
    assert rel_err(grads_g[0][:, 1], grads_o[0]) < 2e-5
    assert float(grads_g[0][:, 0].abs().max()) == 0.0
    for a_, b_ in zip(grads_g[1:], grads_o[1:]):
        assert rel_err(a_, b_) < 2e-5
    # second order: gradient of |d out / d y|^2 w.r.t. everything (the path-length regulariser's shape)
    g2 = torch.autograd.grad(outs_g, ins_g, gg, create_graph=True)
    pen_g = (g2[0] * g2[0]).sum()
    pen_o = (grads_o[0] * grads_o[0]).sum()
    h_g = torch.autograd.grad(pen_g, ins_g, allow_unused=True)      # without demodulation s is linear: some are unused
    h_o = torch.autograd.grad(pen_o, ins_o, allow_unused=True)
    for i, (a_, b_) in enumerate(zip(h_g, h_o)):
        if b_ is None or float(b_.abs().max()) == 0.0:
            assert a_ is None or float(a_.abs().max()) < 1e-6
            continue
        assert rel_err(a_[:, 1] if i == 0 else a_, b_) < 1e-4


# ----------------------------------------------------------------------------- minibatch stddev
@pytest.mark.parametrize('N,G', [(6, 6), (12, 6), (4, 6), (8, 4)])
def test_mbstd(N, G, cuda_device):
    from oracle import networks_stylegan2 as ON
    from inclusivegan_amd.training.networks_stylegan2 import minibatch_stddev_layer
    rng = np.random.RandomState(N * 10 + G)
    x = torch.from_numpy(rng.randn(N, 16, 4, 4))
    xo = x.clone().requires_grad_(True)
    yo = ON.minibatch_stddev_layer(xo, G)
    xg = to_nhwc_cuda(x, cuda_device).requires_grad_(True)
    yg = minibatch_stddev_layer(xg, G)
    assert rel_err(yg, yo) < 1e-5
    dy = torch.from_numpy(rng.randn(*yo.shape))
    (gxo,) = torch.autograd.grad(yo, xo, dy, create_graph=True)
    (gxg,) = torch.autograd.grad(yg, xg, to_nhwc_cuda(dy, cuda_device))
    assert rel_err(gxg, gxo) < 1e-5
    # second order
    yg2 = minibatch_stddev_layer(xg, G)
    (gxg2,) = torch.autograd.grad(yg2, xg, to_nhwc_cuda(dy, cuda_device), create_graph=True)
    (hg,) = torch.autograd.grad((gxg2 * gxg2).sum(), xg)
    (ho,) = torch.autograd.grad((gxo * gxo).sum(), xo)
    assert rel_err(hg, ho) < 1e-4


# ----------------------------------------------------------------------------- LPIPS layer distance
@pytest.mark.parametrize('shape', [(3, 64, 16, 16), (2, 128, 9, 7), (6, 256, 32, 32), (2, 512, 8, 8), (1, 512, 1, 3), (5, 64, 128, 128)])
def test_lpips_layer_distance(shape, cuda_device):
    """Fused normalise/diff/lin/spatial-sum kernel vs the oracle's per-layer arithmetic (oracle/lpips.py:31,38),
    forward and both gradients, including all-zero (post-ReLU) pixels."""
    from inclusivegan_amd import hip_ops
    N, C, H, W = shape
    rng = np.random.RandomState(C + H)
    fa = np.maximum(rng.randn(N, C, H, W), 0.0)
    fb = np.maximum(rng.randn(N, C, H, W) + 0.3, 0.0)
    fa[0, :, 0, 0] = 0.0                                   # a dead pixel: u = 0 / (0 + 1e-10)
    lin = np.abs(rng.randn(C)) / C
    g = rng.randn(N)

    def oracle(a, b):
        ua = a / (torch.sqrt(torch.sum(a * a, dim=1, keepdim=True)) + 1e-10)
        ub = b / (torch.sqrt(torch.sum(b * b, dim=1, keepdim=True)) + 1e-10)
        return ((ua - ub) ** 2 * torch.from_numpy(lin).view(1, C, 1, 1)).sum(dim=(1, 2, 3))

    ao = torch.from_numpy(fa).requires_grad_(True)
    bo = torch.from_numpy(fb).requires_grad_(True)
    do = oracle(ao, bo)
    gao, gbo = torch.autograd.grad(do, [ao, bo], torch.from_numpy(g))
    ag = to_nhwc_cuda(torch.from_numpy(fa), cuda_device).requires_grad_(True)
    bg = to_nhwc_cuda(torch.from_numpy(fb), cuda_device).requires_grad_(True)
    ling = torch.from_numpy(lin).float().to(cuda_device)
    dg = hip_ops.LpipsLayerFn.apply(ag, bg, ling)
    assert rel_err(dg, do) < 1e-5
    gag, gbg = torch.autograd.grad(dg, [ag, bg], torch.from_numpy(g).float().to(cuda_device))
    # At the dead pixel the oracle's autograd gives NaN (sqrt'(0) * 0); the kernel returns the limit
    # g * q / eps with u = 0, which is checked analytically instead.
    live = torch.ones(N, 1, H, W, dtype=torch.bool)
    live[0, :, 0, 0] = False
    zero = torch.zeros((), dtype=torch.float64)
    assert rel_err(torch.where(live, gag.cpu().double(), zero), torch.where(live, gao, zero)) < 2e-5
    assert rel_err(gbg, gbo) < 2e-5
    vb = bo.detach()[0, :, 0, 0]
    vb = vb / (torch.sqrt((vb * vb).sum()) + 1e-10)
    dead = g[0] * 1e10 * 2.0 * torch.from_numpy(lin) * (0.0 - vb)
    assert rel_err(gag[0, :, 0, 0], dead) < 1e-5
    # only one side needs a gradient (the reconstruction term: reals are constants)
    dg2 = hip_ops.LpipsLayerFn.apply(ag, bg.detach(), ling)
    (gag2,) = torch.autograd.grad(dg2, [ag], torch.from_numpy(g).float().to(cuda_device))
    assert torch.equal(gag2, gag)


def test_lpips_pair_table_matches_layer_function(cuda_device):
    """The G loss's four distances in one Function (sample-pair tables, all layers, one row sum) against the per-pair
    LpipsLayerFn on batch slices -- which test_lpips_layer_distance pins to the oracle -- forward and the gradient w.r.t. the
    generated features (the interpolated images get two contributions)."""
    from inclusivegan_amd import hip_ops
    n = 3
    rng = np.random.RandomState(5)
    shapes = [(64, 16, 16), (128, 8, 8), (512, 4, 4)]
    fg = [to_nhwc_cuda(torch.from_numpy(np.maximum(rng.randn(3 * n, *sh), 0).astype(np.float32)), cuda_device).requires_grad_(True) for sh in shapes]
    fr = [to_nhwc_cuda(torch.from_numpy(np.maximum(rng.randn(2 * n, *sh) + 0.2, 0).astype(np.float32)), cuda_device) for sh in shapes]
    lins = [torch.from_numpy((np.abs(rng.randn(sh[0])) / sh[0] / (sh[1] * sh[2])).astype(np.float32)).to(cuda_device) for sh in shapes]
    g = torch.from_numpy(rng.randn(4 * n).astype(np.float32)).to(cuda_device)
    d = hip_ops.LpipsPairsFn.apply(*lins, *fg, *fr, n, len(shapes))
    grads = torch.autograd.grad(d, fg, g)
    ref = torch.zeros(4 * n, device=cuda_device)
    fg2 = [t.detach().clone().requires_grad_(True) for t in fg]
    for a, b, lin in zip(fg2, fr, lins):
        ref = ref + torch.cat([hip_ops.LpipsLayerFn.apply(a[:2 * n], b, lin), hip_ops.LpipsLayerFn.apply(a[2 * n:], b[n:], lin),
                               hip_ops.LpipsLayerFn.apply(a[2 * n:], b[:n], lin)])
    grads2 = torch.autograd.grad(ref, fg2, g)
    assert rel_err(d, ref) < 1e-6
    for ga, gb in zip(grads, grads2):
        assert rel_err(ga, gb) < 1e-6
    # reals are constants: no gradient is produced for them, and a layer whose features need none is skipped
    assert all(not t.requires_grad for t in fr)


@pytest.mark.parametrize('shape', [(3, 64, 16, 16), (2, 128, 6, 10), (1, 512, 2, 2)])
def test_maxpool_tap_matches_oracle(shape, cuda_device):
    """x -> (tap, 2x2 max-pool) and the fused gradient d_tap + route(d_pool) against the framework pooling the oracle uses
    (oracle/lpips.py:27) on CPU, bit for bit, with the ties ReLU produces (windows of equal values, all-zero windows)."""
    from inclusivegan_amd import hip_ops
    N, C, H, W = shape
    rng = np.random.RandomState(H * W)
    x = np.maximum(rng.randn(N, C, H, W), 0).astype(np.float32)
    x[0, :, 0:2, 0:2] = 0.0
    x[0, 1, 0:2, 2:4] = 0.75                                    # a window of four equal positive values
    g_tap = rng.randn(N, C, H, W).astype(np.float32)
    g_pool = rng.randn(N, C, H // 2, W // 2).astype(np.float32)
    xo = torch.from_numpy(x).requires_grad_(True)
    yo = torch.nn.functional.max_pool2d(xo, 2)
    (gxo,) = torch.autograd.grad([xo * 1.0, yo], [xo], [torch.from_numpy(g_tap), torch.from_numpy(g_pool)])
    xg = to_nhwc_cuda(torch.from_numpy(x), cuda_device).requires_grad_(True)
    tap, yg = hip_ops.PoolTapFn.apply(xg)
    assert torch.equal(tap.cpu(), xo.detach()) and torch.equal(yg.cpu(), yo.detach())
    (gxg,) = torch.autograd.grad([tap, yg], [xg], [to_nhwc_cuda(torch.from_numpy(g_tap), cuda_device), to_nhwc_cuda(torch.from_numpy(g_pool), cuda_device)])
    assert torch.equal(gxg.cpu(), gxo)
    # the pooled branch alone (last use of a tap without a distance), and the tap alone
    tap, yg = hip_ops.PoolTapFn.apply(xg)
    (g1,) = torch.autograd.grad([yg], [xg], [to_nhwc_cuda(torch.from_numpy(g_pool), cuda_device)])
    (g1o,) = torch.autograd.grad([torch.nn.functional.max_pool2d(xo, 2)], [xo], [torch.from_numpy(g_pool)])
    assert torch.equal(g1.cpu(), g1o)
    with pytest.raises(ValueError):
        hip_ops.PoolTapFn.apply(torch.zeros(1, 64, 3, 4, device=cuda_device).contiguous(memory_format=torch.channels_last))


def test_lpips_layer_argument_errors(cuda_device):
    from inclusivegan_amd import hip_ops
    x = torch.zeros(1, 96, 4, 4, device=cuda_device).contiguous(memory_format=torch.channels_last)
    with pytest.raises(ValueError):
        hip_ops.LpipsLayerFn.apply(x, x, torch.zeros(96, device=cuda_device))


# ----------------------------------------------------------------------------- nearest neighbour
def test_nn1_exact_vs_oracle_and_dci_golden(cuda_device):
    import os
    from oracle import nn as ONN
    from inclusivegan_amd.dci_code.dci import DCI
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dci_golden.npz'))
    data, queries = g['data'], g['queries']
    db = DCI(data.shape[1], 3, 15, device=cuda_device)
    db.cand_chunk = 700     # force several candidate batches + a ragged tail
    db.add(data, num_levels=3, field_of_view=10, prop_to_retrieve=0.002)
    idx, dist = db.query(queries, num_neighbours=1, field_of_view=200, prop_to_retrieve=1.0)
    idx = np.array([i[0] for i in idx]); dist = np.array([d[0] for d in dist])
    oidx, odist = ONN.nearest_neighbour(data, queries)
    assert idx.dtype == np.int32 and dist.dtype == np.float64
    assert (idx == oidx).all()
    assert np.abs(dist - odist).max() <= 1e-5 * odist.max()
    # the reference's (approximate) DCI can never beat the exact search
    assert (dist <= g['dist'] * (1 + 1e-4) + 1e-6).all()
    assert (idx == g['idx']).mean() >= 0.9
    db.reset()
    assert db.num_points == 0


def test_nn1_large_dim_and_ties(cuda_device):
    from oracle import nn as ONN
    from inclusivegan_amd.dci_code.dci import DCI
    rng = np.random.RandomState(0)
    data = rng.uniform(-1, 1, size=(300, 3 * 32 * 32)).astype(np.float32)
    data[17] = data[5]                       # exact duplicate -> the lower index must win
    q = data[[5, 100, 299]] + rng.randn(3, data.shape[1]).astype(np.float32) * 0.01
    db = DCI(data.shape[1], device=cuda_device)
    db.add(torch.from_numpy(data).to(cuda_device))
    idx, dist = db.query(torch.from_numpy(q).to(cuda_device), num_neighbours=1)
    oidx, odist = ONN.nearest_neighbour(data, q)
    assert [int(i[0]) for i in idx] == [5, 100, 299] == list(oidx)
    assert np.abs(np.array([d[0] for d in dist]) - odist).max() < 1e-12 * odist.max()   # fp32 data: the fp64 direct-difference distance is exact


# ----------------------------------------------------------------------------- optimizer kernels
def test_adam_ema_finite_check(cuda_device):
    from oracle import optimizer as OO
    from inclusivegan_amd import hip_ops
    rng = np.random.RandomState(4)
    n = 10007
    w = rng.randn(n).astype(np.float32); m = np.zeros(n, np.float32)
    wt = torch.from_numpy(w.copy()).to(cuda_device); mt = torch.zeros(n, device=cuda_device); vt = torch.zeros(n, device=cuda_device)
    powt = torch.ones(2, device=cuda_device); flag = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    adam = OO.SimpleAdam(n, learning_rate=0.002 * 0.8, beta1=0.0, beta2=0.99 ** 0.8, epsilon=1e-8)
    wo = w.copy()
    for step in range(5):
        g = rng.randn(n).astype(np.float32)
        if step == 2:
            g[123] = np.inf        # overflow step: skipped on both sides, beta powers must not advance
        gt = torch.from_numpy(g).to(cuda_device)
        flag.zero_()
        hip_ops.finite_check_raw(gt, flag)
        hip_ops.adam_step_raw(wt, gt, mt, vt, 0.002 * 0.8, 0.0, 0.99 ** 0.8, 1e-8, powt, flag)
        applied = adam.apply(wo, g)
        assert bool(flag.item()) == (not applied)
    assert rel_err(wt, wo) < 2e-6
    assert rel_err(vt, adam.v) < 2e-6
    assert abs(float(powt[1]) - float(adam.b2pow)) < 1e-6
    src = torch.from_numpy(rng.randn(n).astype(np.float32)).to(cuda_device)
    dst0 = wt.clone()
    hip_ops.ema_raw(wt, src, 0.9995)
    assert rel_err(wt, OO.ema(dst0.cpu().numpy(), src.cpu().numpy(), 0.9995)) < 1e-6


def test_simple_adam_class_matches_reference_arithmetic(cuda_device):
    """tflib.SimpleAdam (dnnlib/tflib/optimizer.py:290-336) on two loose variables: three steps against the NumPy restatement;
    both variables of a step see the same beta-power pair."""
    from oracle import optimizer as OO
    from inclusivegan_amd.dnnlib.tflib.optimizer import SimpleAdam
    rng = np.random.RandomState(9)
    a0 = rng.randn(40, 8).astype(np.float32); b0 = rng.randn(64).astype(np.float32)
    a = torch.from_numpy(a0.copy()).to(cuda_device).requires_grad_(True); b = torch.from_numpy(b0.copy()).to(cuda_device).requires_grad_(True)
    opt = SimpleAdam(learning_rate=0.01, beta1=0.9, beta2=0.999)
    oa, ob = OO.SimpleAdam(a0.size, 0.01, 0.9, 0.999, 1e-8), OO.SimpleAdam(b0.size, 0.01, 0.9, 0.999, 1e-8)
    wa, wb = a0.reshape(-1).copy(), b0.copy()
    for _ in range(3):
        loss = (a * a).sum() * 0.5 + (b ** 3).sum()
        gv = opt.compute_gradients(loss, [a, b])
        ga, gb = gv[0][0].cpu().numpy().reshape(-1), gv[1][0].cpu().numpy()
        opt.apply_gradients(gv)
        oa.apply(wa, ga); ob.apply(wb, gb)
    assert rel_err(a.detach().reshape(-1), wa) < 2e-6 and rel_err(b.detach(), wb) < 2e-6
    assert len(opt.variables()) == 5


def test_knn_k_neighbours_vs_oracle(cuda_device):
    """DCI.query(num_neighbours = k > 1) (the exclusive IMLE variant asks for num_samples_factor neighbours,
    training_loop.py:386): indices, order and fp64 distances against the brute-force oracle, including a planted near-tie
    and an exact duplicate (lower index first), across several candidate batches."""
    from tests.imle_cases import exact_knn
    from inclusivegan_amd.dci_code.dci import DCI
    rng = np.random.RandomState(3)
    data = rng.uniform(-1, 1, size=(700, 3 * 16 * 16)).astype(np.float32)
    q = rng.uniform(-1, 1, size=(33, data.shape[1])).astype(np.float32)
    data[41] = q[0]; data[41, 5] += 0.25                       # very close to query 0
    data[300] = data[41]                                       # exact duplicate: 41 must come before 300
    data[555] = q[0]; data[555, 9] += np.float32(0.25 * (1 + 2.0 ** -12))      # near-tie behind them
    db = DCI(data.shape[1], 3, 15, device=cuda_device)
    db.cand_chunk = 256
    db.add(torch.from_numpy(data).to(cuda_device))
    k = 10
    idx, dist = db.query(torch.from_numpy(q).to(cuda_device), num_neighbours=k, field_of_view=200, prop_to_retrieve=1.0)
    oi, od = exact_knn(data, q, k)
    assert len(idx) == 33 and idx[0].dtype == np.int32 and dist[0].dtype == np.float64 and idx[0].shape == (k,)
    assert np.array_equal(np.stack(idx), oi)
    assert np.abs(np.stack(dist) - od).max() <= 1e-12 * od.max()
    assert list(idx[0][:3]) == [41, 300, 555]


# ----------------------------------------------------------------------------- training-scalar bookkeeping
def test_autosummary_device_accumulator(cuda_device):
    """dnnlib/tflib/autosummary.py:45-74: [count, sum] of the finite values, accumulated across calls, mean at flush; the
    device path (one kernel per call) against the same bookkeeping done on the host in double."""
    from inclusivegan_amd.dnnlib.tflib import autosummary as AS
    AS._acc.pop('Test/x', None)
    rng = np.random.RandomState(3)
    want_c, want_s = 0.0, 0.0
    for n in (6, 12, 1000):
        x = rng.randn(n).astype(np.float32)
        x[::5] = np.nan if n == 12 else x[::5]
        if n == 1000:
            x[7] = np.inf
        ok = np.isfinite(x)
        want_c += ok.sum(); want_s += x[ok].astype(np.float64).sum()
        t = torch.from_numpy(x).to(cuda_device)
        assert AS.autosummary('Test/x', t) is t
    acc = AS._acc['Test/x'].cpu().numpy()
    assert acc[0] == want_c and abs(acc[1] - want_s) <= 1e-12 * max(1.0, abs(want_s))
    out = AS.flush()
    assert abs(out['Test/x'] - want_s / want_c) < 1e-12
    assert float(AS._acc['Test/x'].abs().sum()) == 0.0


# ----------------------------------------------------------------------------- fused synthesis layer
@pytest.mark.parametrize('case', [(3, 128, 128, 32, 32, True), (2, 512, 512, 8, 8, True), (4, 64, 96, 16, 16, False), (2, 512, 512, 4, 4, True), (1, 32, 32, 32, 32, True)])
def test_fused_synthesis_layer_matches_two_step_form(case, cuda_device):
    """ModConvBanFn (modulated conv with noise + bias + lrelu in its epilogue; backward with the demodulation gradient recovered
    from the activation output) against ModConv2dFn followed by BiasActNoiseFn -- both pinned to the oracle elsewhere -- for the
    value and every gradient, on MFMA tiles, sliced small layers (4x4, 8x8: several samples per tile) and shared / per-sample noise."""
    from inclusivegan_amd import hip_ops
    N, cin, cout, H, W, per_sample = case
    g = torch.Generator().manual_seed(cin + H)
    dev = cuda_device
    x = (torch.randn(N, cin, H, W, generator=g)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(3, 3, cin, cout, generator=g) / np.sqrt(9 * cin)).to(dev).requires_grad_(True)
    s = (torch.rand(N, cin, generator=g) + 0.5).to(dev).requires_grad_(True)
    d = (torch.rand(N, cout, generator=g) + 0.5).to(dev).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev).requires_grad_(True)
    noise = torch.randn(N if per_sample else 1, 1, H, W, generator=g).to(dev)
    strength = torch.tensor(0.3, device=dev).requires_grad_(True)
    dy = torch.randn(N, cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1, 0.7)
    gain = float(np.sqrt(2))
    leaves = [x, w, s, d, b, strength]
    assert hip_ops.modconv_ban_fusable(x, w, d, b, noise, 3)
    y1 = hip_ops.ModConvBanFn.apply(x, w, s, d, b, noise, strength, geom, (H, W), 3, 0.2, gain)
    g1 = torch.autograd.grad(y1, leaves, dy)
    y0 = hip_ops.ModConv2dFn.apply(x, w, s, d, geom, (H, W))
    y0 = hip_ops.bias_act_noise(y0, b, noise, strength, 3, 0.2, gain)
    g0 = torch.autograd.grad(y0, leaves, dy)
    assert rel_err(y1, y0) < 2e-6
    for name, a, r in zip(('x', 'w', 's', 'd', 'b', 'strength'), g1, g0):
        assert rel_err(a, r) < (2e-5 if name == 'd' else 5e-6), name
    # linear activation (ToRGB-like epilogue) and no gradient wanted for d / strength
    y2 = hip_ops.ModConvBanFn.apply(x, w, s, d.detach(), b, noise, strength.detach(), geom, (H, W), 1, 0.0, 1.0)
    (gx2,) = torch.autograd.grad(y2, [x], dy)
    y3 = hip_ops.bias_act_noise(hip_ops.ModConv2dFn.apply(x, w, s, d.detach(), geom, (H, W)), b, noise, strength.detach(), 1, 0.0, 1.0)
    (gx3,) = torch.autograd.grad(y3, [x], dy)
    assert rel_err(y2, y3) < 2e-6 and rel_err(gx2, gx3) < 5e-6


def test_plugin_first_then_torch_share_one_runtime(cuda_device):
    """A process that loads libigan_hip.so before it ever imports torch (what `build()` followed by `smoke()` does) must still
    launch on PyTorch's streams: the binding pulls PyTorch's HIP runtime in first."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from inclusivegan_amd import _abi\n'
            '_abi.get_plugin()\n'
            'import torch\n'
            'from inclusivegan_amd import hip_ops\n'
            'x = torch.randn(4, 64, device="cuda"); w = torch.randn(64, 32, device="cuda")\n'
            'y = hip_ops.matmul(x, w)\n'
            'torch.cuda.synchronize()\n'
            'print("ok", float((y - x @ w).abs().max()))\n') % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stderr[-2000:]
    assert float(r.stdout.split()[1]) < 1e-4


@pytest.mark.parametrize('case', [(3, 64, 33, 33, True, 3), (2, 128, 17, 9, False, 3), (2, 32, 9, 9, True, 1), (1, 8, 5, 7, True, 2)])
def test_fir_with_fused_epilogue_matches_two_step_form(case, cuda_device):
    """FirBanFn (the FIR after an up-convolution with noise + bias + activation in its store) against upfirdn_2d followed by
    BiasActNoiseFn -- each pinned to the oracle by its own tests: same values bit for bit (same arithmetic in the same order),
    same gradients."""
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import _setup_kernel, _simple_upfirdn_2d
    N, C, H, W, per_sample, act = case
    g = torch.Generator().manual_seed(C + H)
    dev = cuda_device
    k = _setup_kernel([1, 3, 3, 1]) * 4.0
    pad0, pad1 = 1, 1                                                   # the post-up-conv filter: out = in - 1
    x = torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.2).to(dev).requires_grad_(True)
    noise = torch.randn(N if per_sample else 1, 1, H - 1, W - 1, generator=g).to(dev)
    strength = torch.tensor(0.4, device=dev).requires_grad_(True)
    dy = torch.randn(N, C, H - 1, W - 1, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gain = float(np.sqrt(2)) if act == 3 else 1.0
    assert hip_ops.fir_ban_fusable(x, k, act)
    y1 = hip_ops.FirBanFn.apply(x, k, pad0, pad1, b, noise, strength, act, 0.2, gain)
    g1 = torch.autograd.grad(y1, [x, b, strength], dy)
    y0 = hip_ops.bias_act_noise(_simple_upfirdn_2d(x, k, pad0=pad0, pad1=pad1, data_format='NCHW'), b, noise, strength, act, 0.2, gain)
    g0 = torch.autograd.grad(y0, [x, b, strength], dy)
    assert tuple(y1.shape) == (N, C, H - 1, W - 1) and torch.equal(y1, y0)
    for a, r in zip(g1, g0):
        assert torch.equal(a, r)


def test_constant_filter_image_is_written_once_and_bit_identical(cuda_device):
    """ABI v9 (w_pieces): a weight marked as a constant of the run (the LPIPS network's filters, metrics/lpips.py) keeps ONE filter image per orientation on
    the tensor; the calls that use it return bit for bit what a call that images its filter itself returns (forward with the fused epilogue, data gradient),
    an in-place change of the weight drops the image, and a filter the piece form does not take gets none."""
    from inclusivegan_amd import hip_ops, _abi
    lib = _abi.get_plugin()
    dev = cuda_device
    g = torch.Generator().manual_seed(3)
    N, C, H = 8, 256, 16           # 2048 rows: the piece form
    x = to_nhwc_cuda(torch.randn(N, C, H, H, generator=g), dev)
    dy = to_nhwc_cuda(torch.randn(N, C, H, H, generator=g), dev)
    w = (torch.randn(3, 3, C, C, generator=g) / 48.0).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    plain = (hip_ops.conv2d_raw(x, w, geom, (H, H), C, bias=b, act=(2, 0.0, 1.0)), hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), C, w_transposed=True))
    assert getattr(w, '_igan_filter_images', None) is None
    wc = hip_ops.mark_constant(w.clone())
    for _ in range(2):
        got = (hip_ops.conv2d_raw(x, wc, geom, (H, H), C, bias=b, act=(2, 0.0, 1.0)), hip_ops.conv2d_raw(dy, wc, hip_ops.dgrad_geom(geom), (H, H), C, w_transposed=True))
        assert torch.equal(got[0], plain[0]) and torch.equal(got[1], plain[1])
    form = lib.igan_conv_piece_form()
    images = wc._igan_filter_images[1]
    assert len(images) == (2 if form != 0 else 0), (form, list(images))
    if form != 0:
        assert lib.igan_filter_image_bytes(3, 3, C, C) > 0 and lib.igan_filter_image_bytes(1, 1, C, C) == 0 and lib.igan_filter_image_bytes(3, 3, 3, 64) == 0
        first = next(iter(images.values()))
        with torch.no_grad():
            wc.mul_(2.0)                 # a new version of the weight: the images are rebuilt, the result follows the weight
        got = hip_ops.conv2d_raw(dy, wc, hip_ops.dgrad_geom(geom), (H, H), C, w_transposed=True)
        assert next(iter(wc._igan_filter_images[1].values())) is not first
        assert rel_err(got, plain[1] * 2.0) < 1e-6
        # a wrong-sized image is rejected by the library, never read
        p = _abi.Conv2DParams(x=x.data_ptr(), w=w.data_ptr(), y=plain[0].data_ptr(), in_scale=None, out_scale=None, workspace=None, workspace_floats=0, N=N, H=H, W=H, Cin=C, OH=H, OW=H,
                              Cout=C, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1, w_transposed=0, splits=1, alpha=1.0, bias=None, act=0, act_alpha=0.0, act_gain=1.0)
        import ctypes
        splits, sliced, wsf = ctypes.c_int(1), ctypes.c_int(0), ctypes.c_size_t(0)
        _abi.check(lib.igan_conv2d_plan(ctypes.byref(p), ctypes.byref(splits), ctypes.byref(sliced), ctypes.byref(wsf)))
        ws = torch.empty((wsf.value,), device=dev)
        p.workspace, p.workspace_floats, p.splits, p.sliced_tiles = ws.data_ptr(), wsf.value, splits.value, sliced.value
        p.w_pieces, p.w_pieces_bytes = first.data_ptr(), 12345
        assert lib.igan_conv2d(None, ctypes.byref(p)) != 0 and b'w_pieces' in lib.igan_last_error()
