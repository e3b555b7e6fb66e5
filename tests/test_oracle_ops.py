"""CPU: the oracle pins itself (the reference has no tests for this path):
two independent restatements of upfirdn_2d agree, analytic known answers hold, the custom-gradient
parameters are the true adjoint (fp64), and the fused_bias_act derivative table equals autograd."""
import numpy as np
import pytest
import torch

from oracle import upfirdn_2d as U
from oracle import fused_bias_act as FB

PARAMS = [
    dict(), dict(padx0=1, padx1=1, pady0=1, pady1=1), dict(padx0=2, padx1=2, pady0=2, pady1=2),
    dict(upx=2, upy=2, padx0=2, padx1=1, pady0=2, pady1=1), dict(downx=2, downy=2, padx0=1, padx1=1, pady0=1, pady1=1),
    dict(upx=3, upy=2, downx=2, downy=3, padx0=2, padx1=3, pady0=1, pady1=0), dict(padx0=-1, padx1=2, pady0=0, pady1=-1),
    dict(upx=2, upy=1, downx=1, downy=2, padx0=0, padx1=1, pady0=3, pady1=0),
]


@pytest.mark.parametrize('kw', PARAMS)
@pytest.mark.parametrize('ksize', [(4, 4), (3, 3), (1, 1), (2, 5)])
def test_upfirdn_loops_equal_conv_formulation(kw, ksize):
    rng = np.random.RandomState(len(kw) * 13 + ksize[0])
    x = rng.randn(2, 7, 6, 3)
    k = rng.randn(*ksize)
    try:
        a = U.upfirdn_2d_loops(x, k, **kw)
    except AssertionError:
        pytest.skip('output smaller than 1x1')
    b = U.upfirdn_2d_ref(torch.from_numpy(x), k, **kw).numpy()
    assert a.shape == b.shape
    assert np.abs(a - b).max() < 1e-12


def test_known_answers():
    # constant image through the normalised FIR: interior == gain (filter_2d), upsample_2d of a constant == constant
    x = torch.ones(1, 2, 12, 12, dtype=torch.float64)
    y = U.filter_2d(x, [1, 3, 3, 1], gain=3.0)
    assert y.shape == x.shape and torch.allclose(y[:, :, 2:-2, 2:-2], torch.full((1, 2, 8, 8), 3.0, dtype=torch.float64))
    y = U.upsample_2d(x, [1, 3, 3, 1])
    assert y.shape == (1, 2, 24, 24) and torch.allclose(y[:, :, 2:-2, 2:-2], torch.ones(1, 2, 20, 20, dtype=torch.float64))
    y = U.downsample_2d(x, [1, 3, 3, 1])
    assert y.shape == (1, 2, 6, 6) and torch.allclose(y[:, :, 1:-1, 1:-1], torch.ones(1, 2, 4, 4, dtype=torch.float64))
    # identity filter, no resampling
    z = torch.randn(1, 5, 4, 3, dtype=torch.float64)
    assert torch.equal(U.upfirdn_2d_ref(z, np.ones((1, 1))), z)
    # out size formula (upfirdn_2d.py:119-120)
    y = U.upfirdn_2d_ref(torch.zeros(1, 9, 5, 1), np.ones((4, 3)), upx=2, upy=3, downx=3, downy=2, padx0=1, padx1=2, pady0=0, pady1=1)
    assert y.shape == (1, (9 * 3 + 0 + 1 - 4) // 2 + 1, (5 * 2 + 1 + 2 - 3) // 3 + 1, 1)


@pytest.mark.parametrize('kw', PARAMS)
def test_gradient_parameters_are_the_adjoint(kw):
    """<op(x), dy> == <x, op_grad(dy)> with the parameters of upfirdn_2d.py:123-128 (fp64)."""
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(2, 8, 7, 2)).requires_grad_(True)
    k = rng.randn(4, 4)
    full = dict(upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0); full.update(kw)
    y = U.upfirdn_2d_ref(x, k, **full)
    dy = torch.from_numpy(rng.randn(*y.shape))
    (gx,) = torch.autograd.grad(y, x, dy)
    gp = U.upfirdn_2d_grad_params(8, 7, k, **full)
    gk = gp.pop('k')
    gx2 = U.upfirdn_2d_ref(dy, gk, **gp)
    assert gx2.shape == gx.shape
    assert (gx - gx2).abs().max() < 1e-12


def test_upsample_conv_equals_direct_statement():
    """conv2d_transpose + flip/regroup (upfirdn_2d.py:286-291) == zero-insert, pad k-1, correlate with w."""
    import torch.nn.functional as F
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(2, 5, 6, 6)); w = torch.from_numpy(rng.randn(3, 3, 5, 7))
    y_ref = U.upsample_conv_2d(x, w, k=[1, 3, 3, 1])
    xu = torch.zeros(2, 5, 11, 11, dtype=torch.float64); xu[:, :, ::2, ::2] = x
    y = F.conv2d(F.pad(xu, [2, 2, 2, 2]), w.permute(3, 2, 0, 1))
    kf = U.setup_kernel([1, 3, 3, 1]) * 4
    y = U.simple_upfirdn_2d(y, kf, pad0=1, pad1=1)
    assert y_ref.shape == (2, 7, 12, 12)
    assert (y - y_ref).abs().max() < 1e-12


ACTS = list(FB.activation_funcs)


@pytest.mark.parametrize('act', ACTS)
def test_fused_bias_act_derivative_table(act):
    func, def_alpha, def_gain, idx, refkind, zero2 = FB.activation_funcs[act]
    rng = np.random.RandomState(idx)
    x = torch.from_numpy(rng.randn(5, 6)).requires_grad_(True)
    alpha = 0.0 if def_alpha is None else def_alpha
    y = FB.fused_bias_act(x, None, act=act)
    k0 = FB.fused_bias_act_kernel_ref(x.detach(), None, None, 0, idx, alpha, def_gain, 1).reshape(5, 6)
    assert (y.detach() - k0).abs().max() < 1e-12
    dy = torch.from_numpy(rng.randn(5, 6)).requires_grad_(True)
    (gx,) = torch.autograd.grad(y, x, dy, create_graph=True)
    ref = x.detach() if refkind == 'x' else y.detach()
    k1 = FB.fused_bias_act_kernel_ref(dy.detach(), None, ref, 1, idx, alpha, def_gain, 1).reshape(5, 6)
    assert (gx.detach() - k1).abs().max() < 1e-10
    # second order: d/dx <gx, v> = grad2 kernel applied to (v * dy)   (fused_bias_act.py:154-158)
    v = torch.from_numpy(rng.randn(5, 6))
    (g2,) = torch.autograd.grad(gx, x, v, allow_unused=True)
    k2 = FB.fused_bias_act_kernel_ref((v * dy.detach()), None, ref, 2, idx, alpha, def_gain, 1).reshape(5, 6)
    if g2 is None:
        g2 = torch.zeros_like(k2)
    # the CUDA table's grad=2 is d(dx)/dx per unit d_dx and carries one gain factor like grad=1
    assert (g2 - k2).abs().max() < 1e-9, act
    assert zero2 == bool(k2.abs().max() == 0)


def test_fused_bias_act_bias_axes():
    x = torch.randn(2, 3, 4, 5, dtype=torch.float64); b = torch.randn(3, dtype=torch.float64)
    y = FB.fused_bias_act(x, b, axis=1, act='lrelu')
    want = torch.nn.functional.leaky_relu(x + b.view(1, 3, 1, 1), 0.2) * np.sqrt(2)
    assert torch.allclose(y, want)
    k = FB.fused_bias_act_kernel_ref(x, b, None, 0, 3, 0.2, float(np.sqrt(2)), 20).reshape(x.shape)   # stepB = H*W
    assert torch.allclose(k, want)


def test_sampled_conv_oracle_matches_torch_autograd():
    """oracle/conv_sample.py (single elements by direct gather, used for the full-size GPU spot checks) equals torch's fp64
    conv2d / zero-insert + correlation and their autograd gradients on small problems of every geometry on the path."""
    from oracle import conv_sample as CS
    rng = np.random.RandomState(0)
    for (N, Cin, H, Cout, K, stride, up, pad, out) in [(2, 5, 8, 7, 3, 1, 1, 1, 8), (2, 4, 9, 6, 3, 2, 1, 0, 4), (2, 4, 5, 6, 3, 1, 2, 2, 11),
                                                        (2, 4, 7, 3, 1, 2, 1, 0, 4), (3, 6, 1, 5, 1, 1, 1, 0, 1)]:
        x = rng.randn(N, Cin, H, H); w = rng.randn(K, K, Cin, Cout); s = rng.rand(N, Cin) + .5; d = rng.rand(N, Cout) + .5
        dy = rng.randn(N, Cout, out, out)
        xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True)
        xs = xt * torch.tensor(s)[:, :, None, None]
        wk = wt.permute(3, 2, 0, 1)
        if up == 1:
            y = torch.nn.functional.conv2d(xs, wk, stride=stride, padding=pad)
        else:
            xu = torch.nn.functional.conv_transpose2d(xs, torch.eye(Cin, dtype=torch.float64)[:, :, None, None], stride=up)   # zero insertion
            y = torch.nn.functional.conv2d(xu, wk, padding=pad)
        y = y * torch.tensor(d)[:, :, None, None] * 0.7
        assert y.shape[-1] == out
        (y * torch.tensor(dy)).sum().backward()
        pick = lambda *dims: np.stack([rng.randint(m, size=24) for m in dims], 1)
        i = pick(N, Cout, out, out)
        assert np.allclose(CS.forward_samples(x, w, i, stride, up, pad, s, d, 0.7), y.detach().numpy()[tuple(i.T)], rtol=1e-11, atol=1e-11)
        i = pick(N, Cin, H, H)
        assert np.allclose(CS.dgrad_samples(dy, w, i, (H, H), stride, up, pad, s, d, 0.7), xt.grad.numpy()[tuple(i.T)], rtol=1e-11, atol=1e-11)
        i = pick(K, K, Cin, Cout)
        assert np.allclose(CS.wgrad_samples(x, dy, i, stride, up, pad, s, d, 0.7), wt.grad.numpy()[tuple(i.T)], rtol=1e-11, atol=1e-11)
