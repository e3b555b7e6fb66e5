"""GPU: the HIP path, through the product's own functions, against golden vectors produced by EXECUTING the reference's Python
(tests/golden/make_ref_ops_golden.py: dnnlib/tflib/ops/upfirdn_2d.py, fused_bias_act.py, training/networks_stylegan2.py run with a
NumPy stand-in for the TensorFlow primitives; fixture tests/golden/ref_ops_golden.npz).  The goldens are float64; the HIP path is
fp32: tolerances 1e-5 (streaming kernels), 1e-4 (whole networks), relative max-abs."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_ops_golden.npz'), allow_pickle=False)


def dev32(a, dev):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize('i', range(int(G['upfirdn_cases'])))
def test_upfirdn_kernel_and_gradients(cuda_device, i):
    """UpFirDn2D on the HIP kernels: y, dx = the gradient the reference defines for the op, and the gradient of that."""
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    p = 'upfirdn_%d_' % i
    upx, upy, downx, downy, px0, px1, py0, py1 = [int(v) for v in G[p + 'params']]
    kw = dict(upx=upx, upy=upy, downx=downx, downy=downy, padx0=px0, padx1=px1, pady0=py0, pady1=py1)
    x = dev32(G[p + 'x'], cuda_device).requires_grad_(True)
    y = upfirdn_2d(x, G[p + 'k'], **kw)
    assert rel_err(y, G[p + 'y']) < 1e-5
    dy = dev32(G[p + 'dy'], cuda_device).requires_grad_(True)
    dx, = torch.autograd.grad(y, x, dy, create_graph=True)
    assert rel_err(dx, G[p + 'dx']) < 1e-5
    d_dy, = torch.autograd.grad(dx, dy, dev32(G[p + 'ddx'], cuda_device))
    assert rel_err(d_dy, G[p + 'd_dy']) < 1e-5


def test_resampling_wrappers(cuda_device):
    from inclusivegan_amd.dnnlib.tflib.ops import upfirdn_2d as PU
    x = dev32(G['wrapper_x'], cuda_device).contiguous(memory_format=torch.channels_last)
    done = 0
    for p in G['wrapper_cases']:
        name = p[:p.rindex('_', 0, -1)]
        k_in = G[p + 'k_in'].tolist() or None
        factor, gain = int(G[p + 'factor_gain'][0]), float(G[p + 'factor_gain'][1])
        if name in ('upsample_conv_2d', 'conv_downsample_2d'):
            y = getattr(PU, name)(x, dev32(G[p + 'w'], cuda_device), k=k_in, factor=factor, gain=gain)
        elif name == 'filter_2d':
            y = PU.filter_2d(x, k_in, gain=gain)
        else:
            y = getattr(PU, name)(x, k=k_in, factor=factor, gain=gain)
        assert rel_err(y, G[p + 'y']) < 1e-5, p
        done += 1
    assert done == len(G['wrapper_cases'])


def test_fused_bias_act_all_activations(cuda_device):
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act
    x, b = dev32(G['act_x'], cuda_device), dev32(G['act_b'], cuda_device)
    for n in [str(v) for v in G['act_names']]:
        assert rel_err(fused_bias_act(x, b, act=n), G['act_%s_default' % n]) < 1e-5, n
        assert rel_err(fused_bias_act(x, b, act=n, alpha=0.3, gain=0.7), G['act_%s_custom' % n]) < 1e-5, n
        assert rel_err(fused_bias_act(x, None, axis=3, act=n), G['act_%s_nobias_axis3' % n]) < 1e-5, n


def _network(kind, arch, dev, prefix, **extra):
    from inclusivegan_amd.dnnlib import tflib
    res, fmap = int(G['net_cfg'][0]), int(G['net_cfg'][1])
    fn = 'inclusivegan_amd.training.networks_stylegan2.' + ('G_main' if kind == 'G' else 'D_stylegan2_feature')
    net = tflib.Network(kind, func_name=fn, num_channels=3, resolution=res, label_size=0, fmap_base=fmap, architecture=arch, device=dev, seed=1, **extra)
    with torch.no_grad():
        for name, v in net.vars.items():
            v.copy_(dev32(G[prefix + name.replace('/', '.')], dev).reshape(v.shape))
    return net


@pytest.mark.parametrize('p', [str(c) for c in G['G_cases']])
def test_generator_against_reference_execution(cuda_device, p):
    """G_main on the HIP path with the reference's variables and the reference run's random draws: images, dlatents and the
    dlatent_avg update, for every architecture and mode the golden holds."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    _, arch, mode, fused, _ = p.split('_')
    res, fmap, latent, dlatent, mfmaps = [int(v) for v in G['net_cfg']]
    net = _network('G', arch, cuda_device, 'Gparam_%s.' % arch, latent_size=latent, dlatent_size=dlatent, mapping_fmaps=mfmaps)
    kw = dict(train=dict(is_training=True), val=dict(is_validation=True, truncation_psi_val=0.7, truncation_cutoff_val=4), plain=dict(truncation_psi=0.5, randomize_noise=False))[mode]
    kinds = [str(k) for k in G[p + 'tape_kinds']]
    tape = tfutil.RandomTape([(k, G['%stape_%03d' % (p, i)]) for i, k in enumerate(kinds)])
    z = dev32(G[p + 'z'], cuda_device)
    with tfutil.use_random(tape), torch.no_grad():
        img, dl = net.get_output_for(z, torch.zeros(z.shape[0], 0, device=cuda_device), return_dlatents=True, fused_modconv=bool(int(fused)), **kw)
    assert tape.pos == len(kinds), 'the HIP path drew %d of the %d tensors the reference run drew' % (tape.pos, len(kinds))
    assert rel_err(dl, G[p + 'dlatents']) < 1e-5
    assert rel_err(img, G[p + 'img']) < 1e-4
    assert rel_err(net.vars['dlatent_avg'], G[p + 'dlatent_avg_after']) < 1e-5


@pytest.mark.parametrize('arch', ['skip', 'resnet', 'orig'])
def test_discriminator_against_reference_execution(cuda_device, arch):
    net = _network('D', arch, cuda_device, 'Dparam_%s.' % arch)
    x = dev32(G['D_%s_x' % arch], cuda_device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        s, f = net.get_output_for(x, torch.zeros(x.shape[0], 0, device=cuda_device), is_training=True, return_features=True)
    assert rel_err(s, G['D_%s_scores' % arch]) < 1e-4 and rel_err(f, G['D_%s_features' % arch]) < 1e-4


# ---- the half instantiations of the two custom ops (upfirdn_2d.cu:323-324, fused_bias_act.cu:185-186) -------------------------

@pytest.mark.parametrize('i', range(int(G['upfirdn_cases'])))
def test_upfirdn_half_instantiation(cuda_device, i):
    """T = half: inputs, taps and outputs in binary16, float accumulation (upfirdn_2d.cu:101,114).  Expected = the float64 oracle on
    the half-rounded inputs and taps, rounded to half; 1.5 half ulps of the largest output.  Gradient through autograd too."""
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    from oracle import upfirdn_2d as OU
    p = 'upfirdn_%d_' % i
    upx, upy, downx, downy, px0, px1, py0, py1 = [int(v) for v in G[p + 'params']]
    kw = dict(upx=upx, upy=upy, downx=downx, downy=downy, padx0=px0, padx1=px1, pady0=py0, pady1=py1)
    xh = torch.from_numpy(G[p + 'x']).to(torch.float16)
    kh = G[p + 'k'].astype(np.float16).astype(np.float64)
    want = OU.upfirdn_2d_ref(xh.double(), kh, **kw)
    x = xh.to(cuda_device).requires_grad_(True)
    y = upfirdn_2d(x, G[p + 'k'], **kw)
    assert y.dtype == torch.float16
    tol = 1.5 * 2.0 ** -11 * float(want.abs().max())
    assert float((y.detach().double().cpu() - want).abs().max()) <= tol
    dyh = torch.from_numpy(G[p + 'dy']).to(torch.float16)
    dx, = torch.autograd.grad(y, x, dyh.to(cuda_device))
    gp = OU.upfirdn_2d_grad_params(xh.shape[1], xh.shape[2], kh, **kw)
    want_dx = OU.upfirdn_2d_ref(dyh.double(), gp['k'], **{n: gp[n] for n in kw})
    assert dx.dtype == torch.float16 and float((dx.double().cpu() - want_dx).abs().max()) <= 1.5 * 2.0 ** -11 * float(want_dx.abs().max())


def test_fused_bias_act_half_instantiation(cuda_device):
    """T = half for every activation and grad = 0 / 1 / 2 through the C ABI wrapper: each element is widened to float, evaluated with
    the float table and rounded on store (fused_bias_act.cu:56-61,113) -- expected = the oracle's restatement of that table in
    float on the half-rounded inputs, rounded to half (one half ulp)."""
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act, activation_funcs
    from oracle import fused_bias_act as OF
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 8, 5, 5, generator=g) * 2).to(torch.float16)
    b = torch.randn(8, generator=g).to(torch.float16)
    ref = torch.rand(3, 8, 5, 5, generator=g).to(torch.float16)
    for name, spec in activation_funcs.items():
        for grad in (0, 1, 2):
            got = hip_ops.fused_bias_act_raw(x.to(cuda_device), b.to(cuda_device), None if grad == 0 else ref.to(cuda_device), grad, spec.hip_idx, 0.2, 1.3, 8, 25)
            want = OF.fused_bias_act_kernel_ref(x.float(), b.float(), None if grad == 0 else ref.float(), grad, spec.hip_idx, 0.2, 1.3, 25).reshape(x.shape)
            assert got.dtype == torch.float16
            err = (got.float().cpu() - want).abs()
            assert bool((err <= 2.0 ** -10 * want.abs() + 1e-4).all()), (name, grad, float(err.max()))
    # the operator surface on half tensors, with gradients (lrelu: the generic Function; piecewise-linear fp32 fast path is not taken)
    xg = x.to(cuda_device).requires_grad_(True)
    bg = b.to(cuda_device).requires_grad_(True)
    y = fused_bias_act(xg, bg, act='lrelu')
    y.float().sum().backward()
    yo = OF.fused_bias_act(x.double(), b.double(), act='lrelu')
    assert y.dtype == torch.float16 and float((y.detach().double().cpu() - yo).abs().max()) <= 2.0 ** -10 * float(yo.abs().max())
    assert xg.grad.dtype == torch.float16 and bg.grad.shape == (8,)
    want_db = torch.where(x.double() + b.double().view(1, 8, 1, 1) > 0, 1.0, 0.2).mul(np.sqrt(2)).sum(dim=(0, 2, 3))
    assert float((bg.grad.double().cpu() - want_db).abs().max()) <= 2e-2 * float(want_db.abs().max())
    with pytest.raises(TypeError):
        hip_ops.fused_bias_act_raw(x.to(cuda_device), b.float().to(cuda_device), None, 0, 3, 0.2, 1.0, 8, 25)      # mixed types
