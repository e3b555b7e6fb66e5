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
