"""The other architectures the reference's build functions accept (G: orig / skip / resnet, D: orig / skip / resnet;
networks_stylegan2.py:326,446): forward + backward run on the HIP path, the grouped style path (all layers of a pass at
once) agrees with the per-layer one, and the validation-mode forward (truncation, fixed noise) works."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('g_arch,d_arch', [('orig', 'orig'), ('skip', 'resnet'), ('resnet', 'skip')])
def test_architectures_forward_backward(g_arch, d_arch, cuda_device):
    from inclusivegan_amd.dnnlib import tflib
    dev = cuda_device
    kw = dict(num_channels=3, resolution=32, label_size=0, fmap_base=512, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture=g_arch, seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture=d_arch, seed=2, **kw)
    z = torch.randn(6, 512, device=dev)
    lab = torch.zeros(6, 0, device=dev)
    img = G.get_output_for(z, lab, is_training=True)
    assert tuple(img.shape) == (6, 3, 32, 32) and bool(torch.isfinite(img).all())
    scores, _ = D.get_output_for(img, lab, is_training=True)
    torch.autograd.backward(torch.nn.functional.softplus(-scores).mean(), inputs=list(G.trainables.values()) + list(D.trainables.values()))
    assert float(G.flat_grads.abs().max()) > 0 and float(D.flat_grads.abs().max()) > 0
    assert bool(torch.isfinite(G.flat_grads).all()) and bool(torch.isfinite(D.flat_grads).all())
    with torch.no_grad():
        val = G.get_output_for(z, lab, is_validation=True, truncation_psi_val=0.7, randomize_noise=False)
    assert tuple(val.shape) == (6, 3, 32, 32) and bool(torch.isfinite(val).all())


def test_grouped_styles_equal_per_layer_styles(cuda_device):
    """IGAN_STYLE_GROUPED=0 (per-layer StyleModFn) and the default grouped launches give the same images and gradients
    (bit-identical: same kernels, same reduction order, only the launch grouping differs)."""
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from inclusivegan_amd.dnnlib import tflib
dev = torch.device('cuda', 0)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1,
                  num_channels=3, resolution=64, label_size=0, fmap_base=2048, device=dev)
torch.manual_seed(5)
z = torch.randn(6, 512, device=dev); lab = torch.zeros(6, 0, device=dev)
img = G.get_output_for(z, lab, is_training=True)
torch.autograd.backward((img * img).mean(), inputs=list(G.trainables.values()))
print('RESULT %%.9e %%.9e' %% (float(img.double().abs().sum()), float(G.flat_grads.double().abs().sum())))
''' % ROOT
    outs = []
    for flag in ('1', '0'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, IGAN_STYLE_GROUPED=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('RESULT')][0])
    assert outs[0] == outs[1], outs
