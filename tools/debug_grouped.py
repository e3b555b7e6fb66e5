#!/usr/bin/env python3
"""Which conv kernels a small G forward + backward launches, and digests of every conv output (debugging aid for
tests/test_gpu_archs.py::test_grouped_styles_equal_per_layer_styles):  IGAN_STYLE_GROUPED=0/1 python tools/debug_grouped.py"""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from inclusivegan_amd import hip_ops  # noqa: E402
from inclusivegan_amd.dnnlib import tflib  # noqa: E402

dev = torch.device('cuda', 0)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1,
                  num_channels=3, resolution=64, label_size=0, fmap_base=2048, device=dev)
orig = hip_ops.conv2d_raw
origw = hip_ops.conv2d_wgrad_raw
log = []


def dig(t):
    return hashlib.sha1(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]


def wrapped(x, w, geom, out_hw, cout, **kw):
    y = orig(x, w, geom, out_hw, cout, **kw)
    log.append('conv %s -> %s  wt %s  s %s d %s xp %s  x %s w %s y %s' % (tuple(x.shape), tuple(y.shape), kw.get('w_transposed', False),
               None if kw.get('in_scale') is None else (dig(kw['in_scale']), kw['in_scale'].data_ptr() & 15, kw['in_scale'].is_contiguous()),
               None if kw.get('out_scale') is None else (dig(kw['out_scale']), kw['out_scale'].data_ptr() & 15),
               kw.get('x_pieces') is not None, dig(x), dig(w), dig(y)))
    return y


def wrappedw(x, dy, geom, **kw):
    dw = origw(x, dy, geom, **kw)
    log.append('wgrad %s %s xp %s dyp %s -> %s' % (tuple(x.shape), tuple(dy.shape), kw.get('x_pieces') is not None, kw.get('dy_pieces') is not None, dig(dw)))
    return dw


hip_ops.conv2d_raw = wrapped
hip_ops.conv2d_wgrad_raw = wrappedw
torch.manual_seed(5)
z = torch.randn(6, 512, device=dev); lab = torch.zeros(6, 0, device=dev)
img = G.get_output_for(z, lab, is_training=True)
torch.autograd.backward((img * img).mean(), inputs=list(G.trainables.values()))
print('\n'.join(log))
print('RESULT %.9e %.9e' % (float(img.double().abs().sum()), float(G.flat_grads.double().abs().sum())))
