"""Eager ops with every torch.empty / empty_like NaN-filled: does any kernel read memory it was supposed to have written?"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
_empty, _empty_like = torch.empty, torch.empty_like
FILL = [True]
def empty(*a, **k):
    t = _empty(*a, **k)
    if FILL[0] and t.is_cuda and t.dtype.is_floating_point: t.fill_(float('nan'))
    return t
def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if FILL[0] and t.is_cuda and t.dtype.is_floating_point: t.fill_(float('nan'))
    return t
torch.empty, torch.empty_like = empty, empty_like
from inclusivegan_amd.dnnlib import tflib
from inclusivegan_amd.dnnlib.tflib import tfutil
from inclusivegan_amd.training import loss as PL
from inclusivegan_amd.training.dataset import SyntheticDataset
dev = torch.device('cuda', 0)
RES, FMAP, B = 32, int(os.environ.get('FMAP', '1024')), 6
kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=12, **kw)
lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=RES, device=dev, seed=13)
ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
lab = torch.zeros(B, 0, device=dev); lab2 = torch.zeros(2 * B, 0, device=dev)
r1 = cl(torch.rand(B, 3, RES, RES, device=dev)); r2 = cl(torch.rand(B, 3, RES, RES, device=dev)); z1 = torch.randn(B, 512, device=dev); z2 = torch.randn(B, 512, device=dev)
reals = cl(torch.rand(2 * B, 3, RES, RES, device=dev))
for name in ('G', 'G_reg', 'D', 'D_reg'):
    out = {}
    for fill in (False, True):
        FILL[0] = fill
        torch.manual_seed(5)
        G.zero_grad(); D.zero_grad(); G.pl_mean_var = torch.zeros((), device=dev)
        with torch.no_grad(): G.vars['dlatent_avg'].zero_()
        if name in ('G', 'G_reg'):
            D.requires_grad_(False)
            loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, r1, lab, z1, r2, lab, z2, NN_rec_lpips_weight=2.5, phase='loss' if name == 'G' else 'reg')
            v = loss if name == 'G' else reg
            grads = torch.autograd.grad(v.mean(), list(G.trainables.values()), allow_unused=True)
            D.requires_grad_(True); names = list(G.trainables)
        else:
            G.requires_grad_(False)
            loss, reg = PL.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase='loss' if name == 'D' else 'reg')
            v = loss if name == 'D' else reg
            grads = torch.autograd.grad(v.mean(), list(D.trainables.values()), allow_unused=True)
            G.requires_grad_(True); names = list(D.trainables)
        torch.cuda.synchronize()
        out[fill] = (v.detach().clone(), [None if g is None else g.detach().clone() for g in grads])
    FILL[0] = False
    v0, g0 = out[False]; v1, g1 = out[True]
    bad = [(n, int(torch.isnan(b).sum()), b.numel()) for n, a, b in zip(names, g0, g1) if b is not None and (torch.isnan(b).any() or not torch.equal(a, b))]
    print('%-6s value plain %.8g nanfill %.8g | %d of %d gradients change when fresh memory is NaN' % (name, float(v0.mean()), float(v1.mean()), len(bad), len(names)), flush=True)
    for n, k, m in bad[:40]:
        print('       %-50s nan %d of %d' % (n, k, m))
