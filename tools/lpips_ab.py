#!/usr/bin/env python3
"""A/B of the G loss gradients with the fused LPIPS path (pair tables + pool/tap) against the per-pair form: relative L2 difference
of every gradient tensor.  usage: python tools/lpips_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd.dnnlib import tflib  # noqa: E402
from inclusivegan_amd.dnnlib.tflib import tfutil  # noqa: E402
from inclusivegan_amd.metrics import lpips as LP  # noqa: E402
from inclusivegan_amd.training import loss as L  # noqa: E402
from inclusivegan_amd.training.dataset import SyntheticDataset  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    res, B = 32, 6
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=1024, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=res, device=dev, seed=3)
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=dev)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    g = torch.Generator(device='cpu').manual_seed(5)
    r1 = cl((torch.rand(B, 3, res, res, generator=g) * 2 - 1).to(dev)); r2 = cl((torch.rand(B, 3, res, res, generator=g) * 2 - 1).to(dev))
    z1 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1).to(dev); z2 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1).to(dev)
    lab = torch.zeros(B, 0, device=dev)
    D.requires_grad_(False)
    params = [p for p in G.trainables.values()]
    out = {}
    tape = None
    for fused in (True, False):
        LP._FUSED = fused
        if tape is None:
            rec = tfutil.RecordingRandom()
            ctx = tfutil.use_random(rec)
        else:
            ctx = tfutil.use_random(tfutil.RandomTape(tape))
        state = {n: v.detach().clone() for n, v in G.vars.items() if not v.requires_grad}
        with ctx:
            loss, _ = L.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, r1, lab, z1, r2, lab, z2, NN_rec_lpips_weight=2.5, phase='loss')
        if tape is None:
            tape = rec.entries
        grads = torch.autograd.grad(loss.mean(), params, allow_unused=True)
        out[fused] = (loss.detach().clone(), grads)
        with torch.no_grad():
            for n, v in G.vars.items():
                if not v.requires_grad:
                    v.copy_(state[n])
    la, ga = out[True]; lb, gb = out[False]
    print('loss diff', float((la - lb).abs().max()), 'of', float(lb.abs().max()))
    worst = 0.0
    for (name, _), a, b in zip(G.trainables.items(), ga, gb):
        if a is None or b is None:
            assert a is None and b is None, name
            continue
        e = float((a - b).norm() / (b.norm() + 1e-30))
        worst = max(worst, e)
        if e > 1e-5:
            print('  %-50s %.3e  (|g| %.3e)' % (name, e, float(b.norm())))
    print('worst relative L2 difference %.3e' % worst)


if __name__ == '__main__':
    main()
