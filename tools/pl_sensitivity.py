#!/usr/bin/env python3
"""CPU experiment (fp64 oracle, no GPU): how strongly do the path-length step's gradients react to a relative error of size eps in the OUTPUTS of the
modulated convolutions of one resolution?  (Round 5: with the fp16 form's row threshold at 1024 the 16x16 layers of BASELINE config 2 ran on the piece
kernels and the path-length step's gradients moved 4-15x further from the oracle, while every single call stayed at 1e-7 of fp64 -- profiles/r05_small_layers.txt
section 5.)  The network is config 2's generator at random initialisation with pl_mean set to a fraction of the batch's mean path length (a trained state:
pl_mean tracks the lengths, so the penalty (pl_lengths - pl_mean)^2 is a difference of nearly equal numbers); the perturbation is y -> y * (1 + eps * xi), xi ~ N(0, 1)
fixed, applied in the forward pass of every modulated convolution whose output is `res` x `res`; reported: the largest per-variable relative deviation of the
gradient (the quantity tests/test_gpu_loop_parity.py bounds by 5e-3) divided by eps.
usage: python tools/pl_sensitivity.py [pl_mean fraction = 0.98] [eps = 1e-6] [resolution = 32] [minibatch = 6]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import loss as OL  # noqa: E402
from oracle import networks_stylegan2 as N  # noqa: E402
from oracle.misc import SeededRandom  # noqa: E402
from inclusivegan_amd.dnnlib import tflib  # noqa: E402


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.98
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=8192, device='cpu')
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    gp = {n: v.detach().double().clone().requires_grad_(v.requires_grad) for n, v in G.vars.items()}
    dp = {n: v.detach().double() for n, v in D.vars.items()}
    cfg = dict(resolution=res, num_channels=3, fmap_base=8192, G_arch='skip', D_arch='resnet', fused_modconv=False)
    names = [n for n, p in gp.items() if p.requires_grad]
    gtr = [gp[n] for n in names]
    gen = torch.Generator().manual_seed(1)
    lat = torch.nn.functional.normalize(torch.randn(batch, 512, generator=gen), dim=1).double()
    orig = N.modulated_conv2d_layer
    target = dict(res=None)
    noise = {}

    def patched(sc, x, y, fmaps, kernel, **kwargs):
        out = orig(sc, x, y, fmaps, kernel, **kwargs)
        if target['res'] is not None and out.shape[2] == target['res'] and kernel == 3:
            key = (sc.prefix if hasattr(sc, 'prefix') else id(sc), tuple(out.shape))
            if key not in noise:
                noise[key] = torch.randn(out.shape, generator=torch.Generator().manual_seed(len(noise) + 7), dtype=out.dtype)
            out = out * (1.0 + eps * noise[key])
        return out

    N.modulated_conv2d_layer = patched

    def greg(pl_mean):
        state = {'pl_mean': torch.tensor(pl_mean, dtype=torch.float64)}
        _, reg, terms = OL.G_loss(gp, dp, None, cfg, SeededRandom(0), batch, None, lat, None, lat, 2.5, phase='reg', state=state)
        g = torch.autograd.grad((reg * 4).mean(), gtr, allow_unused=True)
        return reg.detach(), g, state

    t0 = time.time()
    reg0, _, st = greg(0.0)
    mean_len = float(st['pl_mean']) / 0.01          # pl_mean = 0 + pl_decay * mean(pl_lengths)
    pm = frac * mean_len
    print('mean path length %.6g; pl_mean set to %.3f of it (%.1f s per evaluation)' % (mean_len, frac, time.time() - t0))
    _, g0, _ = greg(pm)
    for r in [x for x in (8, 16, 32, 64, 128) if x <= res]:
        target['res'] = r
        noise.clear()
        reg1, g1, _ = greg(pm)
        worst, which = 0.0, None
        for n, a, b in zip(names, g0, g1):
            if a is None or a.numel() == 1 or not bool(a.abs().sum() > 0):
                continue
            e = float((b - a).norm() / a.norm())
            if e > worst:
                worst, which = e, n
        print('relative error %.0e in the 3x3 modulated convolutions with %2dx%-2d outputs -> largest gradient deviation %.3e (%s) = %.0f x eps' % (eps, r, r, worst, which, worst / eps), flush=True)
    target['res'] = None


if __name__ == '__main__':
    main()
