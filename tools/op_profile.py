#!/usr/bin/env python3
"""Runs ONE training op's device work (loss + backward, eager) a few times, for rocprofv3 --stats.
usage: python tools/op_profile.py {G_train|G_reg|D_train|D_reg} [reps] [resolution] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd.dnnlib import tflib  # noqa: E402
from inclusivegan_amd.training import loss as L  # noqa: E402
from inclusivegan_amd.training.dataset import SyntheticDataset  # noqa: E402


def build(op, res=128, B=6):
    """Networks + inputs of one training op; returns a zero-argument function that runs its device work once."""
    dev = torch.device('cuda', 0)
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=8192, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=res, device=dev, seed=3)
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=dev)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    r1 = cl(torch.rand(B, 3, res, res, device=dev) * 2 - 1); r2 = cl(torch.rand(B, 3, res, res, device=dev) * 2 - 1)
    z1 = torch.nn.functional.normalize(torch.randn(B, 512, device=dev), dim=1); z2 = torch.nn.functional.normalize(torch.randn(B, 512, device=dev), dim=1)
    lab = torch.zeros(B, 0, device=dev); lab2 = torch.zeros(2 * B, 0, device=dev)
    reals = cl(torch.rand(2 * B, 3, res, res, device=dev) * 2 - 1)

    def run():
        if op.startswith('G'):
            D.requires_grad_(False)
            loss, reg = L.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, r1, lab, z1, r2, lab, z2, NN_rec_lpips_weight=2.5,
                                                                phase='loss' if op == 'G_train' else 'reg')
            v = loss if op == 'G_train' else reg * 4
            torch.autograd.grad(v.mean(), [p for p in G.trainables.values() if p.requires_grad], allow_unused=True)      # as Optimizer.differentiate
            D.requires_grad_(True)
        else:
            G.requires_grad_(False)
            loss, reg = L.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase='loss' if op == 'D_train' else 'reg')
            v = loss if op == 'D_train' else reg * 16
            torch.autograd.grad(v.mean(), [p for p in D.trainables.values() if p.requires_grad], allow_unused=True)
            G.requires_grad_(True)

    return run


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else 'G_train'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    run = build(op, res, B)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    print('%s: %.2f ms per call (eager wall)' % (op, (time.perf_counter() - t0) / reps * 1e3), flush=True)


if __name__ == '__main__':
    main()
