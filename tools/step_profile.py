#!/usr/bin/env python3
"""Times the pieces of one training iteration on the GPU (wall clock with synchronisation) so that a
rocprofv3 --kernel-trace --stats run of this script attributes time to kernels.
usage: python tools/step_profile.py [resolution] [batch] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd.dnnlib import tflib  # noqa: E402
from inclusivegan_amd.metrics import lpips as LP  # noqa: E402


def timeit(name, fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%-28s %9.2f ms' % (name, dt * 1e3), flush=True)
    return dt


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda', 0)
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=8192, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    L = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=res, device=dev, seed=3)
    z = torch.randn(B, 512, device=dev)
    lab = torch.zeros(B, 0, device=dev)
    img = torch.randn(2 * B, 3, res, res, device=dev).contiguous(memory_format=torch.channels_last)

    def g_fwd():
        with torch.no_grad():
            return G.get_output_for(z, lab, is_training=True)

    def g_fwd_bwd():
        G.zero_grad()
        out = G.get_output_for(z, lab, is_training=True)
        out.sum().backward()

    def d_fwd_bwd():
        D.zero_grad()
        s, _ = D.get_output_for(img, torch.zeros(2 * B, 0, device=dev), is_training=True)
        s.sum().backward()

    def lpips_fwd_bwd():
        a = (img[:B] * 127.5 + 127.5).requires_grad_(True)
        f = LP.features_of(L, a)
        with torch.no_grad():
            g = LP.features_of(L, img[B:] * 127.5 + 127.5)
        LP.distance_of(L, f, g).sum().backward()

    def d_r1():
        D.zero_grad()
        x = img.detach().requires_grad_(True)
        s, _ = D.get_output_for(x, torch.zeros(2 * B, 0, device=dev), is_training=True)
        (g,) = torch.autograd.grad(s.sum(), [x], create_graph=True)
        (g * g).sum().backward()

    def g_pl():
        G.zero_grad()
        out, dl = G.get_output_for(z[:max(B // 2, 1)], lab[:max(B // 2, 1)], is_training=True, return_dlatents=True)
        (g,) = torch.autograd.grad((out * torch.randn_like(out)).sum(), [dl], create_graph=True)
        (g * g).sum().sqrt().backward()

    print('resolution %d batch %d' % (res, B), flush=True)
    timeit('G forward (no grad)', g_fwd, reps)
    timeit('G forward+backward', g_fwd_bwd, reps)
    timeit('D forward+backward (2B)', d_fwd_bwd, reps)
    timeit('LPIPS fwd(B)+fwd/bwd(B)', lpips_fwd_bwd, reps)
    timeit('D R1 reg (2B)', d_r1, 1)
    timeit('G path-length reg (B/2)', g_pl, 1)


if __name__ == '__main__':
    main()
