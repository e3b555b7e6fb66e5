#!/usr/bin/env python3
"""Rounding error of the forward convolution against fp64 (CPU), for the product path and the measure-only bf16-piece variants
(run once per setting of IGAN_CONV_BF16X3 / IGAN_LIB: the switches are read once per process).  Prints relative L2 and the
signed mean error (a bias shows as a mean far from zero relative to the L2 error) for zero-mean and for positive inputs.
usage: [IGAN_CONV_BF16X3=6] [IGAN_LIB=...] python tools/split_accuracy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    tag = 'BF16X3=%s LIB=%s' % (os.environ.get('IGAN_CONV_BF16X3', '0'), os.path.basename(os.environ.get('IGAN_LIB', 'product')))
    g = torch.Generator().manual_seed(1)
    for (N, C, H, kind) in ((2, 512, 32, 'normal'), (2, 512, 32, 'lrelu'), (2, 512, 32, 'positive'), (2, 128, 64, 'normal')):
        x = torch.randn(N, C, H, H, generator=g)
        if kind == 'lrelu':
            x = torch.nn.functional.leaky_relu(x, 0.2) * 2 ** 0.5
        if kind == 'positive':
            x = x.abs()
        w = torch.randn(3, 3, C, C, generator=g) / (9 * C) ** 0.5
        if kind == 'positive':
            w = w.abs()
        want = torch.nn.functional.conv2d(x.double(), w.double().permute(3, 2, 0, 1), padding=1)
        geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        got = hip_ops.conv2d_raw(xd, w.to(dev), geom, (H, H), C).double().cpu()
        err = got - want
        rms = float(want.pow(2).mean().sqrt())
        print('%-34s N%d C%d %dx%d %-8s rel L2 %.3e   mean err / rms %+.3e   max|err| / rms %.3e' % (
            tag, N, C, H, H, kind, float(err.pow(2).mean().sqrt()) / rms, float(err.mean()) / rms, float(err.abs().max()) / rms))


if __name__ == '__main__':
    main()
