#!/usr/bin/env python3
"""SHA-1 digests of the piece-form convolution family (forward, data gradient, weight gradient) on a few large shapes, for bit-equality
A/Bs of library builds whose arithmetic must not change (e.g. a re-scheduled step):
    python tools/planes_digest.py > a.txt; IGAN_LIB=inclusivegan_amd/csrc/libigan_hip_x.so python tools/planes_digest.py > b.txt; diff a.txt b.txt"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402

dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(5)


def dig(t):
    return hashlib.sha1(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


for name, N, Cin, H, Cout, stride, up, pad, out in [('32x32 C512', 24, 512, 32, 512, 1, 1, 1, 32), ('128x128 C128', 6, 128, 128, 128, 1, 1, 1, 128),
                                                    ('up 32->65 C512->256', 8, 512, 32, 256, 1, 2, 2, 65), ('s2 65->32 C256->512', 8, 256, 65, 512, 2, 1, 0, 32),
                                                    ('19x19 C160->224 ragged', 6, 160, 19, 224, 1, 1, 1, 19)]:
    x = torch.randn(N, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(3, 3, Cin, Cout, generator=g) / (9 * Cin) ** 0.5).to(dev)
    s = (torch.rand(N, Cin, generator=g) + 0.5).to(dev)
    d = (torch.rand(N, Cout, generator=g) + 0.5).to(dev)
    geom = hip_ops.ConvGeom(3, 3, stride, up, pad, pad)
    y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d)
    dy = torch.randn(N, Cout, out, out, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dx = hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True, in_scale=d)
    dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d)
    print('%-26s fwd %s dgrad %s wgrad %s' % (name, dig(y), dig(dx), dig(dw)))
