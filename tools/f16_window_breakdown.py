#!/usr/bin/env python3
"""Which tensors hold the elements the two-piece fp16 form images BELOW its exact window (more than 2^26 below the tensor's largest magnitude;
csrc/conv2d_mfma.hip), and how far below?  Runs the eager device work of each training op once at the bench size
(tools/op_profile.py: random-init config-e, 128x128, minibatch 6) with hip_ops.to_pieces / conv2d_raw / conv2d_wgrad_raw wrapped: for every tensor that
gets a piece image (shared images and the ones the library makes itself) the statistics are taken with torch ops on the same values (x * scale).
usage: python tools/f16_window_breakdown.py [ops = G_train G_reg D_train D_reg]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402
import op_profile  # noqa: E402

stats = defaultdict(lambda: [0, 0, 0, 0, 0, 0, 0.0])      # key -> [tensors, elements, nonzero, below 2^-26, below 2^-32, below 2^-38, smallest nonzero ratio (log2)]
current = ['?']


def caller():
    f = sys._getframe(2)
    while f is not None:
        slf = f.f_locals.get('ctx', None)
        name = f.f_code.co_name
        qual = f.f_code.co_qualname if hasattr(f.f_code, 'co_qualname') else name
        if 'Fn' in qual or name in ('forward', 'backward'):
            cls = f.f_globals.get('__name__', '')
            return qual
        f = f.f_back
    return '?'


def account(kind, x, scale):
    if x.is_meta or not x.is_cuda:      # the networks' template pass
        return
    with torch.no_grad():
        v = x.detach().float()
        if scale is not None:
            v = v * scale.detach().float()[:, :, None, None]
        a = v.abs()
        amax = float(a.max())
        if amax == 0.0:
            return
        nz = a > 0
        r = a / amax
        key = (current[0], kind, caller(), tuple(x.shape))
        s = stats[key]
        s[0] += 1; s[1] += v.numel(); s[2] += int(nz.sum())
        s[3] += int((nz & (r < 2.0 ** -26)).sum()); s[4] += int((nz & (r < 2.0 ** -32)).sum()); s[5] += int((nz & (r < 2.0 ** -38)).sum())
        s[6] = min(s[6], float(torch.log2(r[nz].min()))) if int(nz.sum()) else s[6]


_to_pieces, _conv, _wgrad = hip_ops.to_pieces, hip_ops.conv2d_raw, hip_ops.conv2d_wgrad_raw


def to_pieces(x, scale=None):
    out = _to_pieces(x, scale)
    if out is not None:
        account('shared image', x, scale)
    return out


def conv2d_raw(x, w, geom, out_hw, cout, w_transposed=False, in_scale=None, x_pieces=None, **kw):
    if x_pieces is None and x.dim() == 4 and not x.is_meta and hip_ops.pieces_wanted(geom, x.shape[1], cout) and to_pieces_ok(x):
        account('library image', x, in_scale)
    return _conv(x, w, geom, out_hw, cout, w_transposed=w_transposed, in_scale=in_scale, x_pieces=x_pieces, **kw)


def to_pieces_ok(x):
    from inclusivegan_amd import _abi
    n, c, h, w = x.shape
    return bool(_abi.get_plugin().igan_pieces_image_ok(int(n), int(h * w), int(c)))


hip_ops.to_pieces = to_pieces
hip_ops.conv2d_raw = conv2d_raw


def main():
    ops = sys.argv[1:] or ['G_train', 'G_reg', 'D_train', 'D_reg']
    for op in ops:
        current[0] = op
        run = op_profile.build(op)
        run()
        torch.cuda.synchronize()
    tot = [0, 0, 0, 0, 0]
    print('%-8s %-14s %-34s %-22s %5s %12s %12s %10s %10s %10s %8s' % ('op', 'image', 'made in', 'shape', 'n', 'elements', 'non-zero', '< 2^-26', '< 2^-32', '< 2^-38', 'min log2'))
    for key, s in sorted(stats.items(), key=lambda kv: -kv[1][3]):
        for i in range(5):
            tot[i] += s[1 + i]
        if s[3] == 0:
            continue
        print('%-8s %-14s %-34s %-22s %5d %12d %12d %10d %10d %10d %8.1f' % (key[0], key[1], key[2][:34], 'x'.join(map(str, key[3])), s[0], s[1], s[2], s[3], s[4], s[5], s[6]))
    print('all %d tensor kinds: %d elements, %d non-zero, below the window (ratio to the tensor maximum < 2^-26): %d = %.3g of all; < 2^-32: %d; < 2^-38: %d' % (
        len(stats), tot[0], tot[1], tot[2], tot[2] / max(tot[0], 1), tot[3], tot[4]))
    clean = sum(1 for s in stats.values() if s[3] == 0)
    print('%d of %d tensor kinds have no element below the window' % (clean, len(stats)))


if __name__ == '__main__':
    main()
