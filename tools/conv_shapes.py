#!/usr/bin/env python3
"""Per-shape accounting of the conv family over one training iteration: every conv / dgrad / wgrad launch of
the four training ops is timed with HIP events, grouped by (kind, shape), weighted by how often its op runs
(G_reg every 4th, D_reg every 16th iteration) and ranked by the time it loses against an ideal rate.
usage: python tools/conv_shapes.py [ideal_tflops] [resolution] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402
from inclusivegan_amd.dnnlib import tflib  # noqa: E402
from inclusivegan_amd.training import loss as L  # noqa: E402
from inclusivegan_amd.training.dataset import SyntheticDataset  # noqa: E402


def main():
    ideal = float(sys.argv[1]) if len(sys.argv) > 1 else 122.0
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 6
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

    def run(op):
        if op.startswith('G'):
            D.requires_grad_(False)
            G.zero_grad()
            loss, reg = L.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, r1, lab, z1, r2, lab, z2, NN_rec_lpips_weight=2.5,
                                                                phase='loss' if op == 'G_train' else 'reg')
            v = loss if op == 'G_train' else reg * 4
            torch.autograd.backward(v.mean(), inputs=list(G.trainables.values()))
            D.requires_grad_(True)
        else:
            G.requires_grad_(False)
            D.zero_grad()
            loss, reg = L.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase='loss' if op == 'D_train' else 'reg')
            v = loss if op == 'D_train' else reg * 16
            torch.autograd.backward(v.mean(), inputs=list(D.trainables.values()))
            G.requires_grad_(True)

    weights = {'G_train': 1.0, 'D_train': 1.0, 'G_reg': 0.25, 'D_reg': 1.0 / 16}
    agg = {}
    for op, wgt in weights.items():
        run(op)
        torch.cuda.synchronize()
        hip_ops.shape_log = []
        run(op)
        torch.cuda.synchronize()
        for kind, key, flops, splits, e0, e1 in hip_ops.shape_log:
            a = agg.setdefault((kind, key, splits), [0.0, 0.0, 0.0, set()])
            a[0] += wgt
            a[1] += wgt * flops
            a[2] += wgt * e0.elapsed_time(e1) * 1e-3
            a[3].add(op)
        hip_ops.shape_log = None
    tot_t = sum(v[2] for v in agg.values())
    tot_f = sum(v[1] for v in agg.values())
    print('conv family per iteration: %.2f ms, %.1f GFLOP, %.1f TFLOP/s; at %.0f TFLOP/s it would take %.2f ms' %
          (tot_t * 1e3, tot_f / 1e9, tot_f / tot_t / 1e12, ideal, tot_f / ideal / 1e9))
    print('%-8s %-46s %3s %6s %9s %7s %8s  %s' % ('kind', 'N,H,W,Cin,OH,OW,Cout k/s/u/p', 'spl', 'calls', 'us/call', 'TF/s', 'lost us', 'ops'))
    rows = sorted(agg.items(), key=lambda kv: -(kv[1][2] - kv[1][1] / ideal / 1e12))
    for (kind, key, splits), (calls, flops, secs, ops) in rows[:60]:
        n, h, w, cin, oh, ow, cout, g = key
        desc = '%d,%d,%d,%d,%d,%d,%d %d/%d/%d/%d' % (n, h, w, cin, oh, ow, cout, g.kh, g.stride, g.up, g.pad_y)
        print('%-8s %-46s %3d %6.2f %9.1f %7.1f %8.1f  %s' % (kind, desc, splits, calls, secs / calls * 1e6, flops / secs / 1e12,
                                                             (secs - flops / ideal / 1e12) * 1e6, ','.join(sorted(o[0] + o[2] for o in ops))))


if __name__ == '__main__':
    main()
