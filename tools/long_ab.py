#!/usr/bin/env python3
"""A long same-seed run of the real loop under ONE arithmetic form of the large 3x3 convolutions: BASELINE config 2
(32x32, config-e width fmap_base 8192, minibatch_gpu 6, IMLE with NN_rec_lpips_weight 2.5, lazy regularisation), hipGraphs on, `--iters` iterations
(default 3000) from fixed host and device seeds.  Every training op's loss output is kept on the device and written out at the end: per iteration the
G loss, the D loss, the path-length penalty and pl_mean (every 4th), R1 (every 16th); per 500 iterations the fp16 form's window counter
(igan_debug_f16_window: non-zero elements imaged more than 2^26 below the largest of their own scale group / elements imaged).

    IGAN_CONV_PLANES=2 python tools/long_ab.py --out a.json ; IGAN_CONV_PLANES=0 python tools/long_ab.py --out b.json ; IGAN_CONV_PLANES=1 ... --out c.json
    python tools/long_ab.py --compare a.json b.json c.json > profiles/r05_long_ab.txt

Two fp32-level implementations of a GAN step do not stay on one trajectory: with beta1 = 0 Adam moves every weight by +-lr whatever the size of its
gradient, so the first gradient element whose sign differs in the last bit sends that weight the other way ("Teacher forcing").
What a same-seed A/B can show is (i) that the runs agree to rounding until they separate, (ii) that they separate from EACH OTHER no faster than two
reference-width forms (exact fp32 instruction vs exact three-piece bf16 split) separate, and (iii) that the loss statistics over windows of 250
iterations agree within the spread between those two -- which is what --compare tabulates."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import torch
    import inclusivegan_amd  # noqa: F401  (runtime flags before the first HIP call)
    from inclusivegan_amd import _abi, hostaffinity
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    hostaffinity.limit_host_threads()
    lib = _abi.get_plugin()
    form = lib.igan_conv_piece_form()
    B = 6
    kw = dict(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=args.fmap, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=args.fmap, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=args.res, num_channels=3, label_size=0),
        sched_args=EasyDict(minibatch_gpu_base=B, minibatch_size_base=B), tf_config={'rnd.np_random_seed': 1000},
        total_kimg=10 ** 6, data_size=args.data_size, num_samples_factor=4, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=256)
    vals = {'G': [], 'G_reg': [], 'D': [], 'D_reg': []}
    its = {'G': [], 'G_reg': [], 'D': [], 'D_reg': []}
    plm, window = [], []
    cur = dict(it=0)
    nets = {}

    def on_start(st):
        nets.update(st)
        torch.manual_seed(4242)
        lib.igan_debug_f16_window(None, None, 1)

    def on_op(name, out, feed):
        vals[name].append(out.detach().double().mean().reshape(1))         # stays on the device: no synchronisation per op
        its[name].append(cur['it'] + 1)
        if name == 'G_reg':
            plm.append(nets['G'].pl_mean_var.detach().double().reshape(1).clone())

    def on_it(info):
        cur['it'] = info['iteration']
        if info['iteration'] % 500 == 0 or info['iteration'] >= args.iters:
            below, imaged = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
            lib.igan_debug_f16_window(ctypes.byref(below), ctypes.byref(imaged), 1)
            window.append(dict(iteration=info['iteration'], below=below.value, imaged=imaged.value))
            print('iteration %d  window %d / %d' % (info['iteration'], below.value, imaged.value), flush=True)
        return info['iteration'] >= args.iters

    t0 = time.time()
    TL.training_loop(hooks=dict(on_start=on_start, on_op=on_op, on_iteration=on_it), **kw)
    torch.cuda.synchronize()
    rec = dict(form=form, iters=args.iters, res=args.res, fmap=args.fmap, seconds=time.time() - t0, window=window,
               pl_mean=[float(v) for v in torch.cat(plm).cpu()] if plm else [])
    for k in vals:
        rec[k] = dict(it=its[k], value=[float(v) for v in torch.cat(vals[k]).cpu()] if vals[k] else [])
    json.dump(rec, open(args.out, 'w'))
    print('form %d: %d iterations in %.1f s -> %s' % (form, args.iters, rec['seconds'], args.out))


def compare(paths):
    import numpy as np
    runs = [json.load(open(p)) for p in paths]
    names = {0: 'exact fp32 instruction', 1: 'three bf16 pieces (exact)', 2: 'two fp16 pieces (default)'}
    print('# same-seed runs of config 2 (%dx%d, fmap_base %d), %d iterations each' % (runs[0]['res'], runs[0]['res'], runs[0]['fmap'], runs[0]['iters']))
    for r in runs:
        print('# form %d = %s: %.0f s' % (r['form'], names[r['form']], r['seconds']))
        for w in r['window']:
            if r['form'] == 2:
                print('#    window counter up to iteration %5d: %d of %d imaged elements below 2^-26 of their own scale group (%.2e)' % (
                    w['iteration'], w['below'], w['imaged'], w['below'] / max(w['imaged'], 1)))
    def series(r, k):
        return np.array(r[k]['value'] if k != 'pl_mean' else r['pl_mean'])
    pairs = [(i, j) for i in range(len(runs)) for j in range(i + 1, len(runs))]
    print('\n## how long two forms stay together: first iteration at which the G loss differs by more than 1e-4 / 1e-2 relative')
    for i, j in pairs:
        a, b = series(runs[i], 'G'), series(runs[j], 'G')
        n = min(len(a), len(b))
        rel = np.abs(a[:n] - b[:n]) / np.maximum(np.abs(b[:n]), 1e-12)
        f = lambda t: int(np.argmax(rel > t)) + 1 if np.any(rel > t) else None
        print('form %d vs form %d: rel gap at iteration 1: %.2e, 2: %.2e, 5: %.2e, 10: %.2e;  > 1e-4 from iteration %s,  > 1e-2 from iteration %s' % (
            runs[i]['form'], runs[j]['form'], rel[0], rel[1], rel[4], rel[9], f(1e-4), f(1e-2)))
    W = 250
    print('\n## window means over %d iterations (G loss | D loss | path-length penalty | R1 | pl_mean at the window\'s end)' % W)
    nwin = runs[0]['iters'] // W
    for w in range(nwin):
        row = []
        for r in runs:
            cells = []
            for k in ('G', 'D', 'G_reg', 'D_reg'):
                it = np.array(r[k]['it']); v = series(r, k)
                sel = (it > w * W) & (it <= (w + 1) * W)
                cells.append(float(v[sel].mean()) if sel.any() else float('nan'))
            it = np.array(r['G_reg']['it']); pm = series(r, 'pl_mean')
            sel = it <= (w + 1) * W
            cells.append(float(pm[sel][-1]) if sel.any() else float('nan'))
            row.append(cells)
        print('iterations %5d-%5d  ' % (w * W + 1, (w + 1) * W) + '   '.join('form %d: %8.4f %8.4f %9.3e %9.3e %8.5f' % ((runs[i]['form'],) + tuple(c)) for i, c in enumerate(row)))
    print('\n## largest relative gap between window means (windows of %d iterations), per pair of forms' % W)
    for k, label in (('G', 'G loss'), ('D', 'D loss'), ('G_reg', 'path-length penalty'), ('D_reg', 'R1 penalty'), ('pl_mean', 'pl_mean')):
        line = '%-20s' % label
        for i, j in pairs:
            gaps = []
            for w in range(nwin):
                m = []
                for r in (runs[i], runs[j]):
                    if k == 'pl_mean':
                        it = np.array(r['G_reg']['it']); v = series(r, 'pl_mean'); sel = it <= (w + 1) * W
                        m.append(v[sel][-1] if sel.any() else np.nan)
                    else:
                        it = np.array(r[k]['it']); v = series(r, k); sel = (it > w * W) & (it <= (w + 1) * W)
                        m.append(v[sel].mean() if sel.any() else np.nan)
                gaps.append(abs(m[0] - m[1]) / max(abs(m[1]), 1e-12))
            line += '   form %d vs %d: %.3f' % (runs[i]['form'], runs[j]['form'], np.nanmax(gaps))
        print(line)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=3000)
    ap.add_argument('--res', type=int, default=32)
    ap.add_argument('--fmap', type=int, default=8192)
    ap.add_argument('--data-size', type=int, default=1536)
    ap.add_argument('--out', default='long_ab.json')
    ap.add_argument('--compare', nargs='+')
    a = ap.parse_args()
    if a.compare:
        compare(a.compare)
    else:
        run(a)
