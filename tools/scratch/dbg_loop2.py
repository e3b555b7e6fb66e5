import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.dnnlib.tflib import graphs as _g
_orig = _g.GraphedStep.__call__
_sync = os.environ.get('DBG_SYNC', '')
def _call(self):
    if 'before' in _sync: torch.cuda.synchronize()
    out = _orig(self)
    if 'after' in _sync: torch.cuda.synchronize()
    return out
_g.GraphedStep.__call__ = _call
from tests.util import gloss_tape_in_reference_order
from oracle.train_ops import TrainOps
fmap = int(os.environ.get('FMAP', '1024'))
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil
for mode in os.environ.get('MODES', '1,0').split(','):
    os.environ['IGAN_HIP_GRAPHS'] = mode
    src = tfutil.TapRandom()
    log = dict(ops=[])
    nets = {}
    def on_start(st):
        nets.update(st)
        G, D, lp = st['G'], st['D'], st['lpips']
        log['init'] = dict(G={n: v.detach().cpu().numpy().copy() for n, v in G.vars.items()}, D={n: v.detach().cpu().numpy().copy() for n, v in D.vars.items()},
            lpips={n: v.detach().cpu().numpy().copy() for n, v in lp.vars.items()},
            G_layout={n: (int(o), int(c), tuple(G.vars[n].shape)) for n, (o, c) in G._offsets.items()},
            D_layout={n: (int(o), int(c), tuple(D.vars[n].shape)) for n, (o, c) in D._offsets.items()})
    def on_op(name, out, feed):
        rec = dict(name=name, value=float(out.detach().double().mean()), tape=src.snapshot(name), wG=nets['G'].flat_params.detach().cpu().numpy().copy(), wD=nets['D'].flat_params.detach().cpu().numpy().copy(),
                   gG=nets['G'].flat_grads.detach().cpu().numpy().copy(), gD=nets['D'].flat_grads.detach().cpu().numpy().copy())
        if name in ('D', 'D_reg'): rec['reals'] = feed['reals'].cpu().numpy().copy()
        else:
            for k in ('reals_rec_1', 'latents_rec_1', 'reals_rec_2', 'latents_rec_2'): rec[k] = feed[k].cpu().numpy().copy()
        log['ops'].append(rec)
    TL.training_loop(hooks=dict(on_start=on_start, on_op=on_op, on_iteration=lambda i: i['iteration'] >= 1, random_source=src), **T.loop_kwargs(fmap, 6, data_size=48))
    init = log['init']
    cfg = dict(resolution=32, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
    ops = TrainOps(init['G'], init['D'], init['G_layout'], init['D_layout'], init['lpips'], cfg, world=1, minibatch_gpu=6)
    # keep the oracle's raw gradient: wrap adam.apply
    last = {}
    for key in ('G', 'D'):
        orig = ops.adam[key].apply
        def wrapped(w, g, lr=None, _o=orig, _k=key):
            last[_k] = g.copy(); return _o(w, g, lr)
        ops.adam[key].apply = wrapped
    for r in log['ops']:
        name = r['name']
        if name in ('G', 'G_reg'):
            t = dict(r, tape=gloss_tape_in_reference_order(r['tape'], 6) if name == 'G' else r['tape'])
            v = ops.G_op([t], 'loss' if name == 'G' else 'reg')
        else:
            v = ops.D_op([r], 'loss' if name == 'D' else 'reg')
        if name == 'D': ops.Gs_update()
        k = 'G' if name.startswith('G') else 'D'
        gh, go = r['g' + k], last[k]
        wh, wo = r['w' + k][:ops.w[k].size], ops.w[k]
        gh = gh[:go.size]
        dw = np.abs(wh - wo)
        print('graphs %s %-5s value hip %.9g oracle %.9g | grad rel L2 %.2e max|g| %.2e | weights: max diff %.2e, n(diff>1e-4) %d of %d' % (
            mode, name, r['value'], v[0], np.linalg.norm(gh - go) / np.linalg.norm(go), np.abs(go).max(), dw.max(), int((dw > 1e-4).sum()), dw.size), flush=True)
        bad = np.nonzero(dw > 1e-4)[0]
        if len(bad):
            lay = init[k + '_layout']
            names = {}
            for n, (o, c, _) in lay.items():
                cnt = int(((bad >= o) & (bad < o + c)).sum())
                if cnt: names[n] = (cnt, c)
            print('     ', sorted(names.items(), key=lambda kv: -kv[1][0])[:8])
            i = bad[0]
            print('      e.g. idx %d: g_hip %.3e g_oracle %.3e w_hip %.6f w_oracle %.6f' % (i, gh[i], go[i], wh[i], wo[i]))
