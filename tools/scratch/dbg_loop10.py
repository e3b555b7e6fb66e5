"""Poison G.flat_grads with a marker before a G_reg replay: which variables does the replay fail to (re)write?"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil, graphs
TARGET = os.environ.get('TARGET', 'G_reg')
nets = {}; steps = {}
orig_init = graphs.GraphedStep.__init__
def init(self, *a, **k):
    orig_init(self, *a, **k); steps[self.name] = self
graphs.GraphedStep.__init__ = init
orig_call = graphs.GraphedStep.__call__
def checked(self):
    if not nets or self.graph is None or self.name != TARGET:
        return orig_call(self)
    G, D = nets['G'], nets['D']
    net = G if TARGET.startswith('G') else D
    def save(): return dict(avg=G.vars['dlatent_avg'].detach().clone(), pl=G.pl_mean_var.detach().clone(), rng=torch.cuda.get_rng_state())
    def restore(s):
        with torch.no_grad():
            G.vars['dlatent_avg'].copy_(s['avg']); G.pl_mean_var.copy_(s['pl'])
        torch.cuda.set_rng_state(s['rng'])
    s0 = save()
    res = {}
    for marker in (0.0, 1234.5, float('nan')):
        restore(s0); net.flat_grads.fill_(marker); orig_call(self); torch.cuda.synchronize()
        res[str(marker)] = net.flat_grads.clone()
    restore(s0); net.flat_grads.fill_(777.0); self.fn(); torch.cuda.synchronize(); ge = net.flat_grads.clone()
    print('eager (bucket pre-filled with 777) vs replay with bucket pre-zeroed: rel diff %.3e' % float((ge - res['0.0']).norm() / ge.norm()))
    for n, (o, c) in net._offsets.items():
        a, b, z = res['1234.5'][o:o + c], res['0.0'][o:o + c], ge[o:o + c]
        nn_ = int(torch.isnan(res['nan'][o:o + c]).sum())
        if not torch.equal(a, b) or nn_:
            print('  %-45s marker leaks: max|a-b| %.4g  nan count %d of %d | replay(0-filled) vs eager rel %.3e' % (n, float((a - b).abs().max()), nn_, c, float((b - z).norm() / (z.norm() + 1e-30))))
    os._exit(0)
graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 3), **T.loop_kwargs(1024, 6, data_size=48))
