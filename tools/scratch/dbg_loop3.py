"""In the real loop (graphs on): after every replay, re-run the op's Python eagerly on the tapped draws from the saved state; compare."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil, graphs
fmap = int(os.environ.get('FMAP', '1024'))
tap = tfutil.TapRandom()
nets = {}
orig = graphs.GraphedStep.__call__

def state():
    G = nets['G']
    return dict(avg=G.vars['dlatent_avg'].detach().clone(), pl=G.pl_mean_var.detach().clone())

def restore(s):
    G = nets['G']
    with torch.no_grad():
        G.vars['dlatent_avg'].copy_(s['avg']); G.pl_mean_var.copy_(s['pl'])

def checked(self):
    if not nets or self.graph is None or not self.enabled:
        return orig(self)
    G, D = nets['G'], nets['D']
    net = G if self.name.startswith('G') else D
    s0 = state()
    out = orig(self)
    torch.cuda.synchronize()
    tape = tap.snapshot(self.name)
    vg, gg, s1 = out.detach().clone(), net.flat_grads.clone(), state()
    restore(s0)
    with tfutil.use_random(tfutil.RandomTape(tape)):
        ve = self.fn().detach().clone()
    ge, s2 = net.flat_grads.clone(), state()
    torch.cuda.synchronize()
    nb = sum(0 if torch.equal(ge[o:o + c], gg[o:o + c]) else 1 for n, (o, c) in net._offsets.items())
    print('replay %-6s value equal %s (%.8g vs %.8g) grads equal %s (%d vars differ, rel L2 %.2e) state equal %s %s' % (
        self.name, torch.equal(ve, vg), float(ve.mean()), float(vg.mean()), torch.equal(ge, gg), nb, float((ge - gg).norm() / ge.norm()),
        torch.equal(s1['avg'], s2['avg']), torch.equal(s1['pl'], s2['pl'])), flush=True)
    net.flat_grads.copy_(gg); restore(s1)
    return out

graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 5, random_source=tap), **T.loop_kwargs(fmap, 6, data_size=48))
