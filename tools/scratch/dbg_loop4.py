"""G_reg of iteration 1 in the graph-mode loop: HIP gradient vs fp64 oracle vs fp32 oracle, all from HIP's own pre-op weights."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_loop_parity as T
from oracle import loss as OL
from oracle.misc import Tape
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil
fmap = 1024
src = tfutil.TapRandom()
log = dict(ops=[]); nets = {}
def on_start(st):
    nets.update(st)
def on_op(name, out, feed):
    G = nets['G']
    log['ops'].append(dict(name=name, value=float(out.detach().double().mean()), tape=src.snapshot(name), gG=G.flat_grads.detach().clone(),
                           vars={n: v.detach().cpu().clone() for n, v in G.vars.items()}, pl=float(G.pl_mean_var)))
TL.training_loop(hooks=dict(on_start=on_start, on_op=on_op, on_iteration=lambda i: i['iteration'] >= 1, random_source=src), **T.loop_kwargs(fmap, 6, data_size=48))
G, D = nets['G'], nets['D']
cfg = dict(resolution=32, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
pre, reg = log['ops'][0], log['ops'][1]
assert reg['name'] == 'G_reg'
z = torch.zeros(6, 512)
res = {}
for dt in (torch.float64, torch.float32):
    gp = {n: v.to(dt) for n, v in pre['vars'].items()}       # weights after the G update = before G_reg
    for n in G.trainables: gp[n].requires_grad_(True)
    dp = {n: v.detach().cpu().to(dt) for n, v in D.vars.items()}
    _, ro, _ = OL.G_loss(gp, dp, {}, cfg, Tape(reg['tape'], dt), 6, None, z.to(dt), None, z.to(dt), 2.5, phase='reg', state={})
    (ro * 4).mean().backward()
    res[dt] = (float(ro.mean()), {n: gp[n].grad.double() for n in G.trainables if gp[n].grad is not None})
hip = {n: reg['gG'][o:o + c].double().cpu().reshape(G.vars[n].shape) for n, (o, c) in G._offsets.items()}
def cmp(a, b, filt=lambda n: True):
    num = sum(float((a[n] - b[n]).norm() ** 2) for n in b if filt(n)); den = sum(float(b[n].norm() ** 2) for n in b if filt(n))
    return (num / den) ** .5
o64, o32 = res[torch.float64][1], res[torch.float32][1]
print('value hip %.9g o64 %.9g o32 %.9g' % (reg['value'], res[torch.float64][0], res[torch.float32][0]))
m = lambda n: 'G_mapping' in n
print('grad rel L2: hip vs o64 all %.2e mapping %.2e | o32 vs o64 all %.2e mapping %.2e | hip vs o32 all %.2e mapping %.2e' % (
    cmp(hip, o64), cmp(hip, o64, m), cmp(o32, o64), cmp(o32, o64, m), cmp(hip, o32), cmp(hip, o32, m)))
for n in o64:
    print('   %-45s hip/o64 %.2e  o32/o64 %.2e  |g| %.2e' % (n, cmp(hip, o64, lambda k: k == n), cmp(o32, o64, lambda k: k == n), float(o64[n].norm())))
