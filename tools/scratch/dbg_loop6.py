"""Every aten op + every hip_ops Function of the G_reg op: replay-in-loop vs eager on the same draws; first differing outputs."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
import tests.test_gpu_loop_parity as T
from inclusivegan_amd import hip_ops
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil, graphs

TR = {'cur': None}
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if TR['cur'] is not None:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            TR['cur'].append((str(func), [x for x in args if torch.is_tensor(x)], [o for o in outs if torch.is_tensor(o)]))
        return out
def wrap(cls, meth):
    orig = getattr(cls, meth)
    def w(ctx, *a):
        if TR['cur'] is not None: TR['cur'].append(('>>' + cls.__name__ + '.' + meth, [], []))
        out = orig(ctx, *a)
        if TR['cur'] is not None:
            outs = out if isinstance(out, tuple) else (out,)
            TR['cur'].append(('<<' + cls.__name__ + '.' + meth, [x for x in a if torch.is_tensor(x)], [o for o in outs if torch.is_tensor(o)]))
        return out
    setattr(cls, meth, staticmethod(w))
for name in dir(hip_ops):
    c = getattr(hip_ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, 'forward'); wrap(c, 'backward')

TARGET = os.environ.get('TARGET', 'G_reg')
tap = tfutil.TapRandom()
nets = {}; traces = {}
orig_run = graphs.GraphedStep._run_fn
def run_fn(self):
    if self.name == TARGET:
        TR['cur'] = traces[self.name] = []
        try:
            with Log():
                return orig_run(self)
        finally:
            TR['cur'] = None
    return orig_run(self)
graphs.GraphedStep._run_fn = run_fn
orig_call = graphs.GraphedStep.__call__
junk = []
def snap(tr):
    return [(tag, [t.detach().clone() for t in ins], [t.detach().clone() for t in outs]) for tag, ins, outs in tr]
def checked(self):
    if not nets or self.graph is None or self.name != TARGET:
        return orig_call(self)
    G = nets['G']
    s0 = dict(avg=G.vars['dlatent_avg'].detach().clone(), pl=G.pl_mean_var.detach().clone())
    out = orig_call(self)
    torch.cuda.synchronize()
    tape = tap.snapshot(self.name)
    g_rec = snap(traces[self.name])
    vg = float(out.detach().double().mean())
    with torch.no_grad():
        G.vars['dlatent_avg'].copy_(s0['avg']); G.pl_mean_var.copy_(s0['pl'])
    TR['cur'] = e_tr = []
    with tfutil.use_random(tfutil.RandomTape(tape)), Log():
        ve = float(self.fn().detach().double().mean())
    TR['cur'] = None
    e_rec = snap(e_tr)
    print('REPLAY %s value graph %.8g eager %.8g   records %d / %d' % (self.name, vg, ve, len(g_rec), len(e_rec)), flush=True)
    shown = 0
    def rel(x, y): return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) if x.shape == y.shape else -1
    ctx = []
    i = j = 0
    # the eager run under RandomTape has extra ops (tape uploads); align on op names greedily
    while i < len(g_rec) and j < len(e_rec) and shown < 10:
        a, b = g_rec[i], e_rec[j]
        if a[0] != b[0]:
            # skip ahead in whichever stream re-synchronises sooner
            nj = next((k for k in range(j, min(j + 40, len(e_rec))) if e_rec[k][0] == a[0]), None)
            ni = next((k for k in range(i, min(i + 40, len(g_rec))) if g_rec[k][0] == b[0]), None)
            if nj is not None and (ni is None or nj - j <= ni - i): j = nj; continue
            if ni is not None: i = ni; continue
            print('  cannot align at', i, a[0], j, b[0]); break
        if a[0].startswith('>>') or a[0].startswith('<<'): ctx.append(a[0])
        dout = [k for k, (x, y) in enumerate(zip(a[2], b[2])) if x.dtype.is_floating_point and (x.shape != y.shape or not torch.equal(x, y))]
        din = [k for k, (x, y) in enumerate(zip(a[1], b[1])) if x.dtype.is_floating_point and (x.shape != y.shape or not torch.equal(x, y))]
        if dout or din:
            print('  #%d/%d %s | inputs differ %s %s | outputs differ %s %s | shapes in %s out %s | inside %s' % (i, j, a[0], din, ['%.1e' % rel(a[1][k], b[1][k]) for k in din], dout,
                  ['%.1e' % rel(a[2][k], b[2][k]) for k in dout], [tuple(t.shape) for t in a[1]], [tuple(t.shape) for t in a[2]], ctx[-3:]), flush=True)
            shown += 1
        i += 1; j += 1
    os._exit(0)
graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 3, random_source=tap), **T.loop_kwargs(1024, 6, data_size=48))
