"""First tensor that differs between a captured op's REPLAY inside the real loop and its eager execution on the same draws: an
op-level trace (every aten op on the main thread AND the autograd engine thread, every hip_ops Function) of both runs, compared
record by record.  Used in round 3 to locate the unfaithful replay (profiles/r03_graph_packet_capture.txt).
usage: [DEBUG_CLR_GRAPH_PACKET_CAPTURE=1] [IGAN_GRAPH_VALIDATE=0] [FMAP=1024] [TARGET=G_reg] python tools/graph_trace_diff.py"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
            flat_in = []
            for x in args:
                if torch.is_tensor(x): flat_in.append(x)
                elif isinstance(x, (list, tuple)): flat_in += [y for y in x if torch.is_tensor(y)]
            TR['cur'].append((threading.current_thread().name[:4] + ':' + str(func), flat_in, [o for o in outs if torch.is_tensor(o)]))
        return out
engine_threads = set()
def wrap(cls, meth):
    orig = getattr(cls, meth)
    def w(ctx, *a):
        tid = threading.get_ident()
        if meth == 'backward' and tid != MAIN and tid not in engine_threads:
            engine_threads.add(tid); Log().__enter__()          # stays on for the life of the engine thread
        if TR['cur'] is not None: TR['cur'].append(('>>' + cls.__name__ + '.' + meth, [], []))
        out = orig(ctx, *a)
        if TR['cur'] is not None:
            outs = out if isinstance(out, tuple) else (out,)
            TR['cur'].append(('<<' + cls.__name__ + '.' + meth, [x for x in a if torch.is_tensor(x)], [o for o in outs if torch.is_tensor(o)]))
        return out
    setattr(cls, meth, staticmethod(w))
MAIN = threading.get_ident()
for name in dir(hip_ops):
    c = getattr(hip_ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, 'forward'); wrap(c, 'backward')

TARGET = os.environ.get('TARGET', 'G_reg')
tap = tfutil.TapRandom()
nets = {}; traces = {}
orig_run = graphs.GraphedStep._run_fn
def run_fn(self):
    if self.name == TARGET and torch.cuda.is_current_stream_capturing():
        TR['cur'] = traces[self.name] = []
        try:
            with Log():
                return orig_run(self)
        finally:
            TR['cur'] = None
    return orig_run(self)
graphs.GraphedStep._run_fn = run_fn
orig_call = graphs.GraphedStep.__call__
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
    g_rec = snap(traces[self.name]); gg = G.flat_grads.clone()
    with torch.no_grad():
        G.vars['dlatent_avg'].copy_(s0['avg']); G.pl_mean_var.copy_(s0['pl'])
    TR['cur'] = e_tr = []
    with tfutil.use_random(tfutil.RandomTape(tape)), Log():
        self.fn()
    TR['cur'] = None
    e_rec = snap(e_tr); ge = G.flat_grads.clone()
    print('REPLAY %s final grads rel diff %.3e   records %d / %d' % (self.name, float((gg - ge).norm() / ge.norm()), len(g_rec), len(e_rec)), flush=True)
    def rel(x, y): return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) if x.shape == y.shape else -1
    # align: drop main-thread records that only exist in the eager run (tape uploads): compare per thread-tag subsequences of engine-thread records
    ga = [r for r in g_rec if not r[0].startswith('Main')]; ea = [r for r in e_rec if not r[0].startswith('Main')]
    print('   engine-thread + Function records: %d / %d' % (len(ga), len(ea)))
    shown = 0; ctx = []
    for i, (a, b) in enumerate(zip(ga, ea)):
        if a[0] != b[0]:
            print('   misaligned at %d: %s vs %s' % (i, a[0], b[0])); break
        if a[0][:2] in ('>>', '<<'): ctx.append(a[0])
        din = [k for k, (x, y) in enumerate(zip(a[1], b[1])) if x.dtype.is_floating_point and (x.shape != y.shape or not torch.equal(x, y))]
        dout = [k for k, (x, y) in enumerate(zip(a[2], b[2])) if x.dtype.is_floating_point and (x.shape != y.shape or not torch.equal(x, y))]
        if din or dout:
            print('  #%d %s | in differ %s %s | out differ %s %s | in shapes %s | inside %s' % (i, a[0], din, ['%.1e' % rel(a[1][k], b[1][k]) for k in din], dout, ['%.1e' % rel(a[2][k], b[2][k]) for k in dout],
                  [tuple(t.shape) for t in a[1]], ctx[-2:]), flush=True)
            if shown == 0:
                x, y = a[2][0], b[2][0]
                print('      graph out: stride %s norm %.4e first %s' % (x.stride(), float(x.norm()), x.reshape(-1)[:8].tolist()))
                print('      eager out: stride %s norm %.4e first %s' % (y.stride(), float(y.norm()), y.reshape(-1)[:8].tolist()))
                xi, yi = a[1][0], b[1][0]
                print('      input stride graph %s eager %s; recomputed sum of GRAPH input now: rel to eager out %.2e, rel to graph out %.2e' % (xi.stride(), yi.stride(), rel(xi.sum(dim=(2, 3)), y), rel(xi.sum(dim=(2, 3)), x)))
                print('      ratio graph/eager (first 8): %s' % ((x.reshape(-1)[:8] / y.reshape(-1)[:8]).tolist()))
                for k in range(max(0, i - 6), i): print('      prev #%d %s in %s out %s' % (k, ga[k][0], [tuple(t.shape) for t in ga[k][1]], [tuple(t.shape) for t in ga[k][2]]))
            shown += 1
            if shown >= 3: break
    os._exit(0)
graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 3, random_source=tap), **T.loop_kwargs(int(os.environ.get('FMAP', '1024')), 6, data_size=48))
