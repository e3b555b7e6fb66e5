"""Which aten outputs must be kept alive during the capture of G_reg for its replay to equal eager?  HOLD=<regex on op name>."""
import os, sys, re
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil, graphs
HOLD = re.compile(os.environ.get('HOLD', '^$'))
LO, HI = int(os.environ.get('LO', '0')), int(os.environ.get('HI', '1000000'))
held = []
names = {}
cnt = [0]
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        n = str(func)
        names[n] = names.get(n, 0) + 1
        if HOLD.search(n):
            if LO <= cnt[0] < HI:
                held.append(out)
            cnt[0] += 1
        return out
TARGET = 'G_reg'
tap = tfutil.TapRandom()
nets = {}
orig_run = graphs.GraphedStep._run_fn
def run_fn(self):
    if self.name == TARGET and torch.cuda.is_current_stream_capturing():
        with Log():
            return orig_run(self)
    return orig_run(self)
graphs.GraphedStep._run_fn = run_fn
orig_call = graphs.GraphedStep.__call__
def checked(self):
    if not nets or self.graph is None or self.name != TARGET:
        return orig_call(self)
    G = nets['G']
    s0 = dict(avg=G.vars['dlatent_avg'].detach().clone(), pl=G.pl_mean_var.detach().clone())
    out = orig_call(self)
    torch.cuda.synchronize()
    tape = tap.snapshot(self.name)
    gg = G.flat_grads.clone(); vg = float(out.detach().double().mean())
    with torch.no_grad():
        G.vars['dlatent_avg'].copy_(s0['avg']); G.pl_mean_var.copy_(s0['pl'])
    with tfutil.use_random(tfutil.RandomTape(tape)):
        ve = float(self.fn().detach().double().mean())
    ge = G.flat_grads.clone()
    print('HOLD %-40s [%d,%d) matched %5d held %5d | value graph %.8g eager %.8g | grad rel diff %.3e' % (HOLD.pattern, LO, HI, cnt[0], len(held), vg, ve, float((gg - ge).norm() / ge.norm())), flush=True)
    if os.environ.get('NAMES'):
        print(sorted(names.items(), key=lambda kv: -kv[1]))
    os._exit(0)
graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 3, random_source=tap), **T.loop_kwargs(1024, 6, data_size=48))
