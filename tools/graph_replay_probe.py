"""Is a captured training op's REPLAY inside the real loop equal to its eager execution?  (The probe that exposed the HIP runtime's
graph-packet-capture fault in round 3: profiles/r03_graph_packet_capture.txt.)  Runs training_loop() at 32x32 until the first
replay of TARGET (default G_reg), then from the same state and generator state: replays it twice, runs it eagerly, replays it
after eager runs / replays of the other ops, and prints the relative L2 distance of the gradient buckets.
usage: [DEBUG_CLR_GRAPH_PACKET_CAPTURE=1] [IGAN_GRAPH_VALIDATE=0] [FMAP=1024] [TARGET=G_reg] python tools/graph_replay_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    def replay():
        restore(s0); orig_call(self); torch.cuda.synchronize(); return net.flat_grads.clone(), float(self.out.double().mean())
    def eager(name):
        restore(s0); steps[name].fn(); torch.cuda.synchronize(); return (G if name.startswith('G') else D).flat_grads.clone()
    g1, v1 = replay(); g2, v2 = replay()
    print('replay twice: equal %s (values %.8g %.8g)' % (torch.equal(g1, g2), v1, v2), flush=True)
    ge = eager(TARGET)
    print('eager %s vs replay: rel diff %.3e' % (TARGET, float((ge - g1).norm() / ge.norm())), flush=True)
    g3, v3 = replay()
    print('replay after eager %s: equal to first replay %s, rel diff to eager %.3e' % (TARGET, torch.equal(g3, g1), float((ge - g3).norm() / ge.norm())), flush=True)
    for other in ('G', 'D', 'D_reg', 'G_reg'):
        eager(other)
        g4, v4 = replay()
        print('replay after eager %-6s: equal to first replay %s, rel diff to eager %.3e' % (other, torch.equal(g4, g1), float((ge - g4).norm() / ge.norm())), flush=True)
    for other in ('G', 'D', 'D_reg'):
        restore(s0); orig_call(steps[other]); torch.cuda.synchronize()
        g5, v5 = replay()
        print('replay after REPLAY of %-6s: equal to first replay %s, rel diff to eager %.3e' % (other, torch.equal(g5, g1), float((ge - g5).norm() / ge.norm())), flush=True)
    os._exit(0)
graphs.GraphedStep.__call__ = checked
TL.training_loop(hooks=dict(on_start=lambda st: nets.update(st), on_iteration=lambda i: i['iteration'] >= 3), **T.loop_kwargs(int(os.environ.get('FMAP', '1024')), 6, data_size=48))
