"""Dump the captured G_reg hipGraph as DOT and analyse its shape (roots, forks, joins)."""
import os, sys, re, collections
sys.path.insert(0, os.getcwd())
import torch
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import graphs
_orig_graph = torch.cuda.CUDAGraph
class DbgGraph(_orig_graph):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        return g
made = []
def factory(*a, **k):
    g = _orig_graph(*a, **k)
    g.enable_debug_mode()
    made.append(g)
    return g
torch.cuda.CUDAGraph = factory
def on_start(st):
    os.makedirs('gpurun_out/dot', exist_ok=True)
    for i, g in enumerate(made):
        path = 'gpurun_out/dot/graph%d.dot' % i
        g.debug_dump(path)
        txt = open(path).read()
        edges = re.findall(r'"?([\w ]+?)"?\s*->\s*"?([\w ]+?)"?\s*[;\[]', txt)
        nodes = set(re.findall(r'^\s*"?([\w ]+?)"?\s*\[', txt, re.M))
        succ = collections.Counter(a for a, b in edges); pred = collections.Counter(b for a, b in edges)
        alln = nodes | set(a for a, b in edges) | set(b for a, b in edges)
        roots = [n for n in alln if pred[n] == 0]; leaves = [n for n in alln if succ[n] == 0]
        print('GRAPH %d: %d bytes, %d nodes, %d edges, %d roots, %d leaves, %d forks, %d joins' % (i, len(txt), len(alln), len(edges), len(roots), len(leaves),
              sum(1 for n in alln if succ[n] > 1), sum(1 for n in alln if pred[n] > 1)), flush=True)
    os._exit(0)
TL.training_loop(hooks=dict(on_start=on_start), **T.loop_kwargs(1024, 6, data_size=48))
