"""Allocations made while G_reg is being captured that do NOT land in the graph's private pool, with Python stacks."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil, graphs
TARGET = os.environ.get('TARGET', 'G_reg')
orig_run = graphs.GraphedStep._run_fn
def run_fn(self):
    if self.name == TARGET and torch.cuda.is_current_stream_capturing():
        torch.cuda.memory._record_memory_history(enabled='all', context='alloc', stacks='python', max_entries=200000)
        try:
            return orig_run(self)
        finally:
            snap = torch.cuda.memory._snapshot()
            torch.cuda.memory._record_memory_history(enabled=None)
            segs = [(s['address'], s['address'] + s['total_size'], tuple(s.get('segment_pool_id', (0, 0))), s.get('stream')) for s in snap['segments']]
            def pool_of(addr):
                for lo, hi, pid, st in segs:
                    if lo <= addr < hi: return pid, st
                return None, None
            n_pool = n_def = 0
            seen = {}
            for ev in snap['device_traces'][0]:
                if ev['action'] != 'alloc': continue
                pid, st = pool_of(ev['addr'])
                if pid is not None and pid != (0, 0):
                    n_pool += 1; continue
                n_def += 1
                fr = [f for f in ev.get('frames', []) if 'inclusivegan_amd' in f['filename'] or 'tests/' in f['filename']]
                key = tuple((f['filename'].split('/')[-1], f['line'], f['name']) for f in fr[:6])
                seen.setdefault(key, []).append((ev['size'], ev.get('stream'), pid))
            print('CAPTURE %s: %d allocations in the private pool, %d elsewhere' % (self.name, n_pool, n_def), flush=True)
            for key, v in seen.items():
                print('  x%d sizes %s streams %s pool %s' % (len(v), sorted(set(s for s, _, _ in v))[:6], sorted(set(str(s) for _, s, _ in v)), v[0][2]))
                for k in key: print('       %s:%d %s' % k)
    return orig_run(self)
graphs.GraphedStep._run_fn = run_fn
TL.training_loop(hooks=dict(on_iteration=lambda i: True), **T.loop_kwargs(1024, 6, data_size=48))
