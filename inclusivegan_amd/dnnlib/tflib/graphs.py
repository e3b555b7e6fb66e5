"""hipGraph capture of a training op.

The reference runs each of its four training ops (G_train_op, G_reg_op, D_train_op, D_reg_op,
training_loop.py:294-297,474-479) as one `session.run` of a pre-built TF graph.  The eager PyTorch
equivalent issues ~10^3 small launches per op from Python, which at minibatch_gpu = 6 leaves the GPU
idle between kernels (the low-resolution layers take a few microseconds each).  `GraphedStep` restores
the launch-once behaviour: after `eager_calls` ordinary executions (which also create every lazily
allocated piece of state) the op is captured into a HIP graph and replayed from then on.

Requirements on `fn` (all met by the training ops): reads its inputs from static device buffers,
performs no host synchronisation and no host<->device copies, keeps Python-side state changes out
(they would not be replayed), returns tensors (or a tuple/dict of tensors) that callers only read.
All steps share one memory pool: they never run concurrently.
"""
import os

import torch


def graphs_enabled(default=True):
    v = os.environ.get('IGAN_HIP_GRAPHS')
    if v is None:
        return default
    return v not in ('0', 'false', 'False', '')


class GraphedStep:
    _pool = None
    force_eager = False     # profiling switch: run every step eagerly (per-launch event timing)
    generation = 0          # bump to make every step re-capture its graph at its next call (e.g. after switching on hip_ops.stamp_log)
    after_capture = None    # optional callable run INSIDE the capture, after fn() (e.g. StampLog.fold); receives the step

    def __init__(self, fn, enabled=True, eager_calls=2, name='step'):
        self.fn = fn
        self.enabled = enabled
        self.eager_calls = eager_calls
        self.name = name
        self.calls = 0
        self.graph = None
        self.out = None
        self.captured_generation = 0
        self.replays = 0        # replays of the CURRENT graph (first capture counts: a capture does not execute, the replay after it does)

    def __call__(self):
        if not self.enabled or GraphedStep.force_eager or self.calls < self.eager_calls:
            self.calls += 1
            return self.fn()
        if self.graph is not None and self.captured_generation != GraphedStep.generation:
            torch.cuda.synchronize()
            self.graph = None
            self.out = None
        if self.graph is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # With a process group up, RCCL's watchdog thread polls events while we capture: in the default
            # "global" mode that foreign call would invalidate the capture; only this thread's calls matter here.
            dist_up = torch.distributed.is_available() and torch.distributed.is_initialized()
            with torch.cuda.graph(g, pool=GraphedStep._pool, capture_error_mode='thread_local' if dist_up else 'global'):
                self.out = self.fn()
                if GraphedStep.after_capture is not None:
                    GraphedStep.after_capture(self)
            self.captured_generation = GraphedStep.generation
            self.replays = 0
            if GraphedStep._pool is None:
                GraphedStep._pool = g.pool()
            self.graph = g
        self.calls += 1
        self.replays += 1
        self.graph.replay()
        return self.out
