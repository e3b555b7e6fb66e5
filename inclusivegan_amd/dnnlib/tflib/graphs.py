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

from . import tfutil


def graphs_enabled(default=True):
    v = os.environ.get('IGAN_HIP_GRAPHS')
    if v is None:
        return default
    return v not in ('0', 'false', 'False', '')


def validation_enabled():
    """IGAN_GRAPH_VALIDATE=0 skips the replay-vs-eager check after capture (training_loop.py)."""
    return os.environ.get('IGAN_GRAPH_VALIDATE', '1') != '0'


def runtime_info():
    """What the replay check ran against (recorded in the bench line's `hip_graphs`): the HIP build PyTorch was made for, the ROCm
    release installed on the box, and the runtime's graph-packet-capture setting (inclusivegan_amd/__init__.py)."""
    rel = None
    for path in ('/opt/rocm/.info/version', '/opt/rocm/.info/version-dev'):
        try:
            with open(path) as f:
                rel = f.read().strip()
            break
        except OSError:
            pass
    return dict(torch_hip=getattr(torch.version, 'hip', None), rocm_release=rel, packet_capture=os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'))


def _one_stream():
    return os.environ.get('IGAN_GRAPH_ONE_STREAM', '1') != '0'      # A/B switch (0 = warm up on the default stream, capture on torch's own side stream)


_trace_log = None
_trace_saved = {}


def _trace_begin():
    """DIAGNOSTIC: wrap forward / backward of every autograd Function of hip_ops so that each tensor output leaves (label, [sum, sum of squares] in fp64, kept on
    the device: no host synchronisation) in a list."""
    global _trace_log
    from ... import hip_ops
    _trace_log = []
    log = _trace_log

    def wrap(cls, which):
        fn = getattr(cls, which)
        def wrapped(*args, **kw):
            for j, t in enumerate(args):            # the inputs too: is it the kernel or what it was given?
                if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point() and t.numel() > 0:
                    d = t.detach().double()
                    log.append(('%s.%s in[%d] %s' % (cls.__name__, which, j, tuple(t.shape)), torch.stack([d.sum(), (d * d).sum()])))
            out = fn(*args, **kw)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for j, t in enumerate(outs):
                if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point() and t.numel() > 0:
                    d = t.detach().double()
                    log.append(('%s.%s[%d] %s' % (cls.__name__, which, j, tuple(t.shape)), torch.stack([d.sum(), (d * d).sum()])))
            return out
        return staticmethod(wrapped)

    src = tfutil.random_source()            # every random draw of the op
    for meth in ('normal', 'uniform', 'normal_many'):
        fn0 = getattr(src, meth)
        def drawn(*args, _fn=fn0, _m=meth, **kw):
            out = _fn(*args, **kw)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                d = t.detach().double()
                log.append(('draw %s %s' % (_m, tuple(t.shape)), torch.stack([d.sum(), (d * d).sum()])))
            return out
        _trace_saved[(src, meth)] = fn0
        setattr(src, meth, drawn)
    for name in dir(hip_ops):
        cls = getattr(hip_ops, name)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            _trace_saved[cls] = (cls.__dict__['forward'], cls.__dict__['backward'])
            cls.forward = wrap(cls, 'forward')
            cls.backward = wrap(cls, 'backward')
    return log


def _trace_end():
    for key, val in _trace_saved.items():
        if isinstance(key, tuple):
            try:
                delattr(key[0], key[1])        # the instance attribute that shadowed the class's method
            except AttributeError:
                pass
        else:
            key.forward, key.backward = val
    _trace_saved.clear()


class GraphedStep:
    _pool = None
    _stream = None
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

    def _run_fn(self):
        # a TapRandom source groups the draws by op: tell it which op's Python is about to run (eager call or capture)
        src = tfutil.random_source()
        if hasattr(src, 'begin'):
            src.begin(self.name)
            try:
                return self.fn()
            finally:
                src.begin(None)
        return self.fn()

    @classmethod
    def side_stream(cls):
        """ONE stream for the eager warm-up calls and every capture.  Autograd remembers the stream a parameter's
        gradient-accumulation node was created on and sums the contributions of a parameter with several consumers (every
        modulated weight: convolution + demodulation) on THAT stream; nodes created by a warm-up call on the default
        stream and still alive at capture time would pull that summation out of the captured stream."""
        if cls._stream is None:
            cls._stream = torch.cuda.Stream()
        return cls._stream

    def check_replay(self, state, results, context=()):
        """One replay and one eager execution of the op, from the same values of `state` (tensors the op reads AND updates in
        place) and the same device-generator state, must agree bit for bit in `results(out)` (a list of tensors; `out` is the
        op's return value) and in `state`.  `context`: other captured steps, replayed right before the tested replay -- the
        way the training loop runs them back to back (the runtime fault this check exists for only shows when another
        graph of the shared pool has just run).  Leaves `state` and the generator as it found them.  Returns the list of
        (index, max abs difference) of the disagreeing tensors -- empty when the graph is faithful."""
        assert self.graph is not None, 'check_replay() needs a captured graph'
        saved = [t.detach().clone() for t in state]
        rng = torch.cuda.get_rng_state()

        def reset():
            with torch.no_grad():
                for t, v in zip(state, saved):
                    t.copy_(v)
            torch.cuda.set_rng_state(rng)

        def snapshot(out):
            return [t.detach().clone() for t in list(results(out)) + list(state)]

        for other in context:
            if other.graph is not None:
                other.graph.replay()
        reset()
        trace = os.environ.get('IGAN_GRAPH_CHECK_TRACE') == '1'       # DIAGNOSTIC (with EAGER_TWICE): a device-side checksum of every output of every hip_ops Function, both executions; the first ones that differ are printed
        if os.environ.get('IGAN_GRAPH_CHECK_EAGER_TWICE') == '1':      # DIAGNOSTIC: compare two EAGER executions instead (is the op itself reproducible here?)
            log_a = _trace_begin() if trace else None
            a = snapshot(self._run_fn())
            if trace:
                _trace_end()
        else:
            self.graph.replay()
            a = snapshot(self.out)
        reset()
        src = tfutil.random_source()
        tapped = src.by_op.get(self.name) if hasattr(src, 'by_op') else None     # a TapRandom must keep pointing at the GRAPH's draws
        log_b = _trace_begin() if (trace and os.environ.get('IGAN_GRAPH_CHECK_EAGER_TWICE') == '1') else None
        if _one_stream():       # the eager side runs where the warm-up calls and the capture ran (see side_stream)
            side, cur = GraphedStep.side_stream(), torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                b = snapshot(self._run_fn())
            cur.wait_stream(side)
        else:
            b = snapshot(self._run_fn())
        if log_b is not None:
            _trace_end()
            torch.cuda.synchronize()
            shown = 0
            for k, ((la, ta), (lb, tb)) in enumerate(zip(log_a, log_b)):
                if la != lb or not torch.equal(ta, tb):
                    print('TRACE-DIFF op %s call %d of %d: %s | %s: checksums %s vs %s' % (self.name, k, len(log_a), la, lb, ta.tolist(), tb.tolist()), flush=True)
                    shown += 1
                    if shown >= 6:
                        break
        if tapped is not None:
            src.by_op[self.name] = tapped
        reset()
        torch.cuda.synchronize()
        bad = []
        self.last_diff = []         # DIAGNOSTIC (IGAN_GRAPH_CHECK_VERBOSE=1): (index, flat positions that differ, replayed values, eager values) of the disagreeing tensors
        for i, (x, y) in enumerate(zip(a, b)):
            same = (x == y) | (torch.isnan(x) & torch.isnan(y)) if x.dtype.is_floating_point else (x == y)
            if x.shape != y.shape or not bool(same.all()):
                bad.append((i, float((x.double() - y.double()).abs().nan_to_num(nan=float('inf')).max()) if x.shape == y.shape else float('inf')))
                if os.environ.get('IGAN_GRAPH_CHECK_VERBOSE') == '1' and x.shape == y.shape:
                    pos = (~same).reshape(-1).nonzero().reshape(-1)
                    self.last_diff.append((i, pos.cpu(), x.reshape(-1)[pos].cpu(), y.reshape(-1)[pos].cpu()))
        return bad

    def __call__(self):
        if not self.enabled or GraphedStep.force_eager:
            self.calls += 1
            return self._run_fn()
        if self.calls < self.eager_calls:
            self.calls += 1
            if not _one_stream():
                return self._run_fn()
            s = GraphedStep.side_stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                out = self._run_fn()
            torch.cuda.current_stream().wait_stream(s)
            return out
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
            with torch.cuda.graph(g, pool=GraphedStep._pool, stream=GraphedStep.side_stream() if _one_stream() else None,
                                  capture_error_mode='thread_local' if dist_up else 'global'):
                self.out = self._run_fn()
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
