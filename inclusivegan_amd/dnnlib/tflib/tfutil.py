"""Graph-runtime glue the network / loss code needs, restated for eager PyTorch-ROCm:
lerp / slerp (dnnlib/tflib/tfutil.py:62-87), a variable-scope stack standing in for
tf.variable_scope / tf.get_variable, and an injectable random source standing in for
tf.random_normal / tf.random_uniform (TF's Philox streams cannot be reproduced, so parity tests
inject these tensors explicitly -- SURVEY.md section 7 "RNG parity is impossible").
"""
import contextlib
import threading

import numpy as np
import torch

#----------------------------------------------------------------------------
# Interpolation (tfutil.py:62-87).

def lerp(a, b, t):
    """Linear interpolation."""
    return a + (b - a) * t

def lerp_clip(a, b, t):
    return a + (b - a) * torch.clamp(t, 0.0, 1.0)

def normalize(v):
    """Normalize batch of vectors."""
    return v / torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True))

def slerp(a, b, t):
    """Spherical interpolation of a batch of vectors (output is unit-norm)."""
    a = normalize(a)
    b = normalize(b)
    d = torch.sum(a * b, dim=-1, keepdim=True)
    p = t * torch.acos(d)
    c = normalize(b - d * a)
    d = a * torch.cos(p) + c * torch.sin(p)
    return normalize(d)

#----------------------------------------------------------------------------
# Variable scopes.  A Network installs itself as the current store while its build function
# runs; `get_variable` then resolves "<scope>/<name>" inside that store, creating the variable
# on the first (template) pass exactly like tf.get_variable under reuse=tf.AUTO_REUSE.

_state = threading.local()

def _stack():
    if not hasattr(_state, 'scopes'):
        _state.scopes = []
        _state.store = None
    return _state

@contextlib.contextmanager
def variable_scope(name):
    st = _stack()
    st.scopes.append(name)
    try:
        yield
    finally:
        st.scopes.pop()

@contextlib.contextmanager
def variable_store(store):
    st = _stack()
    prev_store, prev_scopes = st.store, st.scopes
    st.store, st.scopes = store, []
    try:
        yield
    finally:
        st.store, st.scopes = prev_store, prev_scopes

def current_scope():
    return '/'.join(_stack().scopes)

def get_variable(name, shape=None, initializer=None, trainable=True):
    """tf.get_variable: `initializer` is one of ('zeros',), ('normal', std), ('const', value)."""
    st = _stack()
    if st.store is None:
        raise RuntimeError('get_variable() called outside of a Network build function')
    full = '/'.join(st.scopes + [name])
    return st.store._get_variable(full, shape, initializer, trainable)

def derived(key, fn):
    """Per-training-op cache of a tensor derived from variables only, scoped like get_variable
    (see Network.derived)."""
    st = _stack()
    if st.store is None:
        return fn()
    return st.store.derived('/'.join(st.scopes + [key]), fn)

def derived_many(keys, fn_many):
    """Several `derived` entries at once (see Network.derived_many); keys are relative to the current scope."""
    st = _stack()
    if st.store is None:
        return fn_many(list(range(len(keys))))
    return st.store.derived_many(['/'.join(st.scopes + [k]) for k in keys], fn_many)

#----------------------------------------------------------------------------
# Random source.  Default: torch device generator.  Tests install a RandomTape that replays
# recorded tensors in call order, and can record what the default source produced.

class _DefaultRandom:
    def normal(self, shape, device):
        return torch.randn(tuple(int(s) for s in shape), device=device, dtype=torch.float32)

    def uniform(self, shape, device, minval=0.0, maxval=1.0):
        return torch.rand(tuple(int(s) for s in shape), device=device, dtype=torch.float32) * (maxval - minval) + minval

    def randint(self, low, high, device):
        """Scalar integer tensor in [low, high) (tf.random_uniform([], low, high, dtype=int32))."""
        return torch.randint(int(low), int(high), (), device=device)

    def normal_many(self, shapes, device):
        """Several independent normal tensors, as consecutive draws; the device generator fills them in ONE launch."""
        shapes = [tuple(int(s) for s in sh) for sh in shapes]
        sizes = [int(np.prod(sh)) for sh in shapes]
        flat = torch.randn((sum(sizes),), device=device, dtype=torch.float32)
        return [t.view(sh) for t, sh in zip(torch.split(flat, sizes), shapes)]


class RandomTape:
    """Replays tensors in call order: entries are (kind, array-like). Raises when exhausted or on a
    kind/shape mismatch, so a test that injects the wrong number of random draws fails loudly."""

    def __init__(self, entries):
        self.entries = list(entries)
        self.pos = 0

    def _next(self, kind, shape, device):
        if self.pos >= len(self.entries):
            raise RuntimeError('RandomTape exhausted at draw %d (%s %s)' % (self.pos, kind, tuple(shape)))
        k, v = self.entries[self.pos]
        self.pos += 1
        if k != kind:
            raise RuntimeError('RandomTape draw %d: expected %s, tape has %s' % (self.pos - 1, kind, k))
        t = torch.as_tensor(np.asarray(v)).to(device)
        if kind != 'randint':
            t = t.to(torch.float32)
            if tuple(t.shape) != tuple(int(s) for s in shape):
                raise RuntimeError('RandomTape draw %d: shape %s != requested %s' % (self.pos - 1, tuple(t.shape), tuple(shape)))
        return t

    def normal(self, shape, device):
        return self._next('normal', shape, device)

    def uniform(self, shape, device, minval=0.0, maxval=1.0):
        return self._next('uniform', shape, device)

    def randint(self, low, high, device):
        return self._next('randint', (), device)

    def normal_many(self, shapes, device):
        return [self._next('normal', sh, device) for sh in shapes]


class RecordingRandom(_DefaultRandom):
    """Default source that also records every draw (to build a tape for the oracle)."""

    def __init__(self):
        self.entries = []

    def normal(self, shape, device):
        t = super().normal(shape, device)
        self.entries.append(('normal', t.detach().cpu().numpy()))
        return t

    def uniform(self, shape, device, minval=0.0, maxval=1.0):
        t = super().uniform(shape, device, minval, maxval)
        self.entries.append(('uniform', t.detach().cpu().numpy()))
        return t

    def randint(self, low, high, device):
        t = super().randint(low, high, device)
        self.entries.append(('randint', t.detach().cpu().numpy()))
        return t

    def normal_many(self, shapes, device):
        ts = super().normal_many(shapes, device)
        for t in ts:
            self.entries.append(('normal', t.detach().cpu().numpy()))
        return ts


class TapRandom(_DefaultRandom):
    """Default source that keeps the DEVICE tensors of every draw, grouped by the training op that made them, so that the
    draws of a captured hipGraph can be read back after each replay (a replay refills the same tensors; holding the
    references keeps the graph's memory pool from handing their storage to a later allocation).  `begin(name)` is called
    by GraphedStep before every Python execution of an op (eager run or capture); `snapshot(name)` copies the current
    contents to the host as a tape [(kind, ndarray), ...] in call order."""

    def __init__(self):
        self.by_op = {}
        self.current = None

    def begin(self, name):
        """name=None: no op is running (draws made now, e.g. by the IMLE refresh, are not kept)."""
        if name is None:
            self.current = None
        else:
            self.current = self.by_op[name] = []

    def _keep(self, kind, t):
        if self.current is not None:
            self.current.append((kind, t))
        return t

    def normal(self, shape, device):
        return self._keep('normal', super().normal(shape, device))

    def uniform(self, shape, device, minval=0.0, maxval=1.0):
        return self._keep('uniform', super().uniform(shape, device, minval, maxval))

    def randint(self, low, high, device):
        return self._keep('randint', super().randint(low, high, device))

    def normal_many(self, shapes, device):
        return [self._keep('normal', t) for t in super().normal_many(shapes, device)]

    def snapshot(self, name):
        return [(kind, t.detach().cpu().numpy().copy()) for kind, t in self.by_op.get(name, [])]


_random = _DefaultRandom()

def random_source():
    return _random

@contextlib.contextmanager
def use_random(source):
    global _random
    prev = _random
    _random = source
    try:
        yield source
    finally:
        _random = prev

def random_normal(shape, device):
    return _random.normal(shape, device)

def random_normal_many(shapes, device):
    """Consecutive normal draws of the given shapes (same tape order as calling random_normal for each)."""
    return _random.normal_many(shapes, device)

def random_uniform(shape, device, minval=0.0, maxval=1.0):
    return _random.uniform(shape, device, minval, maxval)

def random_int(low, high, device):
    return _random.randint(low, high, device)
