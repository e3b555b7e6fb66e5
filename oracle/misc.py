"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Host-side helpers restated from the reference: training/misc.py:36-41 (adjust_dynamic_range),
:190-203 (normalize, slerp); dnnlib/tflib/tfutil.py:62-87 (lerp, slerp on tensors).
PINNED: tests/golden/misc_golden.npz holds outputs of the reference's own training/misc.py
(imported in the build container by tests/golden/make_golden.py) on seeded inputs.
"""
import numpy as np
import torch


def adjust_dynamic_range(data, drange_in, drange_out):
    if drange_in != drange_out:
        scale = (np.float32(drange_out[1]) - np.float32(drange_out[0])) / (np.float32(drange_in[1]) - np.float32(drange_in[0]))
        bias = (np.float32(drange_out[0]) - np.float32(drange_in[0]) * scale)
        data = data * scale + bias
    return data


def normalize_np(v):
    return v / np.sqrt(np.sum(np.square(v), axis=-1, keepdims=True))


def slerp_np(a, b, t):
    a = normalize_np(a)
    b = normalize_np(b)
    d = np.sum(a * b, axis=-1, keepdims=True)
    p = t * np.arccos(d)
    c = normalize_np(b - d * a)
    d = a * np.cos(p) + c * np.sin(p)
    return normalize_np(d)


def lerp(a, b, t):
    return a + (b - a) * t


def normalize_t(v):
    return v / torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True))


def slerp_t(a, b, t):
    a = normalize_t(a)
    b = normalize_t(b)
    d = torch.sum(a * b, dim=-1, keepdim=True)
    p = t * torch.acos(d)
    c = normalize_t(b - d * a)
    d = a * torch.cos(p) + c * torch.sin(p)
    return normalize_t(d)


class Tape:
    """Replays recorded random draws [(kind, ndarray), ...] in order (kinds: normal/uniform/randint)."""

    def __init__(self, entries, dtype=torch.float32):
        self.entries = list(entries)
        self.pos = 0
        self.dtype = dtype

    def _next(self, kind):
        k, v = self.entries[self.pos]
        self.pos += 1
        assert k == kind, 'tape draw %d is %s, wanted %s' % (self.pos - 1, k, kind)
        return v

    def normal(self, shape):
        v = torch.as_tensor(np.asarray(self._next('normal'))).to(self.dtype)
        assert list(v.shape) == [int(s) for s in shape], (v.shape, shape)
        return v

    def uniform(self, shape):
        v = torch.as_tensor(np.asarray(self._next('uniform'))).to(self.dtype)
        assert list(v.shape) == [int(s) for s in shape], (v.shape, shape)
        return v

    def randint(self, low, high):
        return int(np.asarray(self._next('randint')))


class SeededRandom:
    """Fresh draws from a seeded torch generator (for the CPU baseline timing and self-tests)."""

    def __init__(self, seed=0, dtype=torch.float32):
        self.g = torch.Generator().manual_seed(seed)
        self.dtype = dtype

    def normal(self, shape):
        return torch.randn([int(s) for s in shape], generator=self.g, dtype=self.dtype)

    def uniform(self, shape):
        return torch.rand([int(s) for s in shape], generator=self.g, dtype=self.dtype)

    def randint(self, low, high):
        return int(torch.randint(low, high, (), generator=self.g))
