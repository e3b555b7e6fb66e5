"""Host-side helpers the hot loop uses from the reference's `training/misc.py`:
adjust_dynamic_range (:36-41) and the NumPy slerp / normalize pair (:190-203) that perturbs the
matched IMLE latents (training_loop.py:447).  Pickle / image-grid / resume helpers are snapshot
cosmetics and out of scope."""
import numpy as np


def adjust_dynamic_range(data, drange_in, drange_out):
    """Affine map of `data` from drange_in to drange_out, computed with fp32 scale / bias."""
    if drange_in != drange_out:
        lo_in, hi_in = np.float32(drange_in[0]), np.float32(drange_in[1])
        lo_out, hi_out = np.float32(drange_out[0]), np.float32(drange_out[1])
        scale = (hi_out - lo_out) / (hi_in - lo_in)
        bias = lo_out - lo_in * scale
        data = data * scale + bias
    return data


def normalize(v):
    return v / np.sqrt(np.sum(np.square(v), axis=-1, keepdims=True))


def slerp(a, b, t):
    """Spherical interpolation of a batch of vectors; the result is unit-norm."""
    a = normalize(a)
    b = normalize(b)
    d = np.sum(a * b, axis=-1, keepdims=True)
    p = t * np.arccos(d)
    c = normalize(b - d * a)
    d = a * np.cos(p) + c * np.sin(p)
    return normalize(d)
