"""Running means of training scalars, standing in for `dnnlib/tflib/autosummary.py:45-74`
(`autosummary(name, value)` accumulates [count, sum] and ignores non-finite values, :64).
Accumulation happens on the device without host synchronisation, as in-place adds into
accumulators that persist across flushes -- so the adds can be captured into a hipGraph and replayed
(the accumulator must exist before capture: the loop runs its first iterations eagerly).
`flush()` (called once per tick, training_loop.py:523) reads the means back and zeroes in place."""
import ctypes
from collections import OrderedDict

import torch

_acc = OrderedDict()   # name -> tensor [count, sum] (float64 on the value's device)


def autosummary(name, value):
    """Record `value` (python scalar or tensor of any shape) and pass it through."""
    if torch.is_tensor(value) and value.is_cuda and value.dtype == torch.float32:
        # device path: one launch adds [number of finite values, their sum] into the persistent accumulator
        from ... import _abi
        v = value.detach().reshape(-1)
        v = v if v.is_contiguous() else v.contiguous()
        acc = _acc.get(name)
        if acc is None or acc.device != v.device:
            acc = torch.zeros(2, dtype=torch.float64, device=v.device)
            _acc[name] = acc
        _abi.check(_abi.get_plugin().igan_summary_accumulate(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(v.data_ptr()),
                                                               int(v.numel()), ctypes.c_void_p(acc.data_ptr())))
    elif torch.is_tensor(value):
        v = value.detach().to(torch.float64).reshape(-1)
        ok = torch.isfinite(v)
        cnt = ok.sum().to(torch.float64)
        tot = torch.where(ok, v, torch.zeros_like(v)).sum()
        upd = torch.stack([cnt, tot])
        if name in _acc and _acc[name].device == upd.device:
            _acc[name] += upd
        else:
            _acc[name] = upd
    else:
        v = float(value)
        upd = torch.tensor([1.0, v], dtype=torch.float64)
        _acc[name] = _acc[name] + upd if name in _acc and _acc[name].device == upd.device else upd
    return value


def flush():
    """-> OrderedDict name -> mean; zeroes the accumulators in place."""
    out = OrderedDict()
    for name, t in _acc.items():
        c, s = t.cpu().tolist()
        if c > 0:
            out[name] = s / c
        t.zero_()
    return out
