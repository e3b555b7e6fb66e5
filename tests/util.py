"""Shared helpers for the parity tests."""
import numpy as np
import torch


def rel_err(a, b):
    """max|a-b| / (max|b| + 1e-30) on CPU float64."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b), dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def to_nhwc_cuda(x_nchw, device):
    """CPU NCHW tensor -> CUDA logical-NCHW channels_last fp32."""
    return x_nchw.to(torch.float32).to(device).contiguous(memory_format=torch.channels_last)
