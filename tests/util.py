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


def gloss_tape_in_reference_order(entries, B, calls=4):
    """The HIP G loss draws [interp factors, random latents] and then runs ONE generator pass for the four
    reference calls (G_main num_calls=4: latents2 for all 4B samples, a (coin, cutoff) pair per call, one
    noise tensor of 4B samples per layer).  The oracle makes the reference's four sequential calls
    (loss.py:25,26,39,48), so the recorded draws are re-sliced into that order:
    call 1, call 2, interp factors, call 3, random latents, call 4."""
    t, zr, l2 = entries[0], entries[1], entries[2][1]
    ur = entries[3:3 + 2 * calls]
    noises = entries[3 + 2 * calls:]
    per_call = []
    for k in range(calls):
        sl = slice(k * B, (k + 1) * B)
        per_call.append([('normal', l2[sl]), ur[2 * k], ur[2 * k + 1]] + [('normal', n[1][sl]) for n in noises])
    return per_call[0] + per_call[1] + [t] + per_call[2] + [zr] + per_call[3]
