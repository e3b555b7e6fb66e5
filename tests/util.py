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


LPIPS_VGG = [('conv1', [64, 64]), ('conv2', [128, 128]), ('conv3', [256, 256, 256]), ('conv4', [512, 512, 512]), ('conv5', [512, 512, 512])]


def lpips_params_from_seed(seed):
    """VGG16 + lin weights for the LPIPS stand-in, regenerated from a seed instead of being stored (14.7 M values): NumPy's legacy
    RandomState stream is frozen, so the golden generator and the tests get the same float32 values.  Names and shapes are the
    product's / the oracle's ('conv1_1/weight' HWIO, 'conv1_1/bias', 'lin0/weight')."""
    rng = np.random.RandomState(seed)
    p = {}
    cin = 3
    for block, chans in LPIPS_VGG:
        for li, c in enumerate(chans):
            name = '%s_%d' % (block, li + 1)
            p[name + '/weight'] = (rng.standard_normal((3, 3, cin, c)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
            p[name + '/bias'] = (rng.standard_normal(c) * 0.1).astype(np.float32)
            cin = c
    for i, (_, chans) in enumerate(LPIPS_VGG):
        p['lin%d/weight' % i] = rng.standard_normal(chans[-1]).astype(np.float32)
    return p


def gloss_tape_in_product_order(entries, B, lpips_weight, calls=4):
    """Inverse of gloss_tape_in_reference_order: the draws of a REFERENCE run of the G loss's main term (four sequential generator
    calls: [latents2, coin, cutoff, noise per layer] each, the interpolation factors after call 2, the random latents after call 3;
    loss.py:25-48) re-packed into the order the HIP loss consumes them (see gloss_tape_in_reference_order).  With a weight of 0 the HIP
    loss skips the three reconstruction passes, whose draws cannot reach the result: only [random latents, call 4] are kept."""
    per = (len(entries) - 2) // calls
    assert per * calls + 2 == len(entries)
    c = [entries[0:per], entries[per:2 * per], entries[2 * per + 1:3 * per + 1], entries[3 * per + 2:4 * per + 2]]
    t, zr = entries[2 * per], entries[3 * per + 1]
    if lpips_weight == 0:
        return [zr] + c[3]
    out = [t, zr, ('normal', np.concatenate([np.asarray(k[0][1]) for k in c], axis=0))]
    for k in c:
        out += [k[1], k[2]]
    for j in range(3, per):
        out.append(('normal', np.concatenate([np.asarray(k[j][1]) for k in c], axis=0)))
    return out
