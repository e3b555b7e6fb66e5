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


def synthetic_mnist(directory, seed=5):
    """A stand-in for the MNIST training files (same names, idx headers, 60000 x 28 x 28 uint8 + 60000 labels) from a seed: digits are
    random bytes (with 0 and 255 present), labels mostly 0 and 9 so that a few hundred stacked images already span 000..999."""
    import gzip
    import os
    rng = np.random.RandomState(seed)
    imgs = rng.randint(0, 256, size=(60000, 28, 28)).astype(np.uint8)
    labels = rng.choice(10, size=60000, p=[0.41, .0225, .0225, .0225, .0225, .0225, .0225, .0225, .0225, 0.41]).astype(np.uint8)
    os.makedirs(directory, exist_ok=True)
    with gzip.open(os.path.join(directory, 'train-images-idx3-ubyte.gz'), 'wb', compresslevel=1) as f:
        f.write(b'\x00\x00\x08\x03' + (60000).to_bytes(4, 'big') + (28).to_bytes(4, 'big') + (28).to_bytes(4, 'big') + imgs.tobytes())
    with gzip.open(os.path.join(directory, 'train-labels-idx1-ubyte.gz'), 'wb', compresslevel=1) as f:
        f.write(b'\x00\x00\x08\x01' + (60000).to_bytes(4, 'big') + labels.tobytes())


def synthetic_celeba(root, n=7, seed=6):
    """n aligned-CelebA-shaped PNGs (218 x 178 RGB, names 000001.png ...) under <root>/img and the attribute file at
    <root>/celeba/Anno/list_attr_celeba.txt (count line, 40 names, '<name>.jpg' + 40 values in {-1, 1})."""
    import os
    import PIL.Image
    from inclusivegan_amd.training.imle import CELEBA_ATTRIBUTES
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'img'), exist_ok=True)
    os.makedirs(os.path.join(root, 'celeba', 'Anno'), exist_ok=True)
    lines = ['%d' % n, ' '.join(CELEBA_ATTRIBUTES)]
    for i in range(n):
        PIL.Image.fromarray(rng.randint(0, 256, size=(218, 178, 3)).astype(np.uint8)).save(os.path.join(root, 'img', '%06d.png' % (i + 1)))
        lines.append('%06d.jpg  ' % (i + 1) + ' '.join('%2d' % v for v in rng.choice([-1, 1], size=40)))
    with open(os.path.join(root, 'celeba', 'Anno', 'list_attr_celeba.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return os.path.join(root, 'img')
