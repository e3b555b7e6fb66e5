"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

LPIPS-VGG16 distance restated from Zhang et al. 2018 on PyTorch-CPU.  The reference's own network
(`metrics/vgg16_zhang_perceptual.pkl`, training_loop.py:195) is absent (.MISSING_LARGE_BLOBS:10):
PARITY UNPINNED.  Input contract from training/loss.py:27-31: NCHW images in [0,255], output [N].
`params`: 'convB_L/weight' HWIO, 'convB_L/bias', 'linI/weight' [C] (same names as the product).
"""
import torch
import torch.nn.functional as F

VGG_CFG = [('conv1', [64, 64]), ('conv2', [128, 128]), ('conv3', [256, 256, 256]), ('conv4', [512, 512, 512]), ('conv5', [512, 512, 512])]
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)


def vgg_features(params, images):
    dt = images.dtype
    shift = torch.tensor(SHIFT, dtype=dt).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=dt).view(1, 3, 1, 1)
    x = (images / 127.5 - 1.0 - shift) / scale
    feats = []
    for bi, (block, chans) in enumerate(VGG_CFG):
        if bi > 0:
            x = F.max_pool2d(x, 2)
        for li, _c in enumerate(chans):
            name = '%s_%d' % (block, li + 1)
            w = params[name + '/weight'].to(dt)
            b = params[name + '/bias'].to(dt)
            x = F.relu(F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=1))
        feats.append(x / (torch.sqrt(torch.sum(x * x, dim=1, keepdim=True)) + 1e-10))
    return feats


def feature_distance(params, fa, fb):
    total = None
    for i, (a, b) in enumerate(zip(fa, fb)):
        c = a.shape[1]
        lin = torch.abs(params['lin%d/weight' % i].to(a.dtype)) / c
        d = ((a - b) ** 2 * lin.view(1, c, 1, 1)).sum(dim=1).mean(dim=(1, 2))
        total = d if total is None else total + d
    return total


def lpips(params, images_a, images_b):
    return feature_distance(params, vgg_features(params, images_a), vgg_features(params, images_b))
