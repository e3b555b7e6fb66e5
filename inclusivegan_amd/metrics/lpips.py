"""LPIPS (VGG16, "lin" layers) distance as a `tflib.Network` build function.

The reference loads this network from `metrics/vgg16_zhang_perceptual.pkl`
(training/training_loop.py:195) and calls `lpips.get_output_for(images_a, images_b)` with NCHW
fp32 images in [0, 255], getting a per-sample distance `[N]` (training/loss.py:27-31,41).  The
pickle -- architecture *and* weights -- is absent from the reference tree
(.MISSING_LARGE_BLOBS:10), so this is a restatement of Zhang et al. 2018 ("The Unreasonable
Effectiveness of Deep Features as a Perceptual Metric", vgg variant):

    x <- (x / 127.5 - 1 - shift) / scale,  shift = (-.030,-.088,-.188), scale = (.458,.448,.450)
    VGG16 conv stack, features after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
    each feature unit-normalised over channels:  f / (sqrt(sum_c f^2) + 1e-10)
    d = sum_layers mean_{h,w} sum_c lin_c * (fa_c - fb_c)^2,   lin_c >= 0 (learned 1x1, no bias)

PARITY UNPINNED: neither the architecture file nor the weights exist here; weights are seeded
random (He-normal convs, |N(0,1)|/C non-negative lin weights) and all of them are non-trainable
constants, exactly like the role they play in the reference's training graph.

All 13 convolutions run on the package's f32-MFMA implicit-GEMM kernel with the fused
bias+ReLU kernel behind them; images are evaluated at native resolution (no resize on the
training path).
"""
import os

import numpy as np
import torch

from .. import hip_ops
from ..dnnlib.tflib import tfutil
from ..dnnlib.tflib.tfutil import variable_scope, get_variable
from ..dnnlib.tflib.ops.fused_bias_act import fused_bias_act

_VGG_CFG = [  # (block, [out channels...]); 2x2 max-pool between blocks
    ('conv1', [64, 64]),
    ('conv2', [128, 128]),
    ('conv3', [256, 256, 256]),
    ('conv4', [512, 512, 512]),
    ('conv5', [512, 512, 512]),
]
_SHIFT = (-0.030, -0.088, -0.188)
_SCALE = (0.458, 0.448, 0.450)


def _conv_relu(x, fmaps, name):
    with variable_scope(name):
        cin = int(x.shape[1])
        std = float(np.sqrt(2.0 / (9 * cin)))
        w = get_variable('weight', shape=[3, 3, cin, fmaps], initializer=('normal', std), trainable=False)
        if w.device.type == 'cuda':
            hip_ops.mark_constant(w)        # the perceptual network's weights never change: its filter images are written once, not 36 times per generator step
        b = get_variable('bias', shape=[fmaps], initializer=('zeros',), trainable=False)
        geom, out_hw = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1), (int(x.shape[2]), int(x.shape[3]))
        if hip_ops.conv_bias_act_fusable(x, fmaps, 2):
            return hip_ops.ConvBiasActFn.apply(x, w, b, geom, out_hw, 2, 0.0, 1.0)      # bias + relu in the conv epilogue
        x = hip_ops.conv2d(x, w, geom, out_hw)
        return fused_bias_act(x, b=b, act='relu', gain=1.0)


_const_cache = {}
_FUSED = os.environ.get('IGAN_LPIPS_FUSED', '1') != '0'      # A/B switch: pair-table distances + pool/tap Function


def _consts(dev):
    """Per-device shift / scale constants, created once (no host->device copy inside a captured step)."""
    key = str(dev)
    if key not in _const_cache:
        _const_cache[key] = (torch.tensor(_SHIFT, device=dev, dtype=torch.float32).view(1, 3, 1, 1),
                             torch.tensor(_SCALE, device=dev, dtype=torch.float32).view(1, 3, 1, 1))
    return _const_cache[key]


def vgg_features(images):
    """images: [N,3,H,W] in [0,255] -> list of the 5 raw feature maps (relu1_2 ... relu5_3); the channel
    unit-normalisation is folded into `feature_distance`."""
    dev = images.device
    shift, scale = _consts(dev)
    x = (images / 127.5 - 1.0 - shift) / scale
    x = x.contiguous(memory_format=torch.channels_last) if x.device.type != 'meta' else x
    feats = []
    for bi, (block, chans) in enumerate(_VGG_CFG):
        for li, c in enumerate(chans):
            x = _conv_relu(x, c, '%s_%d' % (block, li + 1))
        if bi + 1 == len(_VGG_CFG):
            feats.append(x)
        elif x.is_cuda and int(x.shape[2]) % 2 == 0 and int(x.shape[3]) % 2 == 0 and _FUSED:
            tap, x = hip_ops.PoolTapFn.apply(x)       # the tap and the 2x2 max-pool that feeds the next block: one backward pass
            feats.append(tap)
        else:                                         # template (meta) pass
            feats.append(x)
            x = torch.nn.functional.max_pool2d(x, 2)
    return feats


def _lin_weights(i, c, hw):
    """|lin_i| / (C * H * W): the non-negative 1x1 weights of layer i with the spatial mean folded in (H * W is a power of two on
    this path, so the result equals dividing the distance afterwards).  Constant: computed once per variable version (kept on
    the variable), never inside a stream capture."""
    lin = get_variable('lin%d/weight' % i, shape=[c], initializer=('normal', 1.0), trainable=False)     # seeded |N(0,1)|/c
    if lin.device.type == 'meta':
        return lin
    cache = getattr(lin, '_igan_lin_weights', None)
    if cache is None or cache[0] != lin._version:
        cache = (lin._version, {})
        lin._igan_lin_weights = cache
    hit = cache[1].get(hw)
    if hit is None:
        with torch.no_grad():
            hit = torch.abs(lin) / c / float(hw)
        if not (lin.is_cuda and torch.cuda.is_current_stream_capturing()):
            cache[1][hw] = hit
    return hit


def feature_distance(feats_a, feats_b):
    """sum_layers mean_hw sum_c lin_c (fa - fb)^2  -> [N]."""
    total = None
    for i, (fa, fb) in enumerate(zip(feats_a, feats_b)):
        # normalise over channels, squared difference, lin weighting, spatial mean: one fused pass
        d = hip_ops.LpipsLayerFn.apply(fa, fb, _lin_weights(i, int(fa.shape[1]), int(fa.shape[2] * fa.shape[3])))
        total = d if total is None else total + d
    return total


def pair_distances(feats_gen, feats_real, n):
    """The four distances of the G loss on features of [rec_1, rec_2, interp] (3n images) and [real_1, real_2] (2n):
    -> (d(rec_1, real_1) [n], d(rec_2, real_2) [n], d(interp, real_2) [n], d(interp, real_1) [n])  (loss.py:31,41)."""
    L = len(feats_gen)
    lins = [_lin_weights(i, int(f.shape[1]), int(f.shape[2] * f.shape[3])) for i, f in enumerate(feats_gen)]
    if feats_gen[0].device.type == 'meta':
        return tuple(torch.empty((n,), device='meta') for _ in range(4))
    if not _FUSED:      # A/B switch: the per-pair form on batch slices
        r1 = [f[:n] for f in feats_real]; r2 = [f[n:] for f in feats_real]
        g1 = [f[:n] for f in feats_gen]; g2 = [f[n:2 * n] for f in feats_gen]; gi = [f[2 * n:] for f in feats_gen]
        return (feature_distance(g1, r1), feature_distance(g2, r2), feature_distance(gi, r2), feature_distance(gi, r1))
    d = hip_ops.LpipsPairsFn.apply(*lins, *feats_gen, *feats_real, n, L)
    return d.view(4, n).unbind(0)


def vgg16_zhang_perceptual(images_a, images_b, resolution=64, **_kwargs):
    """Build function: LPIPS distance between two image batches in [0,255] -> [N]."""
    return feature_distance(vgg_features(images_a), vgg_features(images_b))


# Helpers for the G loss: evaluate each image batch's VGG features once (the reference's graph
# evaluates the interpolated image and each real twice, loss.py:31,41).
def features_of(lpips_net, images):
    with tfutil.variable_store(lpips_net):
        return vgg_features(images)


def distance_of(lpips_net, feats_a, feats_b):
    with tfutil.variable_store(lpips_net):
        return feature_distance(feats_a, feats_b)


def pair_distances_of(lpips_net, feats_gen, feats_real, n):
    with tfutil.variable_store(lpips_net):
        return pair_distances(feats_gen, feats_real, n)
