"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement (PyTorch-CPU, NCHW, fp32 or fp64) of the reference's StyleGAN2 networks,
`training/networks_stylegan2.py`: get_weight :22-36, dense_layer :41-46, conv2d_layer :51-61,
apply_bias_act :66-68, modulated_conv2d_layer :89-127 (BOTH the fused grouped-conv form :108-110
and the non-fused form :112,126), minibatch_stddev_layer :132-144, G_main :151-245,
G_mapping :252-304, G_synthesis_stylegan2 :311-401, D_stylegan2_feature :408-507.

Functional style: `params` is a dict  local variable name -> tensor  using the reference's names
and HWIO / [in,out] layouts (e.g. 'G_synthesis/64x64/Conv0_up/mod_weight'); random draws come from
an explicit `rand` object with .normal(shape) / .uniform(shape) / .randint(low, high) in the
reference's call order (noise :352, style-mix latents :212, coin :218, cutoff :219).

PINNED (round 3) to the reference's own code: tests/golden/ref_ops_golden.npz holds the outputs of training/networks_stylegan2.py
EXECUTED with a NumPy stand-in for the TF primitives (tests/golden/make_ref_ops_golden.py, np_tf.py) -- every layer function, G_main in
training / validation / fixed-noise mode, D, architectures orig / skip / resnet -- and tests/test_ref_ops_golden.py requires this file
to reproduce them to 1e-10.  Unpinned remainder: the TF primitives' own semantics (np_tf.py lists them) and the backward pass
(tf.gradients), which is checked by fp64 gradcheck and the fused == non-fused identity (tests/test_oracle_networks.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import upfirdn_2d as U
from .fused_bias_act import fused_bias_act


class Scope:
    """Tiny stand-in for tf.variable_scope + tf.get_variable over a params dict."""

    def __init__(self, params, prefix=''):
        self.params = params
        self.prefix = prefix

    def sub(self, name):
        return Scope(self.params, self.prefix + name + '/')

    def get(self, name):
        return self.params[self.prefix + name]


def get_weight(sc, shape, gain=1, use_wscale=True, lrmul=1, weight_var='weight'):
    fan_in = np.prod(shape[:-1])
    he_std = gain / np.sqrt(fan_in)
    runtime_coef = he_std * lrmul if use_wscale else lrmul
    w = sc.get(weight_var)
    assert list(w.shape) == list(shape), (sc.prefix + weight_var, w.shape, shape)
    return w * float(runtime_coef)


def dense_layer(sc, x, fmaps, gain=1, use_wscale=True, lrmul=1, weight_var='weight'):
    if x.dim() > 2:
        x = x.reshape(x.shape[0], -1)
    w = get_weight(sc, [x.shape[1], fmaps], gain=gain, use_wscale=use_wscale, lrmul=lrmul, weight_var=weight_var)
    return x @ w


def conv2d_layer(sc, x, fmaps, kernel, up=False, down=False, resample_kernel=None, gain=1, use_wscale=True, lrmul=1, weight_var='weight'):
    assert not (up and down)
    w = get_weight(sc, [kernel, kernel, x.shape[1], fmaps], gain=gain, use_wscale=use_wscale, lrmul=lrmul, weight_var=weight_var)
    if up:
        return U.upsample_conv_2d(x, w, k=resample_kernel)
    if down:
        return U.conv_downsample_2d(x, w, k=resample_kernel)
    return U.conv2d_same(x, w)


def apply_bias_act(sc, x, act='linear', alpha=None, gain=None, lrmul=1, bias_var='bias'):
    b = sc.get(bias_var) * lrmul
    return fused_bias_act(x, b=b, act=act, alpha=alpha, gain=gain)


def modulated_conv2d_layer(sc, x, y, fmaps, kernel, up=False, down=False, demodulate=True, resample_kernel=None,
                           fused_modconv=True, weight_var='weight', mod_weight_var='mod_weight', mod_bias_var='mod_bias'):
    assert not (up and down)
    w = get_weight(sc, [kernel, kernel, x.shape[1], fmaps], weight_var=weight_var)
    ww = w[None]                                                         # [BkkIO] :95
    s = dense_layer(sc, y, fmaps=x.shape[1], weight_var=mod_weight_var)  # [BI] :98
    s = apply_bias_act(sc, s, bias_var=mod_bias_var) + 1                 # :99
    ww = ww * s[:, None, None, :, None]                                  # :100
    if demodulate:
        d = torch.rsqrt(torch.sum(ww * ww, dim=[1, 2, 3]) + 1e-8)        # [BO] :104
        ww = ww * d[:, None, None, None, :]                              # :105
    n, cin, h, wd = x.shape
    if fused_modconv:
        x = x.reshape(1, -1, h, wd)                                      # :109
        w = ww.permute(1, 2, 3, 0, 4).reshape(ww.shape[1], ww.shape[2], ww.shape[3], -1)  # :110
    else:
        x = x * s[:, :, None, None]                                      # :112
    if up:
        x = U.upsample_conv_2d(x, w, k=resample_kernel)
    elif down:
        x = U.conv_downsample_2d(x, w, k=resample_kernel)
    else:
        if fused_modconv:
            x = F.conv2d(x, w.reshape(kernel, kernel, cin, n, fmaps).permute(3, 4, 2, 0, 1).reshape(n * fmaps, cin, kernel, kernel),
                         padding=(kernel - 1) // 2, groups=n)
        else:
            x = U.conv2d_same(x, w)
    if fused_modconv:
        x = x.reshape(-1, fmaps, x.shape[2], x.shape[3])                 # :124
    elif demodulate:
        x = x * d[:, :, None, None]                                      # :126
    return x


def minibatch_stddev_layer(x, group_size=6, num_new_features=1):
    n, c, h, w = x.shape
    g = min(group_size, n)
    y = x.reshape(g, -1, num_new_features, c // num_new_features, h, w)
    y = y - y.mean(dim=0, keepdim=True)
    y = (y * y).mean(dim=0)
    y = torch.sqrt(y + 1e-8)
    y = y.mean(dim=[2, 3, 4], keepdim=True)
    y = y.mean(dim=2)
    y = y.repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)


def nf(stage, fmap_base, fmap_decay=1.0, fmap_min=1, fmap_max=512):
    return int(np.clip(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_min, fmap_max))


def G_mapping(sc, latents_in, dlatent_broadcast=None, mapping_layers=8, mapping_fmaps=512, dlatent_size=512,
              mapping_lrmul=0.01, normalize_latents=True):
    x = latents_in
    if normalize_latents:
        x = x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)   # :289
    for layer_idx in range(mapping_layers):
        lsc = sc.sub('Dense%d' % layer_idx)
        fmaps = dlatent_size if layer_idx == mapping_layers - 1 else mapping_fmaps
        x = apply_bias_act(lsc, dense_layer(lsc, x, fmaps=fmaps, lrmul=mapping_lrmul), act='lrelu', lrmul=mapping_lrmul)
    if dlatent_broadcast is not None:
        x = x[:, None, :].repeat(1, dlatent_broadcast, 1)                    # :300
    return x


def G_synthesis_stylegan2(sc, dlatents_in, rand, resolution=1024, num_channels=3, fmap_base=16 << 10, architecture='skip',
                          randomize_noise=True, resample_kernel=(1, 3, 3, 1), fused_modconv=True):
    resolution_log2 = int(np.log2(resolution))
    num_layers = resolution_log2 * 2 - 2
    act = 'lrelu'
    resample_kernel = list(resample_kernel)
    batch = dlatents_in.shape[0]

    def layer(lsc, x, layer_idx, fmaps, kernel, up=False):
        x = modulated_conv2d_layer(lsc, x, dlatents_in[:, layer_idx], fmaps=fmaps, kernel=kernel, up=up,
                                   resample_kernel=resample_kernel, fused_modconv=fused_modconv)
        if randomize_noise:
            noise = rand.normal([batch, 1, x.shape[2], x.shape[3]]).to(x.dtype)        # :352
        else:
            noise = sc.get('noise%d' % layer_idx).to(x.dtype)
        x = x + noise * lsc.get('noise_strength')                                       # :356
        return apply_bias_act(lsc, x, act=act)

    def torgb(rsc, x, y, res):
        t = apply_bias_act(rsc, modulated_conv2d_layer(rsc, x, dlatents_in[:, res * 2 - 3], fmaps=num_channels, kernel=1,
                                                       demodulate=False, fused_modconv=fused_modconv))
        return t if y is None else y + t

    y = None
    s4 = sc.sub('4x4')
    x = s4.sub('Const').get('const').repeat(batch, 1, 1, 1)                             # :383-384
    x = layer(s4.sub('Conv'), x, layer_idx=0, fmaps=nf(1, fmap_base), kernel=3)
    if architecture == 'skip':
        y = torgb(s4.sub('ToRGB'), x, y, 2)
    for res in range(3, resolution_log2 + 1):
        rsc = sc.sub('%dx%d' % (2 ** res, 2 ** res))
        t = x
        x = layer(rsc.sub('Conv0_up'), x, layer_idx=res * 2 - 5, fmaps=nf(res - 1, fmap_base), kernel=3, up=True)
        x = layer(rsc.sub('Conv1'), x, layer_idx=res * 2 - 4, fmaps=nf(res - 1, fmap_base), kernel=3)
        if architecture == 'resnet':
            t = conv2d_layer(rsc.sub('Skip'), t, fmaps=nf(res - 1, fmap_base), kernel=1, up=True, resample_kernel=resample_kernel)
            x = (x + t) * (1 / np.sqrt(2))
        if architecture == 'skip':
            y = U.upsample_2d(y, k=resample_kernel)                                     # :373,395
        if architecture == 'skip' or res == resolution_log2:
            y = torgb(rsc.sub('ToRGB'), x, y, res)
    return y


def G_main(params, latents_in, rand, resolution, num_channels=3, fmap_base=16 << 10, architecture='skip',
           is_training=False, is_validation=False, return_dlatents=False, truncation_psi=0.6, truncation_cutoff=None,
           truncation_psi_val=None, truncation_cutoff_val=None, dlatent_avg_beta=0.995, style_mixing_prob=0.9,
           fused_modconv=True, state=None, dlatent_size=512, mapping_fmaps=512, randomize_noise=True):
    """`state` (dict) receives the updated non-trainable 'dlatent_avg' (tf.assign :205)."""
    sc = Scope(params)
    if is_validation:
        truncation_psi, truncation_cutoff = truncation_psi_val, truncation_cutoff_val
    if is_training or (truncation_psi is not None and truncation_psi == 1):
        truncation_psi = None
    if is_training:
        truncation_cutoff = None
    if not is_training or (dlatent_avg_beta is not None and dlatent_avg_beta == 1):
        dlatent_avg_beta = None
    if not is_training or (style_mixing_prob is not None and style_mixing_prob <= 0):
        style_mixing_prob = None
    num_layers = int(np.log2(resolution)) * 2 - 2
    msc = sc.sub('G_mapping')
    mkw = dict(dlatent_size=dlatent_size, mapping_fmaps=mapping_fmaps)
    dlatents = G_mapping(msc, latents_in, dlatent_broadcast=num_layers, **mkw)
    if dlatent_avg_beta is not None:
        batch_avg = dlatents[:, 0].mean(dim=0).detach()
        new_avg = batch_avg + (sc.get('dlatent_avg') - batch_avg) * dlatent_avg_beta     # lerp(batch_avg, avg, beta)
        if state is not None:
            state['dlatent_avg'] = new_avg
    if style_mixing_prob is not None:
        latents2 = rand.normal(list(latents_in.shape)).to(latents_in.dtype)              # :212
        dlatents2 = G_mapping(msc, latents2, dlatent_broadcast=num_layers, **mkw)
        layer_idx = torch.arange(num_layers)[None, :, None]
        u = float(rand.uniform([]))                                                      # :218
        r = int(rand.randint(1, num_layers))                                             # :219 (evaluated by tf.cond only if taken;
        mixing_cutoff = r if u < style_mixing_prob else num_layers                       #  the product always draws it)
        dlatents = torch.where(layer_idx < mixing_cutoff, dlatents, dlatents2)           # :221
    if truncation_psi is not None:
        layer_idx = np.arange(num_layers)[np.newaxis, :, np.newaxis]
        layer_psi = np.ones(layer_idx.shape, dtype=np.float32)
        if truncation_cutoff is None:
            layer_psi *= truncation_psi
        else:
            layer_psi = np.where(layer_idx < truncation_cutoff, layer_psi * truncation_psi, layer_psi)
        avg = sc.get('dlatent_avg')
        dlatents = avg + (dlatents - avg) * torch.as_tensor(layer_psi, dtype=dlatents.dtype)   # lerp(avg, dlatents, psi)
    images = G_synthesis_stylegan2(sc.sub('G_synthesis'), dlatents, rand, resolution=resolution, num_channels=num_channels,
                                   fmap_base=fmap_base, architecture=architecture, fused_modconv=fused_modconv, randomize_noise=randomize_noise)
    if return_dlatents:
        return images, dlatents
    return images


def D_stylegan2_feature(params, images_in, resolution, num_channels=3, fmap_base=16 << 10, architecture='resnet',
                        mbstd_group_size=6, mbstd_num_features=1, resample_kernel=(1, 3, 3, 1)):
    sc = Scope(params)
    resolution_log2 = int(np.log2(resolution))
    act = 'lrelu'
    resample_kernel = list(resample_kernel)
    feats = []

    def feature_concat(x):
        length = int(np.prod(x.shape[1:]))
        feats.append((x / np.sqrt(np.float32(length))).reshape(-1, length))

    def fromrgb(fsc, x, y, res):
        t = apply_bias_act(fsc, conv2d_layer(fsc, y, fmaps=nf(res - 1, fmap_base), kernel=1), act=act)
        return t if x is None else x + t

    x = None
    y = images_in
    feature_concat(y)
    for res in range(resolution_log2, 2, -1):
        rsc = sc.sub('%dx%d' % (2 ** res, 2 ** res))
        if architecture == 'skip' or res == resolution_log2:
            x = fromrgb(rsc.sub('FromRGB'), x, y, res)
            feature_concat(x)
        t = x
        c0 = rsc.sub('Conv0')
        x = apply_bias_act(c0, conv2d_layer(c0, x, fmaps=nf(res - 1, fmap_base), kernel=3), act=act)
        c1 = rsc.sub('Conv1_down')
        x = apply_bias_act(c1, conv2d_layer(c1, x, fmaps=nf(res - 2, fmap_base), kernel=3, down=True, resample_kernel=resample_kernel), act=act)
        if architecture == 'resnet':
            t = conv2d_layer(rsc.sub('Skip'), t, fmaps=nf(res - 2, fmap_base), kernel=1, down=True, resample_kernel=resample_kernel)
            x = (x + t) * (1 / np.sqrt(2))
        feature_concat(x)
        if architecture == 'skip':
            y = U.downsample_2d(y, k=resample_kernel)
    s4 = sc.sub('4x4')
    if architecture == 'skip':
        x = fromrgb(s4.sub('FromRGB'), x, y, 2)
    if mbstd_group_size > 1:
        x = minibatch_stddev_layer(x, mbstd_group_size, mbstd_num_features)
    cs = s4.sub('Conv')
    x = apply_bias_act(cs, conv2d_layer(cs, x, fmaps=nf(1, fmap_base), kernel=3), act=act)
    feature_concat(x)
    ds = s4.sub('Dense0')
    x = apply_bias_act(ds, dense_layer(ds, x, fmaps=nf(0, fmap_base)), act=act)
    feature_concat(x)
    os_ = sc.sub('Output')
    x = apply_bias_act(os_, dense_layer(os_, x, fmaps=1))
    feature_concat(x)
    return x.squeeze(1), torch.cat(feats, dim=1)
