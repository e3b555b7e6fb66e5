"""StyleGAN2 generator / discriminator on the HIP path.

Same build-function names, kwargs and return arity as the reference's
`training/networks_stylegan2.py` (G_main :151-245, G_mapping :252-304, G_synthesis_stylegan2
:311-401, D_stylegan2_feature :408-507 and the layer helpers :22-144), so a config dict can point
`func_name` at `inclusivegan_amd.training.networks_stylegan2.<name>`; variable names and HWIO weight
layouts match the reference's (checkpoint interchange).

What is different by design (MI355X-first):
  * tensors are torch/ROCm, activations channels_last (physically NHWC);
  * `modulated_conv2d_layer` always evaluates the mathematically identical *non-fused* form
    (networks_stylegan2.py:112,126): one shared-weight implicit GEMM with M = N*H*W on the f32
    MFMA, modulation folded into the operand load and demodulation into the epilogue, instead of
    the reference's per-sample grouped convolution (:108-110).  The demodulation coefficients
    d[b,o] = rsqrt(sum_i (sum_k w[k,i,o]^2) s[b,i]^2 + 1e-8) are a tiny [N,Cin]x[Cin,Cout] product;
  * dense layers run on the same GEMM kernel as 1x1 convolutions;
  * random draws (noise, style mixing) go through tflib.tfutil's injectable source.
"""
import numpy as np
import torch

from .. import hip_ops
from ..dnnlib.util import EasyDict
from ..dnnlib import tflib
from ..dnnlib.tflib import tfutil
from ..dnnlib.tflib.tfutil import variable_scope, get_variable
from ..dnnlib.tflib.ops.upfirdn_2d import upsample_2d, downsample_2d, upsample_conv_2d, conv_downsample_2d
from ..dnnlib.tflib.ops.fused_bias_act import fused_bias_act

#----------------------------------------------------------------------------
# Get/create weight tensor for a convolution or fully-connected layer (:22-36).

def get_weight(shape, gain=1, use_wscale=True, lrmul=1, weight_var='weight', init_mul=1.0):
    fan_in = np.prod(shape[:-1]) # [kernel, kernel, fmaps_in, fmaps_out] or [in, out]
    he_std = gain / np.sqrt(fan_in) # He init
    if use_wscale:
        init_std = 1.0 / lrmul
        runtime_coef = he_std * lrmul
    else:
        init_std = he_std / lrmul
        runtime_coef = lrmul
    w = get_variable(weight_var, shape=shape, initializer=('normal', init_std * init_mul))
    # The reference returns w * runtime_coef (:36).  Every consumer here is a GEMM / convolution, which is
    # linear in w, so the coefficient rides along as the kernels' output multiplier (ConvGeom.alpha,
    # igan_conv2d_params.alpha) -- forward, data gradient and weight gradient alike -- and no scaled copy
    # of the weights (nor its gradient pass) exists.
    return w, float(runtime_coef)

#----------------------------------------------------------------------------
# Fully-connected layer (:41-46).

def dense_layer(x, fmaps, gain=1, use_wscale=True, lrmul=1, weight_var='weight', init_mul=1.0):
    if x.dim() > 2:
        x = x.reshape(x.shape[0], -1)   # logical NCHW flatten order, like tf.reshape on NCHW
    w, coef = get_weight([int(x.shape[1]), fmaps], gain=gain, use_wscale=use_wscale, lrmul=lrmul, weight_var=weight_var, init_mul=init_mul)
    return hip_ops.matmul(x, w, alpha=coef)

#----------------------------------------------------------------------------
# Convolution layer with optional upsampling or downsampling (:51-61).

def conv2d_layer(x, fmaps, kernel, up=False, down=False, resample_kernel=None, gain=1, use_wscale=True, lrmul=1, weight_var='weight', init_mul=1.0):
    assert not (up and down)
    assert kernel >= 1 and kernel % 2 == 1
    w, coef = get_weight([kernel, kernel, int(x.shape[1]), fmaps], gain=gain, use_wscale=use_wscale, lrmul=lrmul, weight_var=weight_var, init_mul=init_mul)
    if up:
        x = upsample_conv_2d(x, w, data_format='NCHW', k=resample_kernel, gain=coef)     # gain scales the FIR taps: same product
    elif down:
        x = conv_downsample_2d(x, w, data_format='NCHW', k=resample_kernel, gain=coef)
    else:
        p = (kernel - 1) // 2   # SAME, stride 1, odd kernel
        x = hip_ops.conv2d(x, w, hip_ops.ConvGeom(kernel, kernel, 1, 1, p, p, coef), (int(x.shape[2]), int(x.shape[3])))
    return x

#----------------------------------------------------------------------------
# Apply bias and activation func (:66-68).

def apply_bias_act(x, act='linear', alpha=None, gain=None, lrmul=1, bias_var='bias', noise=None, noise_strength=None):
    b = get_variable(bias_var, shape=[int(x.shape[1])], initializer=('zeros',))
    if lrmul != 1:
        b = b * float(lrmul)
    if noise is not None:
        from ..dnnlib.tflib.ops.fused_bias_act import activation_funcs
        spec = activation_funcs[act]
        if spec.hip_idx in (1, 2, 3):
            a = alpha if alpha is not None else (spec.def_alpha or 0.0)
            return hip_ops.bias_act_noise(x, b, noise, noise_strength, spec.hip_idx, a, spec.def_gain if gain is None else gain)
        x = x + noise * noise_strength
    return fused_bias_act(x, b=b, act=act, alpha=alpha, gain=gain)

#----------------------------------------------------------------------------
# conv2d_layer followed by apply_bias_act (:66-68 after :51-61), the pair every discriminator convolution is
# (`apply_bias_act(conv2d_layer(x, ...), act=act)`): same variables in the same order ('weight', then 'bias'), with
# the bias / activation evaluated in the convolution's epilogue where the kernels allow it (no FIR after the
# convolution, more than 4 input channels, piecewise-linear activation).

def conv2d_bias_act_layer(x, fmaps, kernel, down=False, resample_kernel=None, act='linear', alpha=None, gain=None, init_mul=1.0):
    from ..dnnlib.tflib.ops.fused_bias_act import activation_funcs
    spec = activation_funcs[act]
    if not hip_ops.conv_bias_act_fusable(x, fmaps, spec.hip_idx):
        return apply_bias_act(conv2d_layer(x, fmaps=fmaps, kernel=kernel, down=down, resample_kernel=resample_kernel, init_mul=init_mul),
                              act=act, alpha=alpha, gain=gain)
    w, coef = get_weight([kernel, kernel, int(x.shape[1]), fmaps], init_mul=init_mul)
    b = get_variable('bias', shape=[fmaps], initializer=('zeros',))
    a = alpha if alpha is not None else (spec.def_alpha or 0.0)
    g = spec.def_gain if gain is None else gain
    if down:
        # conv_downsample_2d (upfirdn_2d.py:296-332): FIR with gain = coef, then the VALID stride-2 convolution
        from ..dnnlib.tflib.ops.upfirdn_2d import _setup_kernel, _simple_upfirdn_2d
        k = _setup_kernel(resample_kernel if resample_kernel is not None else [1, 1]) * coef
        p = (k.shape[0] - 2) + (kernel - 1)
        x = _simple_upfirdn_2d(x, k, pad0=(p+1)//2, pad1=p//2, data_format='NCHW')
        H, W = int(x.shape[2]), int(x.shape[3])
        geom = hip_ops.ConvGeom(kernel, kernel, 2, 1, 0, 0)
        return hip_ops.ConvBiasActFn.apply(x, w, b, geom, ((H - kernel) // 2 + 1, (W - kernel) // 2 + 1), spec.hip_idx, a, g)
    pd = (kernel - 1) // 2
    geom = hip_ops.ConvGeom(kernel, kernel, 1, 1, pd, pd, coef)
    return hip_ops.ConvBiasActFn.apply(x, w, b, geom, (int(x.shape[2]), int(x.shape[3])), spec.hip_idx, a, g)

#----------------------------------------------------------------------------
# Naive upsampling (nearest neighbor) and downsampling (average pooling) (:73-84).  Not used by config-e/f
# (the resample_kernel paths above are); kept for the operator surface.

def naive_upsample_2d(x, factor=2):
    n, c, h, w = x.shape
    return x.reshape(n, c, h, 1, w, 1).expand(n, c, h, factor, w, factor).reshape(n, c, h * factor, w * factor)

def naive_downsample_2d(x, factor=2):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // factor, factor, w // factor, factor).mean(dim=(3, 5))

#----------------------------------------------------------------------------
# Styles of a whole synthesis pass.  s = A(w_lat) + b + 1 and d = rsqrt(s^2 . sum_kk w^2 + 1e-8) of every modulated layer
# (:99-107) depend on the latents and the weights only, so G_synthesis computes them for all its layers up front in grouped
# launches (hip_ops.style_mod_all) and the layers pick theirs up by variable scope.

_STYLE_CACHE = []       # stack of {scope: (s, d)} of the synthesis passes being built

def _precompute_styles(specs, dlatents, init_mul):
    """specs: list of (scope path, layer index, cin, fmaps, kernel, demodulate) in call order."""
    layers = []
    for scope, idx, cin, fmaps, kernel, demod in specs:
        names = scope.split('/')
        ctxs = [variable_scope(n) for n in names]
        for c in ctxs:
            c.__enter__()
        try:
            w, coef = get_weight([kernel, kernel, cin, fmaps], init_mul=init_mul)
            a_w, a_coef = get_weight([int(dlatents[idx].shape[1]), cin], weight_var='mod_weight', init_mul=init_mul)
            a_b = get_variable('mod_bias', shape=[cin], initializer=('zeros',))
        finally:
            for c in reversed(ctxs):
                c.__exit__(None, None, None)
        layers.append(dict(scope=scope, y=dlatents[idx], a_w=a_w, a_b=a_b, w=w, wsq=None, c_a=a_coef, c_w=coef, demodulate=demod))
    if any(hip_ops._is_meta(l['y']) or not hip_ops.style_mod_fusable(l['y'], l['a_w'], l['w'], l['demodulate']) for l in layers):
        return None
    dem = [l for l in layers if l['demodulate']]
    if dem:
        def _sumsq_many(missing):
            with torch.no_grad():
                return hip_ops.sumsq_taps_grouped_raw([dem[i]['w'] for i in missing])
        for l, v in zip(dem, tfutil.derived_many([l['scope'] + '/weight:sumsq' for l in dem], _sumsq_many)):
            l['wsq'] = v
    res = hip_ops.style_mod_all(layers)
    if res is None:
        if hip_ops._second_order_depth > 0:       # path-length step: the differentiable composites, with what is common to the layers done once
            b1 = hip_ops.style_bias_plus_one([l['a_b'] for l in layers])
            res = [hip_ops.style_mod_composite(l['y'], l['a_w'], l['a_b'], l['w'], l['c_a'], l['c_w'], l['demodulate'], b1=b1[i]) for i, l in enumerate(layers)]
        else:
            return None
    return {l['scope']: sd for l, sd in zip(layers, res)}

#----------------------------------------------------------------------------
# Modulated convolution layer (:89-127).

def modulated_conv2d_layer(x, y, fmaps, kernel, up=False, down=False, demodulate=True, resample_kernel=None, gain=1, use_wscale=True, lrmul=1, fused_modconv=True, weight_var='weight', mod_weight_var='mod_weight', mod_bias_var='mod_bias', init_mul=1.0, epilogue=None):
    """`epilogue=dict(act=, noise=)` (stride-1 and up-sampling layers) asks for the whole synthesis layer -- convolution, noise * noise_strength,
    bias, activation (:349-357), variables created in the reference's order -- from one kernel where that form applies
    (hip_ops.ModConvBanFn), from the convolution + the fused epilogue pass otherwise."""
    assert not (up and down)
    assert kernel >= 1 and kernel % 2 == 1
    del fused_modconv  # both settings are the same function; this path always runs the non-fused algebra

    # Get weight.
    cin = int(x.shape[1])
    w, coef = get_weight([kernel, kernel, cin, fmaps], gain=gain, use_wscale=use_wscale, lrmul=lrmul, weight_var=weight_var, init_mul=init_mul)

    # Modulate: s = A(y) + b + 1 (:99-101); demodulate: d[b,o] = rsqrt(sum_{k,k,i} (w*s)^2 + 1e-8)
    # = rsqrt((s^2) @ (sum_kk w^2) + 1e-8) (:105-107) -- both in hip_ops.style_mod (two launches).
    pre = _STYLE_CACHE[-1].get(tfutil.current_scope()) if _STYLE_CACHE else None
    if pre is not None:
        s, d = pre                      # computed for all layers of this synthesis pass in grouped launches
    else:
        a_w, a_coef = get_weight([int(y.shape[1]), cin], weight_var=mod_weight_var, init_mul=init_mul)
        a_b = get_variable(mod_bias_var, shape=[cin], initializer=('zeros',))
        wsq = None
        if demodulate and not hip_ops._is_meta(y) and hip_ops.style_mod_fusable(y, a_w, w, True):
            def _sumsq():
                with torch.no_grad():
                    return hip_ops.sumsq_taps_raw(w)                   # [I,O] of the raw weights, once per training op
            wsq = tfutil.derived(weight_var + ':sumsq', _sumsq)
        s, d = hip_ops.style_mod(y, a_w, a_b, w, wsq, a_coef, coef, demodulate)

    # Convolution with optional up/downsampling; scales folded into the kernel.
    H, W = int(x.shape[2]), int(x.shape[3])
    if up:
        geom = hip_ops.ConvGeom(kernel, kernel, 1, 2, kernel - 1, kernel - 1, coef)
        out_hw = ((H - 1) * 2 + kernel, (W - 1) * 2 + kernel)
        x = hip_ops.ModConv2dFn.apply(x, w, s, d, geom, out_hw)
        # FIR after the transposed conv (upfirdn_2d.py:272-273,292)
        from ..dnnlib.tflib.ops.upfirdn_2d import _setup_kernel, _simple_upfirdn_2d
        k = _setup_kernel(resample_kernel if resample_kernel is not None else [1, 1]) * 4.0
        p = (k.shape[0] - 2) - (kernel - 1)
        if epilogue is not None:
            # the FIR and the layer epilogue (noise, bias, activation) as one pass over the up-sampled tensor
            from ..dnnlib.tflib.ops.fused_bias_act import activation_funcs
            spec = activation_funcs[epilogue['act']]
            noise = epilogue['noise']
            noise_strength = get_variable('noise_strength', shape=[], initializer=('zeros',))
            b = get_variable('bias', shape=[fmaps], initializer=('zeros',))
            if hip_ops.fir_ban_fusable(x, k, spec.hip_idx):
                return hip_ops.FirBanFn.apply(x, k, (p+1)//2+2-1, p//2+1, b, noise, noise_strength, spec.hip_idx, spec.def_alpha or 0.0, spec.def_gain)
            x = _simple_upfirdn_2d(x, k, pad0=(p+1)//2+2-1, pad1=p//2+1, data_format='NCHW')
            if spec.hip_idx in (1, 2, 3):
                return hip_ops.bias_act_noise(x, b, noise, noise_strength, spec.hip_idx, spec.def_alpha or 0.0, spec.def_gain)
            return fused_bias_act(x + noise * noise_strength, b=b, act=epilogue['act'])
        x = _simple_upfirdn_2d(x, k, pad0=(p+1)//2+2-1, pad1=p//2+1, data_format='NCHW')
    elif down:
        from ..dnnlib.tflib.ops.upfirdn_2d import _setup_kernel, _simple_upfirdn_2d
        k = _setup_kernel(resample_kernel if resample_kernel is not None else [1, 1])
        p = (k.shape[0] - 2) + (kernel - 1)
        # the FIR commutes with the per-channel input scale, so modulation stays inside the conv
        x = _simple_upfirdn_2d(x, k, pad0=(p+1)//2, pad1=p//2, data_format='NCHW')
        H, W = int(x.shape[2]), int(x.shape[3])
        geom = hip_ops.ConvGeom(kernel, kernel, 2, 1, 0, 0, coef)
        x = hip_ops.ModConv2dFn.apply(x, w, s, d, geom, ((H - kernel) // 2 + 1, (W - kernel) // 2 + 1))
    else:
        p = (kernel - 1) // 2
        if epilogue is not None:
            from ..dnnlib.tflib.ops.fused_bias_act import activation_funcs
            spec = activation_funcs[epilogue['act']]
            noise = epilogue['noise']
            noise_strength = get_variable('noise_strength', shape=[], initializer=('zeros',))
            b = get_variable('bias', shape=[fmaps], initializer=('zeros',))
            if hip_ops.modconv_ban_fusable(x, w, d, b, noise, spec.hip_idx):
                return hip_ops.ModConvBanFn.apply(x, w, s, d, b, noise.contiguous(), noise_strength, hip_ops.ConvGeom(kernel, kernel, 1, 1, p, p, coef), (H, W),
                                                  spec.hip_idx, spec.def_alpha or 0.0, spec.def_gain)
            x = hip_ops.ModConv2dFn.apply(x, w, s, d, hip_ops.ConvGeom(kernel, kernel, 1, 1, p, p, coef), (H, W))
            if spec.hip_idx in (1, 2, 3):
                return hip_ops.bias_act_noise(x, b, noise, noise_strength, spec.hip_idx, spec.def_alpha or 0.0, spec.def_gain)
            return fused_bias_act(x + noise * noise_strength, b=b, act=epilogue['act'])
        x = hip_ops.ModConv2dFn.apply(x, w, s, d, hip_ops.ConvGeom(kernel, kernel, 1, 1, p, p, coef), (H, W))
    return x

#----------------------------------------------------------------------------
# Minibatch standard deviation layer (:132-144).

def minibatch_stddev_layer(x, group_size=6, num_new_features=1):
    if num_new_features != 1:
        raise NotImplementedError('minibatch_stddev_layer: only num_new_features=1 is built (the only value the configs use)')
    n = int(x.shape[0])
    g = min(group_size, n)
    if n % g != 0:
        raise ValueError('minibatch must be divisible by (or smaller than) group_size')
    return hip_ops.MbStdFn.apply(x, g)

#----------------------------------------------------------------------------
# Main generator network (:151-245).

def G_main(
    latents_in,                                         # First input: Latent vectors (Z) [minibatch, latent_size].
    labels_in,                                          # Second input: Conditioning labels [minibatch, label_size].
    truncation_psi          = 0.6,
    truncation_cutoff       = None,
    truncation_psi_val      = None,
    truncation_cutoff_val   = None,
    dlatent_avg_beta        = 0.995,
    style_mixing_prob       = 0.9,
    is_training             = False,
    is_validation           = False,
    return_dlatents         = False,
    is_template_graph       = False,
    components              = None,
    mapping_func            = 'G_mapping',
    synthesis_func          = 'G_synthesis_stylegan2',
    init_mul                = 1.0,
    num_calls               = 1,        # extension: evaluate the batch as `num_calls` independent calls of
                                        # batch/num_calls samples (per-call style-mixing cutoff and dlatent_avg
                                        # update): the same function as calling G that many times, in one pass.
    **kwargs):

    # Validate arguments (:171-183).
    assert not is_training or not is_validation
    if components is None:
        components = EasyDict()
    if is_validation:
        truncation_psi = truncation_psi_val
        truncation_cutoff = truncation_cutoff_val
    if is_training or (truncation_psi is not None and truncation_psi == 1):
        truncation_psi = None
    if is_training:
        truncation_cutoff = None
    if not is_training or (dlatent_avg_beta is not None and dlatent_avg_beta == 1):
        dlatent_avg_beta = None
    if not is_training or (style_mixing_prob is not None and style_mixing_prob <= 0):
        style_mixing_prob = None

    # Setup components (:186-191).
    if 'synthesis' not in components:
        components.synthesis = tflib.Network('G_synthesis', func_name=globals()[synthesis_func], init_mul=init_mul, **kwargs)
    num_layers = components.synthesis.input_shape[1]
    dlatent_size = components.synthesis.input_shape[2]
    if 'mapping' not in components:
        components.mapping = tflib.Network('G_mapping', func_name=globals()[mapping_func], dlatent_broadcast=num_layers, init_mul=init_mul, **kwargs)

    # Setup variables (:194-195).
    lod_in = get_variable('lod', shape=[], initializer=('zeros',), trainable=False)
    dlatent_avg = get_variable('dlatent_avg', shape=[dlatent_size], initializer=('zeros',), trainable=False)
    dev = latents_in.device

    # Evaluate mapping network (:198).  With style mixing a second set of latents goes through the same network (:212-213);
    # the mapping is row-wise, so both sets are evaluated as one batch (half the launches of its 8 small layers) -- the
    # random draw of the second set moves in front of the first evaluation, which draws nothing itself.
    batch = int(latents_in.shape[0])
    assert batch % num_calls == 0
    per_call = batch // num_calls
    dlatents2 = None
    if style_mixing_prob is not None and not is_template_graph:
        latents2 = tfutil.random_normal(latents_in.shape, dev)
        both = components.mapping.get_output_for(torch.cat([latents_in, latents2], dim=0), torch.cat([labels_in, labels_in], dim=0),
                                                 is_training=is_training, **kwargs)
        dlatents, dlatents2 = both[:batch], both[batch:]
    else:
        dlatents = components.mapping.get_output_for(latents_in, labels_in, is_training=is_training, **kwargs)

    # Update moving average of W (:202-207), once per (virtual) call, in call order.
    if dlatent_avg_beta is not None and not is_template_graph:
        with torch.no_grad():
            call_avgs = dlatents[:, 0].reshape(num_calls, per_call, -1).mean(dim=1)
            for k in range(num_calls):
                dlatent_avg.copy_(tfutil.lerp(call_avgs[k], dlatent_avg, dlatent_avg_beta))

    # Perform style mixing regularization (:210-221); the coin and the cutoff are per call.
    if style_mixing_prob is not None:
        if dlatents2 is None:       # template pass: same graph, evaluated separately
            latents2 = tfutil.random_normal(latents_in.shape, dev)
            dlatents2 = components.mapping.get_output_for(latents2, labels_in, is_training=is_training, **kwargs)
        layer_idx = torch.arange(num_layers, device=dev)[None, :, None]
        cur_layers = num_layers   # lod is always 0 on this path (no progressive growing, training_loop.py:93-94)
        cutoffs = []
        for _k in range(num_calls):
            u = tfutil.random_uniform((), dev, 0.0, 1.0)
            r = tfutil.random_int(1, cur_layers, dev)
            cutoffs.append(torch.where(u < style_mixing_prob, r, torch.full_like(r, cur_layers)))
        mixing_cutoff = cutoffs[0] if num_calls == 1 else torch.stack(cutoffs).repeat_interleave(per_call)[:, None, None]
        dlatents = torch.where(layer_idx < mixing_cutoff, dlatents, dlatents2)

    # Apply truncation trick (:224-232).
    if truncation_psi is not None:
        layer_idx = np.arange(num_layers)[np.newaxis, :, np.newaxis]
        layer_psi = np.ones(layer_idx.shape, dtype=np.float32)
        if truncation_cutoff is None:
            layer_psi *= truncation_psi
        else:
            layer_psi = np.where(layer_idx < truncation_cutoff, layer_psi * truncation_psi, layer_psi)
        dlatents = tfutil.lerp(dlatent_avg, dlatents, torch.as_tensor(layer_psi, device=dev))

    # Evaluate synthesis network.
    images_out = components.synthesis.get_output_for(dlatents, is_training=is_training, force_clean_graph=is_template_graph, **kwargs)

    if return_dlatents:
        return images_out, dlatents
    return images_out

#----------------------------------------------------------------------------
# Mapping network (:252-304).

def G_mapping(
    latents_in,
    labels_in,
    latent_size             = 512,
    label_size              = 0,
    dlatent_size            = 512,
    dlatent_broadcast       = None,
    mapping_layers          = 8,
    mapping_fmaps           = 512,
    mapping_lrmul           = 0.01,
    mapping_nonlinearity    = 'lrelu',
    normalize_latents       = True,
    dtype                   = 'float32',
    init_mul                = 1.0,
    **_kwargs):

    act = mapping_nonlinearity
    assert dtype == 'float32'
    assert latents_in.shape[1] == latent_size
    x = latents_in.to(torch.float32)
    # (label conditioning is commented out in the reference, :278-284: labels are ignored.)

    # Normalize latents (:289).
    if normalize_latents:
        x = x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)

    # Mapping layers (:292-295).
    for layer_idx in range(mapping_layers):
        with variable_scope('Dense%d' % layer_idx):
            fmaps = dlatent_size if layer_idx == mapping_layers - 1 else mapping_fmaps
            x = apply_bias_act(dense_layer(x, fmaps=fmaps, lrmul=mapping_lrmul, init_mul=init_mul), act=act, lrmul=mapping_lrmul)

    # Broadcast (:298-300).
    if dlatent_broadcast is not None:
        x = x[:, None, :].expand(-1, dlatent_broadcast, -1)
    return x

#----------------------------------------------------------------------------
# StyleGAN2 synthesis network (:311-401).

def G_synthesis_stylegan2(
    dlatents_in,
    dlatent_size        = 512,
    num_channels        = 3,
    resolution          = 1024,
    fmap_base           = 16 << 10,
    fmap_decay          = 1.0,
    fmap_min            = 1,
    fmap_max            = 512,
    randomize_noise     = True,
    architecture        = 'skip',
    nonlinearity        = 'lrelu',
    dtype               = 'float32',
    resample_kernel     = [1,3,3,1],
    fused_modconv       = True,
    init_mul            = 1.0,
    **_kwargs):

    resolution_log2 = int(np.log2(resolution))
    assert resolution == 2**resolution_log2 and resolution >= 4
    def nf(stage): return int(np.clip(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_min, fmap_max))
    assert architecture in ['orig', 'skip', 'resnet']
    assert dtype == 'float32'
    act = nonlinearity
    num_layers = resolution_log2 * 2 - 2
    images_out = None
    assert tuple(dlatents_in.shape[1:]) == (num_layers, dlatent_size)
    dev = dlatents_in.device
    batch = int(dlatents_in.shape[0])
    # One unbind (backward: one stack of the per-layer gradients) instead of a select per use, whose
    # backward is a zero-filled [N, layers, 512] tensor and an add each.
    dlatents_in = dlatents_in.unbind(dim=1)

    # Noise inputs (:342-346).
    noise_inputs = []
    for layer_idx in range(num_layers - 1):
        res = (layer_idx + 5) // 2
        shape = [1, 1, 2**res, 2**res]
        noise_inputs.append(get_variable('noise%d' % layer_idx, shape=shape, initializer=('normal', 1.0), trainable=False))

    # Fresh per-call noise of every layer (:351-352): consecutive draws in layer order, filled by one generator launch.
    fresh_noise = None
    if randomize_noise:
        fresh_noise = tfutil.random_normal_many([[batch, 1, 2 ** ((l + 5) // 2), 2 ** ((l + 5) // 2)] for l in range(num_layers - 1)], dev)

    # Single convolution layer with all the bells and whistles (:349-357).
    def layer(x, layer_idx, fmaps, kernel, up=False):
        noise = fresh_noise[layer_idx] if randomize_noise else noise_inputs[layer_idx]
        # noise, bias and activation ride on the layer's last pass over the activations: the convolution's own epilogue, or -- for
        # up-sampling layers -- the FIR that follows the transposed convolution (same variables in the same order: weight,
        # [mod_weight, mod_bias], noise_strength, bias)
        return modulated_conv2d_layer(x, dlatents_in[layer_idx], fmaps=fmaps, kernel=kernel, up=up, resample_kernel=resample_kernel, fused_modconv=fused_modconv,
                                      init_mul=init_mul, epilogue=dict(act=act, noise=noise))

    # Building blocks for main layers (:360-377).
    def block(x, res): # res = 3..resolution_log2
        t = x
        with variable_scope('Conv0_up'):
            x = layer(x, layer_idx=res*2-5, fmaps=nf(res-1), kernel=3, up=True)
        with variable_scope('Conv1'):
            x = layer(x, layer_idx=res*2-4, fmaps=nf(res-1), kernel=3)
        if architecture == 'resnet':
            with variable_scope('Skip'):
                t = conv2d_layer(t, fmaps=nf(res-1), kernel=1, up=True, resample_kernel=resample_kernel, init_mul=init_mul)
                x = (x + t) * (1 / np.sqrt(2))
        return x
    def upsample(y):
        with variable_scope('Upsample'):
            return upsample_2d(y, k=resample_kernel)
    def torgb(x, y, res): # res = 2..resolution_log2
        with variable_scope('ToRGB'):
            t = apply_bias_act(modulated_conv2d_layer(x, dlatents_in[res*2-3], fmaps=num_channels, kernel=1, demodulate=False, fused_modconv=fused_modconv, init_mul=init_mul))
            return t if y is None else y + t

    # Styles of all layers in grouped launches (skipped for the template pass and where the kernels do not apply).
    styles = None
    if not hip_ops._is_meta(dlatents_in[0]) and dlatents_in[0].is_cuda:
        specs = [('4x4/Conv', 0, nf(1), nf(1), 3, True)]
        if architecture == 'skip':
            specs.append(('4x4/ToRGB', 1, nf(1), num_channels, 1, False))
        for res in range(3, resolution_log2 + 1):
            r = '%dx%d' % (2**res, 2**res)
            specs.append((r + '/Conv0_up', res*2-5, nf(res-2), nf(res-1), 3, True))
            specs.append((r + '/Conv1', res*2-4, nf(res-1), nf(res-1), 3, True))
            if architecture == 'skip' or res == resolution_log2:
                specs.append((r + '/ToRGB', res*2-3, nf(res-1), num_channels, 1, False))
        styles = _precompute_styles(specs, dlatents_in, init_mul)
    _STYLE_CACHE.append(styles or {})
    try:
        # Early layers (:380-388).
        y = None
        with variable_scope('4x4'):
            with variable_scope('Const'):
                x = get_variable('const', shape=[1, nf(1), 4, 4], initializer=('normal', 1.0))
                x = x.expand(batch, -1, -1, -1)
            with variable_scope('Conv'):
                x = layer(x, layer_idx=0, fmaps=nf(1), kernel=3)
            if architecture == 'skip':
                y = torgb(x, y, 2)

        # Main layers (:391-398).
        for res in range(3, resolution_log2 + 1):
            with variable_scope('%dx%d' % (2**res, 2**res)):
                x = block(x, res)
                if architecture == 'skip':
                    y = upsample(y)
                if architecture == 'skip' or res == resolution_log2:
                    y = torgb(x, y, res)
    finally:
        _STYLE_CACHE.pop()
    images_out = y
    return images_out

#----------------------------------------------------------------------------
# StyleGAN2 discriminator returning (scores, features) (:408-507).

def D_stylegan2_feature(
    images_in,
    labels_in,
    num_channels        = 3,
    resolution          = 1024,
    label_size          = 0,
    fmap_base           = 16 << 10,
    fmap_decay          = 1.0,
    fmap_min            = 1,
    fmap_max            = 512,
    architecture        = 'resnet',
    nonlinearity        = 'lrelu',
    mbstd_group_size    = 6,
    mbstd_num_features  = 1,
    dtype               = 'float32',
    resample_kernel     = [1,3,3,1],
    return_features     = False,        # extension: the losses discard features_out (loss.py:49,101-102), so the
                                        # concat is only materialised on request; the 2-tuple return is kept.
    **_kwargs):

    resolution_log2 = int(np.log2(resolution))
    assert resolution == 2**resolution_log2 and resolution >= 4
    def nf(stage): return int(np.clip(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_min, fmap_max))
    assert architecture in ['orig', 'skip', 'resnet']
    assert dtype == 'float32'
    act = nonlinearity
    assert tuple(images_in.shape[1:]) == (num_channels, resolution, resolution)
    images_in = images_in.to(torch.float32)
    if images_in.is_cuda:
        images_in = hip_ops.nhwc(images_in)      # one layout conversion for every consumer (FromRGB forward and weight gradient)

    # Building blocks for main layers (:438-455).
    def fromrgb(x, y, res): # res = 2..resolution_log2
        with variable_scope('FromRGB'):
            t = apply_bias_act(conv2d_layer(y, fmaps=nf(res-1), kernel=1), act=act)
            return t if x is None else x + t
    def block(x, res): # res = 2..resolution_log2
        t = x
        with variable_scope('Conv0'):
            x = conv2d_bias_act_layer(x, fmaps=nf(res-1), kernel=3, act=act)
        if architecture == 'resnet':
            # (x + t) * (1 / sqrt(2))  (:452-455) with the factor carried by the two branches' own multipliers -- the
            # activation gain of Conv1_down and the equalised-LR coefficient of Skip (a FIR tap scale) -- so that neither the
            # forward nor the backward spends a pass over the activations on it.
            from ..dnnlib.tflib.ops.fused_bias_act import activation_funcs
            c = 1 / np.sqrt(2)
            with variable_scope('Conv1_down'):
                x = conv2d_bias_act_layer(x, fmaps=nf(res-2), kernel=3, down=True, resample_kernel=resample_kernel, act=act,
                                          gain=float(activation_funcs[act].def_gain * c))
            with variable_scope('Skip'):
                t = conv2d_layer(t, fmaps=nf(res-2), kernel=1, down=True, resample_kernel=resample_kernel, gain=c)
                x = x + t
            return x
        with variable_scope('Conv1_down'):
            x = conv2d_bias_act_layer(x, fmaps=nf(res-2), kernel=3, down=True, resample_kernel=resample_kernel, act=act)
        return x
    def downsample(y):
        with variable_scope('Downsample'):
            return downsample_2d(y, k=resample_kernel)

    feats = []
    def feature_concat(x): # :457-461
        if return_features:
            length = int(np.prod(x.shape[1:]))
            feats.append((x / np.sqrt(np.float32(length))).reshape(-1, length))

    # Main layers (:463-476).
    x = None
    y = images_in
    feature_concat(y)
    for res in range(resolution_log2, 2, -1):
        with variable_scope('%dx%d' % (2**res, 2**res)):
            if architecture == 'skip' or res == resolution_log2:
                x = fromrgb(x, y, res)
                feature_concat(x)
            x = block(x, res)
            feature_concat(x)
            if architecture == 'skip':
                y = downsample(y)

    # Final layers (:479-490).
    with variable_scope('4x4'):
        if architecture == 'skip':
            x = fromrgb(x, y, 2)
        if mbstd_group_size > 1:
            with variable_scope('MinibatchStddev'):
                x = minibatch_stddev_layer(x, mbstd_group_size, mbstd_num_features)
        with variable_scope('Conv'):
            x = conv2d_bias_act_layer(x, fmaps=nf(1), kernel=3, act=act)
            feature_concat(x)
        with variable_scope('Dense0'):
            x = apply_bias_act(dense_layer(x, fmaps=nf(0)), act=act)
            feature_concat(x)

    # Output layer (:493-496).
    with variable_scope('Output'):
        x = apply_bias_act(dense_layer(x, fmaps=1))
        feature_concat(x)
    scores_out = x.squeeze(1)
    features_out = torch.cat(feats, dim=1) if return_features else None
    return scores_out, features_out

#----------------------------------------------------------------------------
