"""GPU: the HIP path, through the product's own functions, against golden vectors produced by EXECUTING the reference's Python
(tests/golden/make_ref_ops_golden.py: dnnlib/tflib/ops/upfirdn_2d.py, fused_bias_act.py, training/networks_stylegan2.py run with a
NumPy stand-in for the TensorFlow primitives; fixture tests/golden/ref_ops_golden.npz).  The goldens are float64; the HIP path is
fp32: tolerances 1e-5 (streaming kernels), 1e-4 (whole networks), relative max-abs."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_ops_golden.npz'), allow_pickle=False)


def dev32(a, dev):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize('i', range(int(G['upfirdn_cases'])))
def test_upfirdn_kernel_and_gradients(cuda_device, i):
    """UpFirDn2D on the HIP kernels: y, dx = the gradient the reference defines for the op, and the gradient of that."""
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    p = 'upfirdn_%d_' % i
    upx, upy, downx, downy, px0, px1, py0, py1 = [int(v) for v in G[p + 'params']]
    kw = dict(upx=upx, upy=upy, downx=downx, downy=downy, padx0=px0, padx1=px1, pady0=py0, pady1=py1)
    x = dev32(G[p + 'x'], cuda_device).requires_grad_(True)
    y = upfirdn_2d(x, G[p + 'k'], **kw)
    assert rel_err(y, G[p + 'y']) < 1e-5
    dy = dev32(G[p + 'dy'], cuda_device).requires_grad_(True)
    dx, = torch.autograd.grad(y, x, dy, create_graph=True)
    assert rel_err(dx, G[p + 'dx']) < 1e-5
    d_dy, = torch.autograd.grad(dx, dy, dev32(G[p + 'ddx'], cuda_device))
    assert rel_err(d_dy, G[p + 'd_dy']) < 1e-5


def test_resampling_wrappers(cuda_device):
    from inclusivegan_amd.dnnlib.tflib.ops import upfirdn_2d as PU
    x = dev32(G['wrapper_x'], cuda_device).contiguous(memory_format=torch.channels_last)
    done = 0
    for p in G['wrapper_cases']:
        name = p[:p.rindex('_', 0, -1)]
        k_in = G[p + 'k_in'].tolist() or None
        factor, gain = int(G[p + 'factor_gain'][0]), float(G[p + 'factor_gain'][1])
        if name in ('upsample_conv_2d', 'conv_downsample_2d'):
            y = getattr(PU, name)(x, dev32(G[p + 'w'], cuda_device), k=k_in, factor=factor, gain=gain)
        elif name == 'filter_2d':
            y = PU.filter_2d(x, k_in, gain=gain)
        else:
            y = getattr(PU, name)(x, k=k_in, factor=factor, gain=gain)
        assert rel_err(y, G[p + 'y']) < 1e-5, p
        done += 1
    assert done == len(G['wrapper_cases'])


def test_fused_bias_act_all_activations(cuda_device):
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act
    x, b = dev32(G['act_x'], cuda_device), dev32(G['act_b'], cuda_device)
    for n in [str(v) for v in G['act_names']]:
        assert rel_err(fused_bias_act(x, b, act=n), G['act_%s_default' % n]) < 1e-5, n
        assert rel_err(fused_bias_act(x, b, act=n, alpha=0.3, gain=0.7), G['act_%s_custom' % n]) < 1e-5, n
        assert rel_err(fused_bias_act(x, None, axis=3, act=n), G['act_%s_nobias_axis3' % n]) < 1e-5, n


def _network(kind, arch, dev, prefix, **extra):
    from inclusivegan_amd.dnnlib import tflib
    res, fmap = int(G['net_cfg'][0]), int(G['net_cfg'][1])
    fn = 'inclusivegan_amd.training.networks_stylegan2.' + ('G_main' if kind == 'G' else 'D_stylegan2_feature')
    net = tflib.Network(kind, func_name=fn, num_channels=3, resolution=res, label_size=0, fmap_base=fmap, architecture=arch, device=dev, seed=1, **extra)
    with torch.no_grad():
        for name, v in net.vars.items():
            v.copy_(dev32(G[prefix + name.replace('/', '.')], dev).reshape(v.shape))
    return net


@pytest.mark.parametrize('p', [str(c) for c in G['G_cases']])
def test_generator_against_reference_execution(cuda_device, p):
    """G_main on the HIP path with the reference's variables and the reference run's random draws: images, dlatents and the
    dlatent_avg update, for every architecture and mode the golden holds."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    _, arch, mode, fused, _ = p.split('_')
    res, fmap, latent, dlatent, mfmaps = [int(v) for v in G['net_cfg']]
    net = _network('G', arch, cuda_device, 'Gparam_%s.' % arch, latent_size=latent, dlatent_size=dlatent, mapping_fmaps=mfmaps)
    kw = dict(train=dict(is_training=True), val=dict(is_validation=True, truncation_psi_val=0.7, truncation_cutoff_val=4), plain=dict(truncation_psi=0.5, randomize_noise=False))[mode]
    kinds = [str(k) for k in G[p + 'tape_kinds']]
    tape = tfutil.RandomTape([(k, G['%stape_%03d' % (p, i)]) for i, k in enumerate(kinds)])
    z = dev32(G[p + 'z'], cuda_device)
    with tfutil.use_random(tape), torch.no_grad():
        img, dl = net.get_output_for(z, torch.zeros(z.shape[0], 0, device=cuda_device), return_dlatents=True, fused_modconv=bool(int(fused)), **kw)
    assert tape.pos == len(kinds), 'the HIP path drew %d of the %d tensors the reference run drew' % (tape.pos, len(kinds))
    assert rel_err(dl, G[p + 'dlatents']) < 1e-5
    assert rel_err(img, G[p + 'img']) < 1e-4
    assert rel_err(net.vars['dlatent_avg'], G[p + 'dlatent_avg_after']) < 1e-5


@pytest.mark.parametrize('arch', ['skip', 'resnet', 'orig'])
def test_discriminator_against_reference_execution(cuda_device, arch):
    net = _network('D', arch, cuda_device, 'Dparam_%s.' % arch)
    x = dev32(G['D_%s_x' % arch], cuda_device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        s, f = net.get_output_for(x, torch.zeros(x.shape[0], 0, device=cuda_device), is_training=True, return_features=True)
    assert rel_err(s, G['D_%s_scores' % arch]) < 1e-4 and rel_err(f, G['D_%s_features' % arch]) < 1e-4


# ---- the half instantiations of the two custom ops (upfirdn_2d.cu:323-324, fused_bias_act.cu:185-186) -------------------------

@pytest.mark.parametrize('i', range(int(G['upfirdn_cases'])))
def test_upfirdn_half_instantiation(cuda_device, i):
    """T = half: inputs, taps and outputs in binary16, float accumulation (upfirdn_2d.cu:101,114).  Expected = the float64 oracle on
    the half-rounded inputs and taps, rounded to half; 1.5 half ulps of the largest output.  Gradient through autograd too."""
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import upfirdn_2d
    from oracle import upfirdn_2d as OU
    p = 'upfirdn_%d_' % i
    upx, upy, downx, downy, px0, px1, py0, py1 = [int(v) for v in G[p + 'params']]
    kw = dict(upx=upx, upy=upy, downx=downx, downy=downy, padx0=px0, padx1=px1, pady0=py0, pady1=py1)
    xh = torch.from_numpy(G[p + 'x']).to(torch.float16)
    kh = G[p + 'k'].astype(np.float16).astype(np.float64)
    want = OU.upfirdn_2d_ref(xh.double(), kh, **kw)
    x = xh.to(cuda_device).requires_grad_(True)
    y = upfirdn_2d(x, G[p + 'k'], **kw)
    assert y.dtype == torch.float16
    tol = 1.5 * 2.0 ** -11 * float(want.abs().max())
    assert float((y.detach().double().cpu() - want).abs().max()) <= tol
    dyh = torch.from_numpy(G[p + 'dy']).to(torch.float16)
    dx, = torch.autograd.grad(y, x, dyh.to(cuda_device))
    gp = OU.upfirdn_2d_grad_params(xh.shape[1], xh.shape[2], kh, **kw)
    want_dx = OU.upfirdn_2d_ref(dyh.double(), gp['k'], **{n: gp[n] for n in kw})
    assert dx.dtype == torch.float16 and float((dx.double().cpu() - want_dx).abs().max()) <= 1.5 * 2.0 ** -11 * float(want_dx.abs().max())


def test_fused_bias_act_half_instantiation(cuda_device):
    """T = half for every activation and grad = 0 / 1 / 2 through the C ABI wrapper: each element is widened to float, evaluated with
    the float table and rounded on store (fused_bias_act.cu:56-61,113) -- expected = the oracle's restatement of that table in
    float on the half-rounded inputs, rounded to half (one half ulp)."""
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dnnlib.tflib.ops.fused_bias_act import fused_bias_act, activation_funcs
    from oracle import fused_bias_act as OF
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 8, 5, 5, generator=g) * 2).to(torch.float16)
    b = torch.randn(8, generator=g).to(torch.float16)
    ref = torch.rand(3, 8, 5, 5, generator=g).to(torch.float16)
    for name, spec in activation_funcs.items():
        for grad in (0, 1, 2):
            got = hip_ops.fused_bias_act_raw(x.to(cuda_device), b.to(cuda_device), None if grad == 0 else ref.to(cuda_device), grad, spec.hip_idx, 0.2, 1.3, 8, 25)
            want = OF.fused_bias_act_kernel_ref(x.float(), b.float(), None if grad == 0 else ref.float(), grad, spec.hip_idx, 0.2, 1.3, 25).reshape(x.shape)
            assert got.dtype == torch.float16
            err = (got.float().cpu() - want).abs()
            assert bool((err <= 2.0 ** -10 * want.abs() + 1e-4).all()), (name, grad, float(err.max()))
    # the operator surface on half tensors, with gradients (lrelu: the generic Function; piecewise-linear fp32 fast path is not taken)
    xg = x.to(cuda_device).requires_grad_(True)
    bg = b.to(cuda_device).requires_grad_(True)
    y = fused_bias_act(xg, bg, act='lrelu')
    y.float().sum().backward()
    yo = OF.fused_bias_act(x.double(), b.double(), act='lrelu')
    assert y.dtype == torch.float16 and float((y.detach().double().cpu() - yo).abs().max()) <= 2.0 ** -10 * float(yo.abs().max())
    assert xg.grad.dtype == torch.float16 and bg.grad.shape == (8,)
    want_db = torch.where(x.double() + b.double().view(1, 8, 1, 1) > 0, 1.0, 0.2).mul(np.sqrt(2)).sum(dim=(0, 2, 3))
    assert float((bg.grad.double().cpu() - want_db).abs().max()) <= 2e-2 * float(want_db.abs().max())
    with pytest.raises(TypeError):
        hip_ops.fused_bias_act_raw(x.to(cuda_device), b.float().to(cuda_device), None, 0, 3, 0.2, 1.0, 8, 25)      # mixed types


# ---- round 4: the training-side statements (tests/golden/make_ref_train_golden.py -> ref_train_golden.npz) ----------------------
# training/loss.py, training_loop.process_reals, dnnlib/tflib/optimizer.py (Optimizer + SimpleAdam) and
# Network.setup_as_moving_average_of EXECUTED by the reference; here the HIP path (product functions) meets the same vectors.

TG = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_train_golden.npz'), allow_pickle=False)


def _tg_tape(prefix):
    kinds = [str(k) for k in TG[prefix + 'tape_kinds']]
    return [(k, TG['%stape_%03d' % (prefix, i)]) for i, k in enumerate(kinds)]


def _loss_networks(dev):
    from inclusivegan_amd.dnnlib import tflib
    from tests.util import lpips_params_from_seed
    res, fmap, B, latent, dlatent, mfmaps, seed = [int(v) for v in TG['loss_cfg']]
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=fmap, device=dev, seed=1)
    Gn = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', latent_size=latent, dlatent_size=dlatent, mapping_fmaps=mfmaps, **kw)
    Dn = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', **kw)
    Ln = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=res, device=dev, seed=1)
    lp = lpips_params_from_seed(seed)
    with torch.no_grad():
        for net, src in ((Gn, lambda n: TG['loss_Gparam.' + n.replace('/', '.')]), (Dn, lambda n: TG['loss_Dparam.' + n.replace('/', '.')]), (Ln, lambda n: lp[n])):
            for name, v in net.vars.items():
                v.copy_(dev32(src(name), dev).reshape(v.shape))
    return Gn, Dn, Ln, res, B


@pytest.mark.parametrize('w', [2.5, 0.0], ids=['weight_2.5', 'weight_0'])
def test_generator_loss_against_reference_execution(cuda_device, w):
    """G_logistic_ns_rec_interp_arb_pathreg on the HIP path vs the values the reference's own loss.py produced on the same weights, inputs and
    draws: main term (rec + interp LPIPS + adversarial) and the path-length regulariser with its pl_mean update."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    from tests.util import gloss_tape_in_product_order
    Gn, Dn, Ln, res, B = _loss_networks(cuda_device)
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=cuda_device)
    cl = lambda a: dev32(a, cuda_device).contiguous(memory_format=torch.channels_last)
    lab = torch.zeros(B, 0, device=cuda_device)
    p = 'Gloss_w%d_' % int(w * 10)
    args = (Gn, Dn, Ln, ts, B, cl(TG['loss_reals_rec_1']), lab, dev32(TG['loss_latents_rec_1'], cuda_device), cl(TG['loss_reals_rec_2']), lab, dev32(TG['loss_latents_rec_2'], cuda_device))
    tape = tfutil.RandomTape(gloss_tape_in_product_order(_tg_tape(p + 'main_'), B, w))
    with tfutil.use_random(tape), torch.no_grad():
        loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(*args, NN_rec_lpips_weight=w, phase='loss')
    assert reg is None and tape.pos == len(tape.entries)
    assert rel_err(loss, TG[p + 'loss']) < 2e-4
    Gn.pl_mean_var = torch.tensor(float(TG[p + 'pl_mean_before']), device=cuda_device)
    tape = tfutil.RandomTape(_tg_tape(p + 'pl_'))
    with tfutil.use_random(tape):
        loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(*args, NN_rec_lpips_weight=w, phase='reg')
    assert loss is None and tape.pos == len(tape.entries)
    assert rel_err(reg.detach(), TG[p + 'reg']) < 1e-3
    assert abs(float(Gn.pl_mean_var) - float(TG[p + 'pl_mean_after'])) < 1e-5 * float(TG[p + 'pl_mean_after'])


def test_discriminator_loss_against_reference_execution(cuda_device):
    """D_logistic_r1 on the HIP path (the one-pass interleaved form for the main term, the second-order path for R1) vs the reference's
    own loss.py."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    Gn, Dn, _, res, B = _loss_networks(cuda_device)
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=cuda_device)
    reals = dev32(TG['Dloss_reals'], cuda_device).contiguous(memory_format=torch.channels_last)
    lab = torch.zeros(2 * B, 0, device=cuda_device)
    for phase, key, tol in (('loss', 'Dloss_loss', 2e-4), ('reg', 'Dloss_reg', 1e-3), ('both', None, None)):
        tape = tfutil.RandomTape(_tg_tape('Dloss_') if phase != 'reg' else [])
        with tfutil.use_random(tape):
            loss, reg = PL.D_logistic_r1(Gn, Dn, ts, B, reals, lab, gamma=float(TG['Dloss_gamma']), phase=phase)
        assert tape.pos == len(tape.entries)
        if phase == 'both':
            assert rel_err(loss.detach(), TG['Dloss_loss']) < 2e-4 and rel_err(reg.detach(), TG['Dloss_reg']) < 1e-3
        else:
            assert rel_err((loss if phase == 'loss' else reg).detach(), TG[key]) < tol, phase


def test_process_reals_against_reference_execution(cuda_device):
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training.training_loop import process_reals
    for p in [str(c) for c in TG['preals_cases']]:
        lod, mirror, d0, d1 = TG[p + 'cfg']
        x = torch.from_numpy(TG[p + 'x']).to(cuda_device)
        with tfutil.use_random(tfutil.RandomTape([('uniform', TG[p + 'coin'])] if mirror else [])):
            y, _ = process_reals(x, None, lod, bool(mirror), [d0, d1], [-1, 1])
        assert y.dtype == torch.float32 and rel_err(y, TG[p + 'y']) < 1e-6, p


def _toy_build(latents_in, **_kw):
    """Four trainables with the shapes of the optimizer goldens (one of them a scalar: the bucket pads every slot to 16 bytes)."""
    from inclusivegan_amd.dnnlib.tflib.tfutil import get_variable
    for i, s in enumerate([(3, 3, 4, 5), (5,), (7, 2), ()]):
        get_variable('w%d' % i, shape=list(s), initializer=('zeros',))
    return latents_in


@pytest.mark.parametrize('p', [str(c) for c in TG['opt_cases']])
def test_optimizer_against_reference_execution(cuda_device, p):
    """tflib.Optimizer.apply_updates (finite gate + flat Adam kernel, slots and beta powers shared between a main and a `share=`
    optimizer that take turns) vs the weights the reference's Optimizer + SimpleAdam produced step by step.  The reference's devices are
    summed here the way GradientExchange / allreduce_mean_ do it: each gradient times 1 / devices, then added (optimizer.py:186,199)."""
    from inclusivegan_amd.dnnlib import tflib
    lr, b1, b2, eps, devices, _mult, steps = TG[p + 'hp']
    net = tflib.Network('T', func_name=_toy_build, device=cuda_device, latent_size=4)
    names = list(net.trainables)
    with torch.no_grad():
        for i, n in enumerate(names):
            net.vars[n].copy_(dev32(TG['%sw0_%d' % (p, i)], cuda_device).reshape(net.vars[n].shape))
    main = tflib.Optimizer(name='Train', learning_rate=lr, beta1=b1, beta2=b2, epsilon=eps)
    reg = tflib.Optimizer(name='Reg', share=main, learning_rate=lr, beta1=b1, beta2=b2, epsilon=eps)
    for s in range(int(steps)):
        opt = (main, reg)[s % 2]
        opt.mark_registered(net)
        net.flat_grads.zero_()
        with torch.no_grad():
            for i, n in enumerate(names):
                for d in range(int(devices)):
                    net.trainables[n].grad.add_(dev32(TG['%sgrad_s%d_d%d_%d' % (p, s, d, i)], cuda_device).reshape(net.trainables[n].shape) * np.float32(1.0 / devices))
        opt.apply_updates()
        for i, n in enumerate(names):
            want = TG['%sw_s%d_%d' % (p, s, i)]
            got = net.vars[n].detach().double().cpu().numpy().reshape(want.shape)
            assert np.abs(got - want).max() <= 3e-6 * max(np.abs(want).max(), lr), (p, s, n)      # fp32 Adam: an update is ~lr, whatever the weight's size
    assert main.overflow_count() == (1 if p == 'opt_two_devices_' else 0)


def test_moving_average_against_reference_execution(cuda_device):
    """Network.setup_as_moving_average_of (flat EMA kernel for the trainables, copy / lerp for the rest) vs network.py:341-351 executed."""
    from inclusivegan_amd import hip_ops
    trainable = set(str(n) for n in TG['ema_trainable'])
    names = [k[len('ema_src.'):] for k in TG.files if k.startswith('ema_src.')]
    for p in [str(c) for c in TG['ema_cases']]:
        beta, beta_nt = TG[p + 'betas']
        tn = [n for n in names if n.replace('.', '/') in trainable]
        dst = torch.cat([dev32(TG['ema_dst.' + n], cuda_device).reshape(-1) for n in tn])
        src = torch.cat([dev32(TG['ema_src.' + n], cuda_device).reshape(-1) for n in tn])
        hip_ops.ema_raw(dst, src, float(beta))
        want = np.concatenate([TG[p + 'after.' + n].reshape(-1) for n in tn])
        assert np.abs(dst.double().cpu().numpy() - want).max() <= 1e-6 * np.abs(want).max(), p
    # through the Network method, non-trainables included (beta_nontrainable = 0: copied)
    from inclusivegan_amd.dnnlib import tflib
    res, fmap, _B, latent, dlatent, mfmaps, _seed = [int(v) for v in TG['loss_cfg']]
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=fmap, device=cuda_device, architecture='skip', latent_size=latent, dlatent_size=dlatent, mapping_fmaps=mfmaps)
    Gn = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', seed=1, **kw)
    Gs = tflib.Network('Gs', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', seed=2, **kw)
    with torch.no_grad():
        Gn.vars['dlatent_avg'].normal_()
    before = {n: v.detach().double().cpu().numpy().copy() for n, v in Gs.vars.items()}
    beta = float(TG['ema_1_betas'][0])
    Gs.setup_as_moving_average_of(Gn, beta=beta)()
    for n, v in Gs.vars.items():
        src = Gn.vars[n].detach().double().cpu().numpy()
        want = src + (before[n] - src) * (beta if n in Gs.trainables else 0.0)          # lerp(src, dst, beta), network.py:348
        assert np.abs(v.detach().double().cpu().numpy() - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-6), n
