"""GPU parity of the full G / D / loss / optimizer path against the CPU oracle, on
Stacked-MNIST-shaped inputs (32x32) with reduced channel widths so the fp64 oracle runs in seconds.
Random draws made by the HIP path are recorded and replayed into the oracle (RNG parity with TF is
impossible, SURVEY.md section 7, so every random tensor is injected).

Tolerances: network outputs / loss scalars 1e-4 .. 2e-4 relative (fp32 HIP vs fp64 oracle through ~20
layers); regulariser values (built from fp32 gradients) 1e-3; parameter gradients 5e-3 in relative
L2 norm per tensor."""
import numpy as np
import pytest
import torch

from tests.util import rel_err, gloss_tape_in_reference_order as _gloss_tape_in_reference_order

pytestmark = pytest.mark.gpu

RES = 32
FMAP = 1024      # nf: 512,256,128,64,32 at 4..32  (config-e uses 8192)


def _nets(dev, label_size=0, fmap=FMAP, res=RES):
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=res, label_size=label_size, fmap_base=fmap, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=12, **kw)
    rng = np.random.RandomState(0)
    with torch.no_grad():   # biases / noise strengths are zero-initialised; make them matter
        for net in (G, D):
            for n, v in net.vars.items():
                if n.endswith('bias') or n.endswith('noise_strength'):
                    v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(dev).reshape(v.shape))
    return G, D


def _oracle_params(net, dtype=torch.float64):
    p = {n: v.detach().to(dtype).cpu() for n, v in net.vars.items()}
    for n in net.trainables:
        p[n].requires_grad_(True)
    return p


def test_generator_forward_matches_oracle(cuda_device):
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from oracle import networks_stylegan2 as ON
    from oracle.misc import Tape
    G, _ = _nets(cuda_device)
    z = torch.randn(4, 512, device=cuda_device)
    lab = torch.zeros(4, 0, device=cuda_device)
    gp = _oracle_params(G)
    for training in (True, False):
        rec = tfutil.RecordingRandom()
        with tfutil.use_random(rec), torch.no_grad():
            img = G.get_output_for(z, lab, is_training=training, is_validation=not training)
        for fused in (True, False):
            img_o = ON.G_main(gp, z.double().cpu(), Tape(rec.entries, torch.float64), RES, fmap_base=FMAP, architecture='skip',
                              is_training=training, is_validation=not training, fused_modconv=fused)
            assert rel_err(img, img_o) < 1e-4, (training, fused)


def test_discriminator_forward_and_features(cuda_device):
    from oracle import networks_stylegan2 as ON
    _, D = _nets(cuda_device)
    img = torch.randn(12, 3, RES, RES, device=cuda_device)
    lab = torch.zeros(12, 0, device=cuda_device)
    with torch.no_grad():
        s, f = D.get_output_for(img, lab, is_training=True, return_features=True)
        s2, f2 = D.get_output_for(img, lab, is_training=True)
    assert f2 is None and torch.equal(s, s2)
    so, fo = ON.D_stylegan2_feature(_oracle_params(D), img.double().cpu(), RES, fmap_base=FMAP, architecture='resnet')
    assert rel_err(s, so) < 1e-4
    assert tuple(f.shape) == tuple(fo.shape)
    assert rel_err(f, fo) < 1e-4


def _grad_errs(net, oparams):
    """Relative L2 error per trainable.  (A max-abs metric is not robust here: lrelu's derivative is
    discontinuous, and an fp32 pre-activation within rounding of 0 can land on the other side of the
    kink than its fp64 oracle twin, which changes a single summand of a bias gradient by O(1).)"""
    errs = {}
    sc_h, sc_o = [], []
    for n, v in net.trainables.items():
        go = oparams[n].grad
        if go is None:
            continue
        gh = v.grad.detach().double().cpu()
        if go.numel() == 1:      # scalar parameters (noise_strength): single, heavily cancelling sums --
            sc_h.append(gh.reshape(1)); sc_o.append(go.reshape(1))   # judged jointly as one vector
            continue
        errs[n] = float((gh - go).norm() / (go.norm() + 1e-30))
    if sc_o:
        errs['<all scalar parameters>'] = float((torch.cat(sc_h) - torch.cat(sc_o)).norm() / (torch.cat(sc_o).norm() + 1e-30))
    return errs


def _check_losses_and_gradients(dev, res, fmap, B, lpips_weight, seed, only_g_loss=False):
    """G loss (rec + interp LPIPS + adversarial), G path-length reg, D loss, D R1 reg through the HIP path: values and the gradients
    w.r.t. every trainable, including the second-order paths, against the fp64 oracle on the same weights, inputs and draws."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    from oracle import loss as OL
    from oracle.misc import Tape
    G, D = _nets(dev, fmap=fmap, res=res)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=res, device=dev, seed=13)
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=dev)
    g = torch.Generator().manual_seed(seed)
    reals1 = (torch.rand(B, 3, res, res, generator=g) * 2 - 1); reals2 = (torch.rand(B, 3, res, res, generator=g) * 2 - 1)
    z1 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1); z2 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1)
    lab = torch.zeros(B, 0, device=dev)
    cfg = dict(resolution=res, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
    lpo = {n: v.detach().double().cpu() for n, v in lp.vars.items()}
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    worst = {}

    for phase in (('loss',) if only_g_loss else ('loss', 'reg')):
        G.zero_grad(); D.zero_grad()
        D.requires_grad_(False)
        rec = tfutil.RecordingRandom()
        with tfutil.use_random(rec):
            loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, cl(reals1), lab, z1.to(dev), cl(reals2), lab, z2.to(dev),
                                                                NN_rec_lpips_weight=lpips_weight, phase=phase)
        val = loss if phase == 'loss' else reg
        torch.autograd.backward(val.mean(), inputs=list(G.trainables.values()))
        D.requires_grad_(True)
        gp = _oracle_params(G); dp = _oracle_params(D)
        gp['dlatent_avg'] = torch.zeros_like(gp['dlatent_avg']) if phase == 'loss' else gp['dlatent_avg']
        entries = _gloss_tape_in_reference_order(rec.entries, B) if (phase == 'loss' and lpips_weight != 0) else rec.entries
        lo, ro, _ = OL.G_loss(gp, dp, lpo, cfg, Tape(entries, torch.float64), B, reals1.double(), z1.double(), reals2.double(), z2.double(),
                              lpips_weight, phase=phase, state={})
        vo = lo if phase == 'loss' else ro
        vo.mean().backward()
        worst['G_' + phase + '_value'] = rel_err(val, vo)
        assert rel_err(val, vo) < (2e-4 if phase == 'loss' else 1e-3), phase
        errs = _grad_errs(G, gp)
        w = max(errs, key=errs.get)
        assert errs[w] < 5e-3, (phase, w, errs[w])
        worst['G_' + phase] = errs[w]
        G.pl_mean_var = torch.zeros((), device=dev)

    for phase in (() if only_g_loss else ('loss', 'reg')):
        G.zero_grad(); D.zero_grad()
        reals = torch.rand(2 * B, 3, res, res, generator=g) * 2 - 1
        lab2 = torch.zeros(2 * B, 0, device=dev)
        rec = tfutil.RecordingRandom()
        gp = _oracle_params(G); dp = _oracle_params(D)
        with tfutil.use_random(rec):
            loss, reg = PL.D_logistic_r1(G, D, ts, B, cl(reals), lab2, gamma=100, phase=phase)
        val = loss if phase == 'loss' else reg
        torch.autograd.backward(val.mean(), inputs=list(D.trainables.values()))
        lo, ro, _ = OL.D_loss(gp, dp, cfg, Tape(rec.entries, torch.float64), B, reals.double(), gamma=100, phase=phase, state={})
        vo = lo if phase == 'loss' else ro
        vo.mean().backward()
        worst['D_' + phase + '_value'] = rel_err(val, vo)
        assert rel_err(val, vo) < (2e-4 if phase == 'loss' else 1e-3), phase
        errs = _grad_errs(D, dp)
        w = max(errs, key=errs.get)
        assert errs[w] < 5e-3, (phase, w, errs[w])
        worst['D_' + phase] = errs[w]
    return worst


def test_losses_and_gradients_match_oracle(cuda_device):
    """All four phases at 32x32, reduced width (the config-e width at 32x32 is covered op by op through the real loop in
    tests/test_gpu_loop_parity.py)."""
    _check_losses_and_gradients(cuda_device, RES, FMAP, 6, 2.5, seed=5)


@pytest.mark.parametrize('lpips_weight', [2.5, 0.0], ids=['config4_imle', 'config3_adversarial_only'])
def test_losses_and_gradients_at_128_config_e(cuda_device, lpips_weight):
    """BASELINE configs 3 and 4 at their own size: config-e-Gskip-Dresnet, 128x128, fmap_base 8192, batch 2 -- the four phases
    (G loss with NN_rec_lpips_weight 2.5 / 0, path-length regulariser at pl batch 1, D loss, R1) with the gradient of every
    trainable against the fp64 oracle (relative L2 per variable, 5e-3)."""
    worst = _check_losses_and_gradients(cuda_device, 128, 8192, 2, lpips_weight, seed=9, only_g_loss=(lpips_weight == 0))      # the other three phases do not depend on the weight
    print('128x128 config-e, lpips weight %g: worst per-variable gradient error by phase %s' % (lpips_weight, worst))


def test_training_ops_are_bit_reproducible(cuda_device):
    """The device work of the G and D steps (loss + backward into the flat bucket), run twice from the same RNG state
    on the same inputs, gives bit-identical gradients: every kernel on the path reduces in a fixed order."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    dev = cuda_device
    G, D = _nets(dev)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=RES, device=dev, seed=13)
    ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
    B = 6
    g = torch.Generator().manual_seed(7)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    r1 = cl(torch.rand(B, 3, RES, RES, generator=g) * 2 - 1); r2 = cl(torch.rand(B, 3, RES, RES, generator=g) * 2 - 1)
    z1 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1).to(dev)
    z2 = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=1).to(dev)
    reals = cl(torch.rand(2 * B, 3, RES, RES, generator=g) * 2 - 1)
    lab = torch.zeros(B, 0, device=dev); lab2 = torch.zeros(2 * B, 0, device=dev)

    def g_step(phase):
        G.zero_grad(); D.requires_grad_(False)
        G.pl_mean_var = torch.zeros((), device=dev)
        loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, r1, lab, z1, r2, lab, z2, NN_rec_lpips_weight=2.5, phase=phase)
        torch.autograd.backward((loss if phase == 'loss' else reg).mean(), inputs=list(G.trainables.values()))
        D.requires_grad_(True)
        return G.flat_grads.clone()

    def d_step(phase):
        D.zero_grad(); G.requires_grad_(False)
        loss, reg = PL.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase=phase)
        torch.autograd.backward((loss if phase == 'loss' else reg).mean(), inputs=list(D.trainables.values()))
        G.requires_grad_(True)
        return D.flat_grads.clone()

    avg0 = G.vars['dlatent_avg'].detach().clone()
    for step in (g_step, d_step):
        for phase in ('loss', 'reg'):
            outs = []
            for _ in range(2):
                torch.manual_seed(99)
                with torch.no_grad():
                    G.vars['dlatent_avg'].copy_(avg0)       # the G pass moves it (networks_stylegan2.py:203-209)
                outs.append(step(phase))
            assert torch.equal(outs[0], outs[1]), (step.__name__, phase)
            assert float(outs[0].abs().max()) > 0


def test_optimizer_step_and_ema_match_oracle(cuda_device):
    """Optimizer.register_gradients/apply_updates (flat bucket, finite check, Adam with lazy-reg
    beta scaling, shared slots) and Gs EMA against the NumPy SimpleAdam restatement."""
    from inclusivegan_amd.dnnlib import tflib
    from oracle import optimizer as OO
    dev = cuda_device
    G, _ = _nets(dev)
    Gs = G.clone('Gs')
    ema = Gs.setup_as_moving_average_of(G, beta=0.5 ** (12 / 10000.0))
    c = 4 / 5
    opt = tflib.Optimizer(name='TrainG', learning_rate=lambda: 0.002 * c, beta1=0.0 ** c, beta2=0.99 ** c, epsilon=1e-8)
    reg_opt = tflib.Optimizer(name='RegG', share=opt, learning_rate=lambda: 0.002 * c, beta1=0.0 ** c, beta2=0.99 ** c, epsilon=1e-8)
    w0 = G.flat_params.detach().cpu().numpy().copy()
    adam = OO.SimpleAdam(w0.size, 0.002 * c, 0.0 ** c, 0.99 ** c, 1e-8)
    wo = w0.copy(); gs_o = w0.copy()
    for step, o in enumerate([opt, reg_opt, opt]):
        loss = sum((p * p).sum() * (0.5 + 0.1 * step) for p in G.trainables.values())
        o.register_gradients(loss, G)
        g = G.flat_grads.detach().cpu().numpy().copy()
        o.apply_updates()
        ema()
        adam.apply(wo, g)
        gs_o = OO.ema(gs_o, wo, 0.5 ** (12 / 10000.0))
    assert rel_err(G.flat_params, wo) < 2e-6
    assert rel_err(Gs.flat_params, gs_o) < 2e-6


def test_training_loop_runs_and_learns_shapes(cuda_device):
    """A few iterations of the real loop on tiny synthetic data: the IMLE refresh, all four step kinds
    (G, G-reg, D, D-reg) execute, parameters move, nothing becomes non-finite."""
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    seen = []
    def on_it(info):
        seen.append(info['cur_nimg'])
        return len(seen) >= 3
    refresh = []
    out = TL.training_loop(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=512, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=512, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=32, num_channels=3, label_size=10, label_kind='onehot'),
        sched_args=EasyDict(minibatch_gpu_base=6, minibatch_size_base=6), tf_config={'rnd.np_random_seed': 1000},
        total_kimg=1, data_size=48, num_samples_factor=4, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=64,
        hooks=dict(on_iteration=on_it, on_refresh=refresh.append))
    assert seen == [12, 24, 36] and len(refresh) == 1
    for net in (out['G'], out['D'], out['Gs']):
        assert bool(torch.isfinite(net.flat_params).all())


def test_g_loss_adversarial_only_config3(cuda_device):
    """BASELINE config 3: NN_rec_lpips_weight = 0 (training/loss.py with the reconstruction / interpolation terms weighted to
    zero; tests/test_oracle_networks.py shows they then contribute exactly nothing).  The HIP loss takes its adversarial-only
    branch: value and the gradient of every G trainable against the oracle."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    from oracle import loss as OL
    from oracle.misc import Tape
    dev = cuda_device
    G, D = _nets(dev)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=RES, device=dev, seed=13)
    ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
    B = 6
    g = torch.Generator().manual_seed(8)
    reals1 = torch.rand(B, 3, RES, RES, generator=g) * 2 - 1; reals2 = torch.rand(B, 3, RES, RES, generator=g) * 2 - 1
    z1 = torch.randn(B, 512, generator=g); z2 = torch.randn(B, 512, generator=g)
    lab = torch.zeros(B, 0, device=dev)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    cfg = dict(resolution=RES, num_channels=3, fmap_base=FMAP, G_arch='skip', D_arch='resnet')
    G.zero_grad(); D.requires_grad_(False)
    gp = _oracle_params(G); dp = _oracle_params(D)
    rec = tfutil.RecordingRandom()
    with tfutil.use_random(rec):
        loss, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, cl(reals1), lab, z1.to(dev), cl(reals2), lab, z2.to(dev),
                                                            NN_rec_lpips_weight=0.0, phase='loss')
    assert reg is None and tuple(loss.shape) == (B,)
    torch.autograd.backward(loss.mean(), inputs=list(G.trainables.values()))
    D.requires_grad_(True)
    lo, _, terms = OL.G_loss(gp, dp, {}, cfg, Tape(rec.entries, torch.float64), B, reals1.double(), z1.double(), reals2.double(), z2.double(),
                             0.0, phase='loss', state={})
    assert set(terms) == {'loss_G_arb'}
    lo.mean().backward()
    assert rel_err(loss, lo) < 2e-4
    errs = _grad_errs(G, gp)
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5e-3, (worst, errs[worst])


def test_networks_at_config_e_width(cuda_device):
    """BASELINE config 2 / config-e proper: fmap_base 8192 (512 channels up to 32x32) at 32x32, batch 2 -- G (training and
    validation mode) and D (scores and features) against the fp64 oracle at FULL width."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from oracle import networks_stylegan2 as ON
    from oracle.misc import Tape
    dev = cuda_device
    kw = dict(num_channels=3, resolution=32, label_size=0, fmap_base=8192, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=21, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=22, **kw)
    assert G.vars['G_synthesis/32x32/Conv1/weight'].shape == (3, 3, 512, 512)
    rng = np.random.RandomState(1)
    with torch.no_grad():
        for net in (G, D):
            for n, v in net.vars.items():
                if n.endswith('bias') or n.endswith('noise_strength'):
                    v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(dev).reshape(v.shape))
    z = torch.randn(2, 512, device=dev)
    lab = torch.zeros(2, 0, device=dev)
    gp = _oracle_params(G); dp = _oracle_params(D)
    for training in (True, False):
        rec = tfutil.RecordingRandom()
        with tfutil.use_random(rec), torch.no_grad():
            img = G.get_output_for(z, lab, is_training=training, is_validation=not training)
        with torch.no_grad():
            img_o = ON.G_main(gp, z.double().cpu(), Tape(rec.entries, torch.float64), 32, fmap_base=8192, architecture='skip',
                              is_training=training, is_validation=not training, fused_modconv=False)
        assert rel_err(img, img_o) < 1e-4, training
    x = torch.randn(6, 3, 32, 32, device=dev)
    with torch.no_grad():
        s, f = D.get_output_for(x, torch.zeros(6, 0, device=dev), is_training=True, return_features=True)
        so, fo = ON.D_stylegan2_feature(dp, x.double().cpu(), 32, fmap_base=8192, architecture='resnet')
    assert rel_err(s, so) < 1e-4 and rel_err(f, fo) < 1e-4


def test_generator_and_discriminator_at_128_full_config(cuda_device):
    """The bench configuration itself: config-e-Gskip-Dresnet at 128x128, fmap_base 8192 -- one G forward (batch 1, validation
    mode: deterministic up to the recorded noise) and one D forward (batch 2) against the fp64 oracle."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from oracle import networks_stylegan2 as ON
    from oracle.misc import Tape
    dev = cuda_device
    kw = dict(num_channels=3, resolution=128, label_size=0, fmap_base=8192, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=31, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=32, **kw)
    rng = np.random.RandomState(2)
    with torch.no_grad():
        for net in (G, D):
            for n, v in net.vars.items():
                if n.endswith('bias') or n.endswith('noise_strength'):
                    v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(dev).reshape(v.shape))
    z = torch.randn(1, 512, device=dev)
    rec = tfutil.RecordingRandom()
    with tfutil.use_random(rec), torch.no_grad():
        img = G.get_output_for(z, torch.zeros(1, 0, device=dev), is_validation=True)
    assert tuple(img.shape) == (1, 3, 128, 128)
    with torch.no_grad():
        img_o = ON.G_main(_oracle_params(G), z.double().cpu(), Tape(rec.entries, torch.float64), 128, fmap_base=8192, architecture='skip',
                          is_validation=True, fused_modconv=False)
    assert rel_err(img, img_o) < 1e-4
    x = torch.randn(2, 3, 128, 128, device=dev)
    with torch.no_grad():
        s, _ = D.get_output_for(x, torch.zeros(2, 0, device=dev), is_training=True)
        so, _ = ON.D_stylegan2_feature(_oracle_params(D), x.double().cpu(), 128, fmap_base=8192, architecture='resnet')
    assert rel_err(s, so) < 1e-4


@pytest.mark.parametrize('variant', ['exclusive+projection', 'mirror+attributes'])
def test_training_loop_optional_paths(cuda_device, variant):
    """The loop's optional branches run end to end on the device: exclusive k-NN assignment with a random projection in front
    (training_loop.py:28-35,205-213,382-396), mirror augmentation (:47-49) with the attribute AND-mask selection (:416-424) on
    CelebA-shaped labels."""
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    from inclusivegan_amd.training import imle
    seen, batches = [], []
    def on_it(info):
        seen.append(info['cur_nimg'])
        return len(seen) >= 2
    extra = dict(exclusive_retrieved_code=True, init_proj_dim=24) if variant.startswith('exclusive') else dict(
        mirror_augment=True, attr_interesting='Smiling', attr_names=list(imle.CELEBA_ATTRIBUTES))
    label = dict(label_size=10, label_kind='onehot') if variant.startswith('exclusive') else dict(label_size=40, label_kind='attributes')
    out = TL.training_loop(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=512, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=512, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=32, num_channels=3, **label),
        sched_args=EasyDict(minibatch_gpu_base=3, minibatch_size_base=3), tf_config={'rnd.np_random_seed': 1000},
        total_kimg=1, data_size=48, num_samples_factor=4, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=64,
        hooks=dict(on_iteration=on_it, on_batch=batches.append), **extra)
    assert seen == [6, 12] and len(batches) == 2
    for net in (out['G'], out['D'], out['Gs']):
        assert bool(torch.isfinite(net.flat_params).all())
    if variant.startswith('mirror'):
        col = imle.CELEBA_ATTRIBUTES.index('Smiling')
        for b in batches:
            assert b['labels_rec_1'].shape == (3, 40) and bool((b['labels_rec_1'][:, col] == 1).all()) and bool((b['labels_rec_2'][:, col] == 1).all())


def test_training_loop_config2_full_width(cuda_device):
    """BASELINE config 2 as the loop runs it: Stacked-MNIST-shaped 32x32 data with 1000-d one-hot labels, config-e width
    (fmap_base 8192), minibatch_gpu 6, IMLE refresh + all four step kinds through captured graphs."""
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    seen = []
    def on_it(info):
        seen.append(info['cur_nimg'])
        return len(seen) >= 5
    refresh = []
    out = TL.training_loop(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=8192, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=8192, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=32, num_channels=3, label_size=1000, label_kind='onehot'),
        sched_args=EasyDict(minibatch_gpu_base=6, minibatch_size_base=6), tf_config={'rnd.np_random_seed': 1000},
        total_kimg=1, data_size=96, num_samples_factor=10, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=128,
        hooks=dict(on_iteration=on_it, on_refresh=refresh.append))
    assert seen == [12, 24, 36, 48, 60] and len(refresh) == 1
    assert out['G'].flat_params.numel() > 20_000_000
    for net in (out['G'], out['D'], out['Gs']):
        assert bool(torch.isfinite(net.flat_params).all())
