"""oracle/ (and the product's host helpers) against tests/golden/ref_train_golden.npz: golden vectors produced by EXECUTING the
reference's own training/loss.py, training_loop.process_reals, dnnlib/tflib/optimizer.py (Optimizer + SimpleAdam),
Network.setup_as_moving_average_of and the optimizer set-up / registration statements of training_loop.py
(tests/golden/make_ref_train_golden.py).  CPU only; the HIP path meets the same vectors in tests/test_gpu_ref_golden.py."""
import os

import numpy as np
import torch

from oracle import loss as OL
from oracle import optimizer as OO
from oracle import train_ops as OTO
from oracle import training_loop as OT
from oracle.misc import Tape
from tests.util import lpips_params_from_seed

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_train_golden.npz'))


def T64(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300)) if a.size else 0.0


def params(prefix, grad=False):
    return {k[len(prefix):].replace('.', '/'): T64(G[k]).requires_grad_(grad) for k in G.files if k.startswith(prefix)}


def tape(prefix):
    kinds = [str(k) for k in G[prefix + 'tape_kinds']]
    return [(k, G['%stape_%03d' % (prefix, i)]) for i, k in enumerate(kinds)]


def loss_cfg():
    res, fmap, B, latent, dlatent, mfmaps, seed = [int(v) for v in G['loss_cfg']]
    cfg = dict(resolution=res, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet', latent_size=latent,
               G_kwargs=dict(dlatent_size=dlatent, mapping_fmaps=mfmaps))
    return cfg, B, seed


def test_generator_loss_and_path_length_terms():
    """loss.py:19-91 executed by the reference: the two reconstruction distances * 0.5 * w, the interpolation term (lerp from the
    distance to real_2 towards the distance to real_1 by the interpolation factor) * 0.4 * w, softplus(-D(G(z))), and the path-length
    statistics around tf.gradients: lengths, pl_mean update with decay 0.01, (length - mean)^2 * 2."""
    cfg, B, seed = loss_cfg()
    gp, dp = params('loss_Gparam.', grad=True), params('loss_Dparam.')
    lp = {k: T64(v) for k, v in lpips_params_from_seed(seed).items()}
    assert [int(v) for v in G['Gloss_w25_lpips_real_sets']] == [1, 2, 2, 1]            # :31 (rec_1|real_1, rec_2|real_2), :41 (interp|real_2, interp|real_1)
    for w in (2.5, 0.0):
        p = 'Gloss_w%d_' % int(w * 10)
        state = dict(pl_mean=T64(G[p + 'pl_mean_before']).reshape(()))
        rand = Tape(tape(p + 'main_') + tape(p + 'pl_'), torch.float64)
        loss, reg, terms = OL.G_loss(gp, dp, lp, cfg, rand, B, T64(G['loss_reals_rec_1']), T64(G['loss_latents_rec_1']), T64(G['loss_reals_rec_2']),
                                     T64(G['loss_latents_rec_2']), w, state=state, literal_zero_weight=True)
        assert rand.pos == len(rand.entries)
        for name in ('loss_NN_rec_lpips', 'loss_NN_interp_lpips', 'loss_G_arb', 'pl_penalty'):
            assert rel(terms[name].detach(), G[p + 'term_' + name]) < 1e-9 or (w == 0 and name != 'loss_G_arb' and np.all(G[p + 'term_' + name] == 0)), (w, name)
        assert rel(loss.detach(), G[p + 'loss']) < 1e-9 and rel(reg.detach(), G[p + 'reg']) < 1e-9
        assert rel(np.asarray(state['pl_mean']).reshape(()), G[p + 'pl_mean_after']) < 1e-12
    # weight 0: the skipped form (what the HIP loss does) gives the reference's value too
    from tests.util import gloss_tape_in_product_order
    p = 'Gloss_w0_'
    loss, _, _ = OL.G_loss(gp, dp, lp, cfg, Tape(gloss_tape_in_product_order(tape(p + 'main_'), B, 0.0), torch.float64), B, T64(G['loss_reals_rec_1']),
                           T64(G['loss_latents_rec_1']), T64(G['loss_reals_rec_2']), T64(G['loss_latents_rec_2']), 0.0, phase='loss', state={})
    assert rel(loss.detach(), G[p + 'loss']) < 1e-9


def test_discriminator_loss_and_r1():
    """loss.py:93-113: softplus(D(G(z))) + softplus(-D(reals)) over 2 * minibatch samples; R1 = sum(grad^2) * gamma / 2."""
    cfg, B, _ = loss_cfg()
    rand = Tape(tape('Dloss_'), torch.float64)
    loss, reg, terms = OL.D_loss(params('loss_Gparam.'), params('loss_Dparam.'), cfg, rand, B, T64(G['Dloss_reals']), gamma=float(G['Dloss_gamma']), state={})
    assert rand.pos == len(rand.entries)
    assert rel(loss.detach(), G['Dloss_loss']) < 1e-9 and rel(reg.detach(), G['Dloss_reg']) < 1e-9
    assert rel(terms['loss_D'].detach(), G['Dloss_term_loss_D']) < 1e-9 and rel(terms['gradient_penalty_D'].detach(), G['Dloss_term_gradient_penalty_D']) < 1e-9


def test_process_reals():
    """training_loop.py:40-60 incl. the mirror coin (< 0.5 keeps), the fade and the upscale; oracle and the product's torch form."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import training_loop as PT
    for p in [str(c) for c in G['preals_cases']]:
        lod, mirror, d0, d1 = G[p + 'cfg']
        x = G[p + 'x']
        y, _ = OT.process_reals(x, None, lod, bool(mirror), [d0, d1], [-1, 1], coin=G[p + 'coin'])
        assert y.shape == G[p + 'y'].shape and rel(y, G[p + 'y']) < 1e-6, p
        with tfutil.use_random(tfutil.RandomTape([('uniform', G[p + 'coin'])] if mirror else [])):
            yp, _ = PT.process_reals(torch.from_numpy(x), None, lod, bool(mirror), [d0, d1], [-1, 1])
        assert rel(yp.numpy(), G[p + 'y']) < 1e-6, p


def test_optimizer_scaling_allsum_gate_and_adam():
    """optimizer.py:169-239 + SimpleAdam :303-336 executed by the reference on one / two devices (incl. a step with a non-finite
    gradient on one device: nobody updates; and the accumulation branch with multiplier 1) against oracle/optimizer.py driven the
    way oracle/train_ops.py drives it (scale by 1 / devices, sum, gate, Adam in float32)."""
    nv = int(G['opt_num_vars'])
    for p in [str(c) for c in G['opt_cases']]:
        lr, b1, b2, eps, devices, _mult, steps = G[p + 'hp']
        w = [G['%sw0_%d' % (p, i)].astype(np.float32) for i in range(nv)]
        sizes = [a.size for a in w]
        flat = np.concatenate([a.reshape(-1) for a in w])
        adam = OO.SimpleAdam(flat.size, lr, b1, b2, eps)
        skipped = 0
        for s in range(int(steps)):
            grads = [np.concatenate([G['%sgrad_s%d_d%d_%d' % (p, s, d, i)].reshape(-1) for i in range(nv)]).astype(np.float32) for d in range(int(devices))]
            total = np.zeros_like(flat)
            for g in grads:
                total += g * np.float32(1.0 / devices)                  # TrainOps.average
            before = flat.copy()
            applied = adam.apply(flat, total)
            skipped += not applied
            if not applied:
                assert np.array_equal(flat, before)
            want = np.concatenate([G['%sw_s%d_%d' % (p, s, i)].reshape(-1) for i in range(nv)])
            assert np.abs(flat - want).max() <= 2e-6 * np.abs(want).max(), (p, s)
        assert skipped == (1 if p == 'opt_two_devices_' else 0)        # the case with an inf on device 1 at step 2
        assert sum(sizes) == flat.size
    # TrainOps.average is that scaling
    t = OTO.TrainOps.__new__(OTO.TrainOps)
    t.world = 2
    a, b = np.float32([1, 2, 3]), np.float32([10, 20, 30])
    assert np.array_equal(t.average([a, b]), a * np.float32(0.5) + b * np.float32(0.5))


def test_moving_average():
    """network.py:341-351: lerp(src, dst, beta) for the trainables, beta_nontrainable for the rest, variables missing in src untouched."""
    trainable = set(str(n) for n in G['ema_trainable'])
    names = [k[len('ema_dst.'):] for k in G.files if k.startswith('ema_dst.')]
    for p in [str(c) for c in G['ema_cases']]:
        beta, beta_nt = G[p + 'betas']
        for n in names:
            dst = G['ema_dst.' + n]
            want = G[p + 'after.' + n]
            if 'ema_src.' + n not in G.files:
                assert np.array_equal(want, dst)
                continue
            b = beta if n.replace('.', '/') in trainable else beta_nt
            got = OO.ema(dst.astype(np.float32), G['ema_src.' + n].astype(np.float32), b)
            assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (p, n)


def test_optimizer_setup_and_registered_objectives():
    """training_loop.py:242-255, :222, :283-291 executed from the reference's syntax tree: oracle helpers and the product's."""
    from inclusivegan_amd.training import training_loop as PT
    for p in [str(c) for c in G['setup_cases']]:
        lazy, gi, di, lrate = G[p + 'cfg']
        lazy = bool(lazy)
        for name, interval in (('TrainG', gi), ('RegG', gi), ('TrainD', di), ('RegD', di)):
            lr, b1, b2, eps, shared = G[p + name]
            assert shared == float(name.startswith('Reg'))
            assert np.allclose(OT.lazy_regularization_args(lrate, 0.0, 0.99, interval, lazy), (lr, b1, b2), rtol=1e-15, atol=0)
            ratio, args = PT.lazy_regularization_args(dict(beta1=0.0, beta2=0.99, epsilon=1e-8), interval, lazy)
            assert np.allclose((lrate * ratio, args['beta1'], args['beta2'], args['epsilon']), (lr, b1, b2, eps), rtol=1e-15, atol=0)
        obj = OT.registered_objectives(G[p + 'in_G_loss'], G[p + 'in_G_reg'], G[p + 'in_D_loss'], G[p + 'in_D_reg'], gi, di, lazy)
        for name in ('TrainG', 'RegG', 'TrainD', 'RegD'):
            assert np.allclose(np.asarray(obj[name], np.float64).reshape(-1), G[p + name + '_registered'].reshape(-1), rtol=1e-14, atol=0), (p, name)
    for mb, kimg, beta in G['gs_beta']:
        assert abs(OT.smoothing_beta(mb, kimg) - beta) <= 1e-15 and abs(PT.smoothing_beta(mb, kimg) - beta) <= 1e-15
