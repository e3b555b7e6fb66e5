"""Loss functions on the HIP path: same names, arguments and `(loss, reg)` return arity as the
reference's `training/loss.py` (G_logistic_ns_rec_interp_arb_pathreg :19-91, D_logistic_r1 :93-113).

The reference builds one graph holding both terms and lets TensorFlow prune whichever the
executed op does not need (G_train_op vs G_reg_op, training_loop.py:474-479).  Eager code has no
pruning, so both functions take an extra keyword `phase` in {'both','loss','reg'}: 'both'
evaluates everything like the reference graph; 'loss' / 'reg' return `None` for the other term.

Random draws (interp factors :36, latents :46,59,98, path-length noise :64) go through
tflib.tfutil's injectable source.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..dnnlib import tflib
from ..dnnlib.tflib import tfutil
from ..dnnlib.tflib.autosummary import autosummary
from .. import hip_ops
from ..metrics import lpips as lpips_mod

#----------------------------------------------------------------------------

def _lpips_pair(lpips, a, b):
    return lpips.get_output_for(a, b)

def G_logistic_ns_rec_interp_arb_pathreg(G, D, lpips, training_set, minibatch_size, reals_rec_1, labels_rec_1, latents_rec_1, reals_rec_2, labels_rec_2, latents_rec_2,
    NN_rec_lpips_weight,
    pl_minibatch_shrink=2, pl_decay=0.01, pl_weight=2.0, phase='both'):
    with G.derived_scope(), D.derived_scope():     # weights are constant inside one loss evaluation
        return _G_loss_impl(G, D, lpips, training_set, minibatch_size, reals_rec_1, labels_rec_1, latents_rec_1, reals_rec_2, labels_rec_2,
                            latents_rec_2, NN_rec_lpips_weight, pl_minibatch_shrink, pl_decay, pl_weight, phase)

def _G_loss_impl(G, D, lpips, training_set, minibatch_size, reals_rec_1, labels_rec_1, latents_rec_1, reals_rec_2, labels_rec_2, latents_rec_2,
    NN_rec_lpips_weight, pl_minibatch_shrink, pl_decay, pl_weight, phase):
    assert phase in ('both', 'loss', 'reg')
    dev = latents_rec_1.device
    latent_shape = G.input_shapes[0][1:]
    loss = None
    reg = None

    if phase in ('both', 'loss'):
        # The reference evaluates G four times on the same weights (rec_1 :25, rec_2 :26, interp :39,
        # arb :48).  Here the four latent batches go through ONE generator pass (G_main num_calls=4 keeps
        # the per-call semantics: own style-mixing cutoff, own dlatent_avg update) and ONE VGG pass per
        # gradient role -- bigger GEMMs, a quarter of the launches, identical values.
        latents_random_shape = [minibatch_size] + latent_shape
        if NN_rec_lpips_weight != 0:
            interp_factors = tfutil.random_uniform([minibatch_size, 1], dev, 0.0, 1.0)          # :36
            latents_random = tfutil.random_normal(latents_random_shape, dev)                     # :46
            interp_latents = tflib.slerp(latents_rec_2, latents_rec_1, interp_factors)          # :37
            interp_labels = tflib.lerp(labels_rec_2, labels_rec_1, interp_factors)              # :38
            labels_random = training_set.get_random_labels_tf(minibatch_size)                    # :47
            z_all = torch.cat([latents_rec_1, latents_rec_2, interp_latents, latents_random], dim=0)
            l_all = torch.cat([labels_rec_1, labels_rec_2, interp_labels, labels_random], dim=0)
            imgs = G.get_output_for(z_all, l_all, is_training=True, num_calls=4)
            n = minibatch_size
            arb_images_out = imgs[3 * n:]
            gen_255 = (imgs[:3 * n] + 1) * (255 / 2)                                             # :27-28,40
            reals_255 = (torch.cat([reals_rec_1, reals_rec_2], dim=0) + 1) * (255 / 2)           # :29-30

            with torch.no_grad():
                f_real = lpips_mod.features_of(lpips, reals_255)
            f_gen = lpips_mod.features_of(lpips, gen_255)
            # the four distances (rec_1/real_1, rec_2/real_2 :31; interp/real_2, interp/real_1 :41) in one pass over the features
            d_rec_1, d_rec_2, d_interp_2, d_interp_1 = lpips_mod.pair_distances_of(lpips, f_gen, f_real, n)

            loss_NN_rec_lpips = (d_rec_1 + d_rec_2) * 0.5   # :31
            loss_NN_rec_lpips = loss_NN_rec_lpips * NN_rec_lpips_weight
            loss_NN_rec_lpips = autosummary('Loss/loss_NN_rec_lpips', loss_NN_rec_lpips)
            loss = loss_addup(loss, loss_NN_rec_lpips)

            loss_NN_interp_lpips = tflib.lerp(d_interp_2, d_interp_1, interp_factors.squeeze(1))   # :41
            loss_NN_interp_lpips = loss_NN_interp_lpips * (NN_rec_lpips_weight * 0.4)
            loss_NN_interp_lpips = autosummary('Loss/loss_NN_interp_lpips', loss_NN_interp_lpips)
            loss = loss_addup(loss, loss_NN_interp_lpips)
        else:
            latents_random = tfutil.random_normal(latents_random_shape, dev)
            labels_random = training_set.get_random_labels_tf(minibatch_size)
            arb_images_out = G.get_output_for(latents_random, labels_random, is_training=True)

        # loss.py:49-52
        arb_scores_out, _ = D.get_output_for(arb_images_out, labels_random, is_training=True)
        loss_G_arb = F.softplus(-arb_scores_out)
        loss_G_arb = autosummary('Loss/loss_G_arb', loss_G_arb)
        loss = loss_addup(loss, loss_G_arb)

    # Path length regularization (loss.py:55-89).
    if phase in ('both', 'reg'):
        pl_minibatch = minibatch_size // pl_minibatch_shrink
        pl_latents = tfutil.random_normal([pl_minibatch] + latent_shape, dev)
        pl_labels = training_set.get_random_labels_tf(pl_minibatch)
        with hip_ops.second_order():      # pl_grads below is itself differentiated
            fake_images_out, fake_dlatents_out = G.get_output_for(pl_latents, pl_labels, is_training=True, return_dlatents=True)

        # Compute |J*y|.
        pl_noise = tfutil.random_normal(fake_images_out.shape, dev) / np.sqrt(np.prod(G.output_shape[2:]))
        pl_grads = torch.autograd.grad(torch.sum(fake_images_out * pl_noise), [fake_dlatents_out], create_graph=True)[0]
        pl_lengths = torch.sqrt(torch.mean(torch.sum(pl_grads * pl_grads, dim=2), dim=1))

        # Track exponential moving average of |J*y| (per replica, like the per-tower variable :70).
        if not hasattr(G, 'pl_mean_var'):
            G.pl_mean_var = torch.zeros((), device=dev, dtype=torch.float32)
        pl_mean_old = G.pl_mean_var.clone()
        pl_mean = pl_mean_old + pl_decay * (torch.mean(pl_lengths) - pl_mean_old)   # differentiable, like :71
        G.pl_mean_var.copy_(pl_mean.detach())                                        # tf.assign (:72)

        # Calculate (|J*y|-a)^2.
        pl_penalty = (pl_lengths - pl_mean) ** 2
        reg = pl_penalty * pl_weight
        reg = autosummary('Loss/pl_penalty', reg)

    return loss, reg

def D_logistic_r1(G, D, training_set, minibatch_size, reals, labels,
    gamma=10.0, phase='both'):
    with G.derived_scope(), D.derived_scope():
        return _D_loss_impl(G, D, training_set, minibatch_size, reals, labels, gamma, phase)

def _D_loss_impl(G, D, training_set, minibatch_size, reals, labels, gamma, phase):
    assert phase in ('both', 'loss', 'reg')
    dev = reals.device
    latent_shape = G.input_shapes[0][1:]
    loss = None
    reg = None

    need_reg = phase in ('both', 'reg')
    if need_reg:
        reals = reals.detach().requires_grad_(True)

    if phase in ('both', 'loss'):
        # loss.py:98-105
        latents_random = tfutil.random_normal([minibatch_size * 2] + latent_shape, dev)
        labels_random = training_set.get_random_labels_tf(minibatch_size * 2)
        with torch.no_grad():   # only D's trainables are differentiated in this step (training_loop.py:291)
            arb_images_out = G.get_output_for(latents_random, labels_random, is_training=True)
        gs = D.static_kwargs.get('mbstd_group_size', 6)
        if phase == 'loss' and arb_images_out.shape == reals.shape and (gs <= 1 or int(reals.shape[0]) % gs == 0):
            # D(fakes) and D(reals) (:101-102) as ONE pass: the two minibatches are interleaved so that
            # every sample keeps the minibatch-stddev group it has in a separate pass (all other layers are
            # per-sample), i.e. the scores are those of the two reference calls.
            perm, inv = _mbstd_preserving_interleave(int(reals.shape[0]), gs, dev)
            both = torch.cat([arb_images_out, reals], dim=0).index_select(0, perm)
            both_labels = torch.cat([labels_random, labels], dim=0).index_select(0, perm)
            scores, _ = D.get_output_for(both, both_labels, is_training=True)
            scores = scores.index_select(0, inv)
            arb_scores_out, real_scores_out = scores[:reals.shape[0]], scores[reals.shape[0]:]
        else:
            arb_scores_out, _ = D.get_output_for(arb_images_out, labels_random, is_training=True)
            real_scores_out, _ = D.get_output_for(reals, labels, is_training=True)
        loss_D = F.softplus(arb_scores_out) + F.softplus(-real_scores_out)
        loss_D = autosummary('Loss/loss_D', loss_D)
        loss = loss_addup(loss, loss_D)
    else:
        real_scores_out, _ = D.get_output_for(reals, labels, is_training=True)

    if need_reg:
        # loss.py:107-111
        real_grads = torch.autograd.grad(torch.sum(real_scores_out), [reals], create_graph=True)[0]
        gradient_penalty = torch.sum(real_grads * real_grads, dim=[1, 2, 3])
        reg = gradient_penalty * (gamma * 0.5)
        reg = autosummary('Loss/gradient_penalty_D', reg)

    return loss, reg

#----------------------------------------------------------------------------

_interleave_cache = {}

def _mbstd_preserving_interleave(n, group_size, device):
    """Index tensors (perm, inv) for evaluating two minibatches a, b of n samples each in one D pass.
    minibatch_stddev_layer (networks_stylegan2.py:132-144) groups sample i of an n-batch with
    {i mod M} where M = n / G, G = min(group_size, n).  In the combined 2n-batch M' = 2M, so placing
    a's group m at residue m and b's group m at residue M + m,
        combined[g * 2M + m]     = a[g * M + m],    combined[g * 2M + M + m] = b[g * M + m],
    reproduces both partitions.  perm indexes cat(a, b); inv undoes it."""
    key = (n, group_size, str(device))
    if key not in _interleave_cache:
        G = min(group_size, n) if group_size > 1 else 1
        assert n % G == 0
        M = n // G
        perm = np.empty(2 * n, dtype=np.int64)
        for g in range(G):
            for m in range(M):
                perm[g * 2 * M + m] = g * M + m
                perm[g * 2 * M + M + m] = n + g * M + m
        inv = np.empty_like(perm)
        inv[perm] = np.arange(2 * n)
        _interleave_cache[key] = (torch.from_numpy(perm).to(device), torch.from_numpy(inv).to(device))
    return _interleave_cache[key]

def loss_addup(loss, loss_):
    if loss is None:
        L = loss_
    else:
        L = loss + loss_
    return L
