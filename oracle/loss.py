"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of training/loss.py: G_logistic_ns_rec_interp_arb_pathreg :19-91 and
D_logistic_r1 :93-113, on the oracle networks.  Random draws from `rand` in the reference's order.
PINNED (round 4): tests/golden/ref_train_golden.npz holds the outputs of the reference's own loss.py executed under tests/golden/np_tf.py
(both functions, weights 2.5 and 0, every autosummary term, pl_mean update); tests/test_ref_train_golden.py requires this file to reproduce
them to 1e-9.  `tf.gradients` is the one statement that cannot be executed (inherent).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import networks_stylegan2 as N
from . import lpips as L
from .misc import lerp, slerp_t


def G_loss(G_params, D_params, lpips_params, cfg, rand, minibatch_size, reals_rec_1, latents_rec_1, reals_rec_2, latents_rec_2,
           NN_rec_lpips_weight, pl_minibatch_shrink=2, pl_decay=0.01, pl_weight=2.0, phase='both', state=None, literal_zero_weight=False):
    """cfg: dict(resolution, num_channels, fmap_base, G_arch, D_arch).  state: dict with 'pl_mean', 'dlatent_avg'.
    The reference evaluates the reconstruction / interpolation terms unconditionally and multiplies them by
    NN_rec_lpips_weight (:31-32,41-42); with a weight of 0 they contribute exactly 0 to the value and to every gradient, so by
    default they are skipped here (as on the HIP path).  literal_zero_weight=True evaluates them anyway, like the reference
    graph does -- tests/test_oracle_networks.py uses it to show that skipping is exact."""
    state = state if state is not None else {}
    dt = latents_rec_1.dtype
    def G(z, **kw):
        p = dict(G_params)
        if 'dlatent_avg' in state:
            p['dlatent_avg'] = state['dlatent_avg']
        return N.G_main(p, z, rand, cfg['resolution'], num_channels=cfg['num_channels'], fmap_base=cfg['fmap_base'],
                        architecture=cfg['G_arch'], is_training=True, state=state, fused_modconv=cfg.get('fused_modconv', True),
                        **cfg.get('G_kwargs', {}), **kw)
    latent_size = int(latents_rec_1.shape[1])        # G.input_shapes[0][1:] (:46,59); 512 in every BASELINE config
    def D(img):
        return N.D_stylegan2_feature(D_params, img, cfg['resolution'], num_channels=cfg['num_channels'], fmap_base=cfg['fmap_base'],
                                     architecture=cfg['D_arch'])
    loss = reg = None
    terms = {}
    if phase in ('both', 'loss'):
        if NN_rec_lpips_weight != 0 or literal_zero_weight:
            rec1 = (G(latents_rec_1) + 1) * (255 / 2)                                    # :25-28
            rec2 = (G(latents_rec_2) + 1) * (255 / 2)
            real1 = (reals_rec_1 + 1) * (255 / 2)                                        # :29-30
            real2 = (reals_rec_2 + 1) * (255 / 2)
            t = rand.uniform([minibatch_size, 1]).to(dt)                                 # :36
            zt = slerp_t(latents_rec_2, latents_rec_1, t)                                # :37
            interp = (G(zt) + 1) * (255 / 2)                                             # :39-40
            l_rec = (L.lpips(lpips_params, rec1, real1) + L.lpips(lpips_params, rec2, real2)) * 0.5 * NN_rec_lpips_weight   # :31-32
            l_int = lerp(L.lpips(lpips_params, interp, real2), L.lpips(lpips_params, interp, real1), t.squeeze(1)) * (NN_rec_lpips_weight * 0.4)  # :41-42
            terms['loss_NN_rec_lpips'] = l_rec
            terms['loss_NN_interp_lpips'] = l_int
            loss = l_rec + l_int
        z = rand.normal([minibatch_size, latent_size]).to(dt)                            # :46
        scores, _ = D(G(z))                                                              # :48-49
        l_adv = F.softplus(-scores)                                                      # :50
        terms['loss_G_arb'] = l_adv
        loss = l_adv if loss is None else loss + l_adv
    if phase in ('both', 'reg'):
        pl_minibatch = minibatch_size // pl_minibatch_shrink                             # :58
        z = rand.normal([pl_minibatch, latent_size]).to(dt)                              # :59
        imgs, dl = G(z, return_dlatents=True)                                            # :61
        noise = rand.normal(list(imgs.shape)).to(dt) / np.sqrt(np.prod(imgs.shape[2:]))  # :64
        g = torch.autograd.grad(torch.sum(imgs * noise), [dl], create_graph=True)[0]     # :65
        pl_lengths = torch.sqrt(torch.mean(torch.sum(g * g, dim=2), dim=1))              # :66
        pl_mean_var = state.get('pl_mean', torch.zeros((), dtype=dt))
        pl_mean = pl_mean_var + pl_decay * (torch.mean(pl_lengths) - pl_mean_var)        # :71
        state['pl_mean'] = pl_mean.detach()                                              # :72
        reg = (pl_lengths - pl_mean) ** 2 * pl_weight                                    # :76,88
        terms['pl_penalty'] = reg
    return loss, reg, terms


def D_loss(G_params, D_params, cfg, rand, minibatch_size, reals, gamma=10.0, phase='both', state=None):
    state = state if state is not None else {}
    dt = reals.dtype
    def D(img):
        return N.D_stylegan2_feature(D_params, img, cfg['resolution'], num_channels=cfg['num_channels'], fmap_base=cfg['fmap_base'],
                                     architecture=cfg['D_arch'])
    loss = reg = None
    terms = {}
    if phase in ('both', 'reg'):
        reals = reals.detach().requires_grad_(True)
    if phase in ('both', 'loss'):
        z = rand.normal([minibatch_size * 2, cfg.get('latent_size', 512)]).to(dt)        # :98
        p = dict(G_params)
        if 'dlatent_avg' in state:
            p['dlatent_avg'] = state['dlatent_avg']
        with torch.no_grad():
            fakes = N.G_main(p, z, rand, cfg['resolution'], num_channels=cfg['num_channels'], fmap_base=cfg['fmap_base'],
                             architecture=cfg['G_arch'], is_training=True, state=state, fused_modconv=cfg.get('fused_modconv', True),
                             **cfg.get('G_kwargs', {}))
        fake_scores, _ = D(fakes)                                                        # :101
        real_scores, _ = D(reals)                                                        # :102
        loss = F.softplus(fake_scores) + F.softplus(-real_scores)                        # :103
        terms['loss_D'] = loss
    else:
        real_scores, _ = D(reals)
    if phase in ('both', 'reg'):
        rg = torch.autograd.grad(torch.sum(real_scores), [reals], create_graph=True)[0]  # :108
        reg = torch.sum(rg * rg, dim=[1, 2, 3]) * (gamma * 0.5)                          # :109-110
        terms['gradient_penalty_D'] = reg
    return loss, reg, terms
