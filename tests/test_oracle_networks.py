"""CPU: self-consistency of the oracle networks (fused == non-fused modulated conv, fp64), shapes,
and the equivalence used by the HIP design: demodulation coefficients from sum_k w^2."""
import numpy as np
import torch

from oracle import networks_stylegan2 as N
from oracle.misc import SeededRandom, Tape


def _params(res=16, fmap=256, seed=0, dtype=torch.float64):
    """Random parameters with the reference's names, built by the product's Network on the meta device."""
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=fmap, device='cpu')
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=seed, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=seed + 1, **kw)
    rng = np.random.RandomState(seed)
    gp = {n: v.detach().to(dtype) for n, v in G.vars.items()}
    dp = {n: v.detach().to(dtype) for n, v in D.vars.items()}
    for p in (gp, dp):
        for n in p:
            if n.endswith('bias') or n.endswith('noise_strength'):
                p[n] = torch.from_numpy(np.asarray(rng.randn(*p[n].shape) * 0.1)).to(dtype).reshape(p[n].shape)
    return gp, dp


class _Replay:
    def __init__(self, seed):
        self.rec = []
        self.src = SeededRandom(seed, torch.float64)
    def normal(self, shape):
        v = self.src.normal(shape); self.rec.append(('normal', v.numpy())); return v
    def uniform(self, shape):
        v = self.src.uniform(shape); self.rec.append(('uniform', v.numpy())); return v
    def randint(self, lo, hi):
        v = self.src.randint(lo, hi); self.rec.append(('randint', np.asarray(v))); return v


def test_fused_equals_nonfused_modconv_through_generator():
    gp, _ = _params()
    z = torch.randn(3, 512, dtype=torch.float64)
    r = _Replay(1)
    a = N.G_main(gp, z, r, 16, fmap_base=256, architecture='skip', is_training=True, fused_modconv=True)
    b = N.G_main(gp, z, Tape(r.rec, torch.float64), 16, fmap_base=256, architecture='skip', is_training=True, fused_modconv=False)
    assert a.shape == (3, 3, 16, 16)
    assert (a - b).abs().max() / a.abs().max() < 1e-12


def test_demod_from_summed_squares():
    rng = np.random.RandomState(0)
    w = torch.from_numpy(rng.randn(3, 3, 5, 7)); s = torch.from_numpy(rng.randn(4, 5))
    ww = w[None] * s[:, None, None, :, None]
    d_ref = torch.rsqrt((ww * ww).sum(dim=[1, 2, 3]) + 1e-8)
    d = torch.rsqrt((s * s) @ (w * w).sum(dim=(0, 1)) + 1e-8)
    assert torch.allclose(d, d_ref, rtol=1e-12)


def test_discriminator_shapes_and_mbstd():
    _, dp = _params()
    img = torch.randn(6, 3, 16, 16, dtype=torch.float64)
    s, f = N.D_stylegan2_feature(dp, img, 16, fmap_base=256, architecture='resnet')
    assert s.shape == (6,)
    assert f.shape[0] == 6 and f.shape[1] == 3 * 256 + 32 * 256 + 64 * 64 + 128 * 16 + 128 * 16 + 256 + 1   # image, FromRGB, 16x16 block, 8x8 block, 4x4 conv, Dense0, Output
    x = torch.randn(12, 4, 4, 4, dtype=torch.float64)
    y = N.minibatch_stddev_layer(x, 6)
    assert y.shape == (12, 5, 4, 4)
    # sample n receives the statistic of group n % M (M = 2)
    assert torch.equal(y[0, 4], y[2, 4]) and not torch.equal(y[0, 4], y[1, 4])
    g0 = x[0::2]
    want = torch.sqrt(((g0 - g0.mean(0, keepdim=True)) ** 2).mean(0) + 1e-8).mean()
    assert torch.allclose(y[0, 4], want.expand(4, 4))


def test_zero_weight_reconstruction_terms_contribute_exactly_nothing():
    """BASELINE config 3 (NN_rec_lpips_weight = 0, adversarial only): the reference still evaluates the reconstruction and
    interpolation terms and multiplies them by the weight (training/loss.py:31-32,41-42).  Evaluated literally, they add
    exactly 0 to the loss and to every gradient -- so the HIP path (and the oracle by default) may skip them."""
    from oracle import loss as OL
    from oracle import lpips as OLP
    from inclusivegan_amd.dnnlib import tflib
    gp, dp = _params(res=16, fmap=128)
    for p in gp.values():
        p.requires_grad_(True)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=16, device='cpu', seed=3)
    lpo = {n: v.detach().double() for n, v in lp.vars.items()}
    cfg = dict(resolution=16, num_channels=3, fmap_base=128, G_arch='skip', D_arch='resnet')
    B = 2
    g = torch.Generator().manual_seed(0)
    r1 = torch.rand(B, 3, 16, 16, generator=g, dtype=torch.float64) * 2 - 1; r2 = torch.rand(B, 3, 16, 16, generator=g, dtype=torch.float64) * 2 - 1
    z1 = torch.randn(B, 512, generator=g, dtype=torch.float64); z2 = torch.randn(B, 512, generator=g, dtype=torch.float64)
    loss, _, terms = OL.G_loss(gp, dp, lpo, cfg, SeededRandom(5, torch.float64), B, r1, z1, r2, z2, 0.0, phase='loss', state={}, literal_zero_weight=True)
    assert set(terms) == {'loss_NN_rec_lpips', 'loss_NN_interp_lpips', 'loss_G_arb'}
    assert float(terms['loss_NN_rec_lpips'].abs().max()) == 0.0 and float(terms['loss_NN_interp_lpips'].abs().max()) == 0.0
    assert torch.equal(loss, terms['loss_G_arb'])
    params = [p for p in gp.values() if p.requires_grad]
    g_all = torch.autograd.grad(loss.mean(), params, retain_graph=True, allow_unused=True)
    g_adv = torch.autograd.grad(terms['loss_G_arb'].mean(), params, allow_unused=True)
    for a, b in zip(g_all, g_adv):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
