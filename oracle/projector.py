"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The LPIPS projector of projector_lpips.py:46-162 restated on the oracle networks (torch CPU, fp64 forward / gradient; the Adam
update in float32 NumPy exactly as the reference's SimpleAdam arithmetic, oracle/optimizer.py): per step
    t = step / num_steps; noise_strength = f0 * max(0, 1 - t / noise_ramp)^2                                      (:131-132)
    lr = lr0 * (0.5 - 0.5 cos(pi * min(1, (1 - t) / rampdown))) * min(1, t / rampup)                                (:133-136)
    latents_expr = slerp(z, noise, noise_strength); loss = sum LPIPS((G(latents_expr) + 1) * 127.5, target)         (:59-80)
    z <- Adam(z, d loss / d z, lr; beta1 .9, beta2 .999, eps 1e-8)                                                  (:83-86)
Parity unpinned at the reference level (TensorFlow graph code); the arithmetic follows the cited lines."""
import numpy as np
import torch

from . import networks_stylegan2 as N
from . import lpips as L
from .misc import slerp_t
from .optimizer import SimpleAdam


def schedule(step, num_steps, lr0=0.1, f0=0.05, rampdown=0.25, rampup=0.05, noise_ramp=0.75):
    t = step / num_steps
    noise_strength = f0 * max(0.0, 1.0 - t / noise_ramp) ** 2
    lr_ramp = min(1.0, (1.0 - t) / rampdown)
    lr_ramp = 0.5 - 0.5 * np.cos(lr_ramp * np.pi)
    lr_ramp = lr_ramp * min(1.0, t / rampup)
    return noise_strength, lr0 * lr_ramp


def forward(G_params, lpips_params, cfg, z, noise, noise_strength, target_255, rand):
    latents = slerp_t(z, noise, noise_strength)
    img = N.G_main(G_params, latents, rand, cfg['resolution'], num_channels=cfg['num_channels'], fmap_base=cfg['fmap_base'],
                   architecture=cfg['G_arch'], is_validation=True, fused_modconv=False)
    dist = L.lpips(lpips_params, (img + 1) * (255 / 2), target_255)
    return latents, img, dist


class ProjectorOracle:
    def __init__(self, G_params, lpips_params, cfg, num_steps, init_latents, target_images, f0=0.05):
        self.G, self.lp, self.cfg = G_params, lpips_params, cfg
        self.num_steps, self.f0 = num_steps, f0
        self.z = np.asarray(init_latents, dtype=np.float32).copy()
        self.target = (torch.as_tensor(np.asarray(target_images), dtype=torch.float64) + 1) * (255 / 2)
        self.adam = SimpleAdam(self.z.size, 0.1, 0.9, 0.999, 1e-8)
        self.step_idx = 0

    def step(self, noise, rand):
        ns, lr = schedule(self.step_idx, self.num_steps, f0=self.f0)
        z = torch.from_numpy(self.z).double().requires_grad_(True)
        _, _, dist = forward(self.G, self.lp, self.cfg, z, torch.as_tensor(noise, dtype=torch.float64), ns, self.target, rand)
        loss = dist.sum()
        (g,) = torch.autograd.grad(loss, [z])
        flat = self.z.reshape(-1)
        self.adam.apply(flat, g.numpy().astype(np.float32).reshape(-1), lr=lr)
        self.z = flat.reshape(self.z.shape)
        self.step_idx += 1
        return dist.detach().numpy(), float(loss)
