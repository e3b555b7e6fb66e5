"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The device side of one training iteration of the reference, `training/training_loop.py`, restated on the oracle networks /
losses / Adam so that a recorded run of the HIP loop can be replayed op by op:

    :242-255   one optimizer per network and role; with lazy regularisation the learning rate is scaled by
               mb_ratio = interval / (interval + 1) and beta1, beta2 are raised to that power; the regularisation optimizer
               shares the main one's slots (`share=`)
    :257-291   one tower per GPU: each evaluates the losses on ITS slice of the minibatch with its own random draws, its own
               `pl_mean` and (tower 0 = the network itself, the others clones) its own `dlatent_avg`;
               G_opt.register_gradients(mean(G_loss)), G_reg_opt.register_gradients(mean(G_reg * G_reg_interval)), same for D
    dnnlib/tflib/optimizer.py:169-201   gradients scaled by 1 / num_gpus and summed over the towers
    :222,296   Gs <- lerp(G, Gs, 0.5 ** (minibatch_size / (G_smoothing_kimg * 1000))), non-trainables copied
    :474-479   op order G_train, [G_reg], D_train, Gs_update, [D_reg]  (driven by the caller from the recorded log)

Weights live as flat fp32 NumPy arrays in the HIP buckets' layout (`layout`: name -> (offset, count, shape)); every op builds
`dtype` (fp64 by default) parameter tensors from them, evaluates oracle/loss.py, and applies oracle/optimizer.py.
Round 4: the pieces it composes are pinned to the reference's executed code -- the losses (oracle/loss.py), the optimizer set-up, the
registered objectives and Gs_beta (oracle/training_loop.py helpers), scaling / summation / gate / Adam (oracle/optimizer.py).
"""
import numpy as np
import torch

from . import loss as OL
from . import optimizer as OO
from . import training_loop as OT
from .misc import adjust_dynamic_range, Tape


class TrainOps:
    def __init__(self, G_vars, D_vars, G_layout, D_layout, lpips_params, cfg, *, world=1, minibatch_gpu, lrate=0.002,
                 lazy_regularization=True, G_reg_interval=4, D_reg_interval=16, beta1=0.0, beta2=0.99, epsilon=1e-8,
                 G_smoothing_kimg=10.0, NN_rec_lpips_weight=2.5, gamma=100.0, drange_data=(0, 255), drange_net=(-1, 1),
                 dtype=torch.float64):
        self.cfg, self.world, self.B, self.dtype = cfg, world, minibatch_gpu, dtype
        self.lpips = {n: torch.as_tensor(np.asarray(v)).to(dtype) for n, v in lpips_params.items()}
        self.layout = dict(G=G_layout, D=D_layout)
        self.fixed = {}     # non-trainable variables (dlatent_avg, lod, ...), name -> tensor
        self.w = {}
        for key, vars_, layout in (('G', G_vars, G_layout), ('D', D_vars, D_layout)):
            n = max(off + cnt for off, cnt, _ in layout.values())
            flat = np.zeros(n, np.float32)
            for name, (off, cnt, _) in layout.items():
                flat[off:off + cnt] = np.asarray(vars_[name], np.float32).reshape(-1)
            self.w[key] = flat
            self.fixed[key] = {name: torch.as_tensor(np.asarray(v)).to(dtype) for name, v in vars_.items() if name not in layout}
        self.w['Gs'] = self.w['G'].copy()
        self.lw, self.gamma = NN_rec_lpips_weight, gamma
        self.interval = dict(G=G_reg_interval, D=D_reg_interval)
        self.adam = {}
        for key in ('G', 'D'):
            lr, b1, b2 = OT.lazy_regularization_args(lrate, beta1, beta2, self.interval[key], lazy_regularization)      # :247-251
            self.adam[key] = OO.SimpleAdam(self.w[key].size, lr, b1, b2, epsilon)
        self.Gs_beta = OT.smoothing_beta(minibatch_gpu * world, G_smoothing_kimg)                   # :222
        self.drange = (list(drange_data), list(drange_net))
        # per-tower state (:70 pl_mean under the tower's scope; dlatent_avg of the tower's own G)
        avg0 = self.fixed['G']['dlatent_avg']
        self.state = [dict(dlatent_avg=avg0.clone(), pl_mean=torch.zeros((), dtype=dtype)) for _ in range(world)]

    # ------------------------------------------------------------------
    def params(self, key, grad):
        p = dict(self.fixed[key])
        for name, (off, cnt, shape) in self.layout[key].items():
            t = torch.from_numpy(self.w[key][off:off + cnt].astype(np.float64)).to(self.dtype).reshape(tuple(shape))
            p[name] = t.requires_grad_(True) if grad else t
        return p

    def _flat_grad(self, key, p):
        g = np.zeros(self.w[key].size, np.float32)
        for name, (off, cnt, _) in self.layout[key].items():
            if p[name].grad is not None:
                g[off:off + cnt] = p[name].grad.reshape(-1).numpy().astype(np.float32)
        return g

    def _reals(self, x):
        """process_reals (:40-60) without mirroring at lod 0: cast + dynamic range."""
        x = adjust_dynamic_range(np.asarray(x).astype(np.float32), self.drange[0], self.drange[1])
        return torch.from_numpy(np.ascontiguousarray(x)).to(self.dtype)

    def average(self, grads):
        total = np.zeros_like(grads[0])
        for g in grads:                                    # optimizer.py:186,199: scaled by 1 / num_gpus, then summed
            total += g * np.float32(1.0 / self.world)
        return total

    # ------------------------------------------------------------------
    def G_op(self, towers, phase, apply=True):
        """towers: one dict per GPU with reals_rec_1/2 (dataset range), latents_rec_1/2, tape (reference call order).
        -> (per-tower mean loss, averaged flat fp32 gradient); apply=False leaves the weights alone (the caller compares first)."""
        values, grads = [], []
        for r, t in enumerate(towers):
            gp, dp = self.params('G', True), self.params('D', False)
            z = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dtype)
            lo, ro, _ = OL.G_loss(gp, dp, self.lpips, self.cfg, Tape(t['tape'], self.dtype), self.B,
                                  self._reals(t['reals_rec_1']), z(t['latents_rec_1']), self._reals(t['reals_rec_2']), z(t['latents_rec_2']),
                                  self.lw, phase=phase, state=self.state[r])
            v = lo if phase == 'loss' else ro * self.interval['G']              # :288,290
            v.mean().backward()
            values.append(float((lo if phase == 'loss' else ro).detach().mean()))
            grads.append(self._flat_grad('G', gp))
        avg = self.average(grads)
        if apply:
            self.adam['G'].apply(self.w['G'], avg)
        return values, avg

    def D_op(self, towers, phase, apply=True):
        """towers: one dict per GPU with reals (uint8 / dataset range, this tower's slice) and tape."""
        values, grads = [], []
        for r, t in enumerate(towers):
            gp, dp = self.params('G', False), self.params('D', True)
            lo, ro, _ = OL.D_loss(gp, dp, self.cfg, Tape(t['tape'], self.dtype), self.B, self._reals(t['reals']),
                                  gamma=self.gamma, phase=phase, state=self.state[r])
            v = lo if phase == 'loss' else ro * self.interval['D']              # :289,291
            v.mean().backward()
            values.append(float((lo if phase == 'loss' else ro).detach().mean()))
            grads.append(self._flat_grad('D', dp))
        avg = self.average(grads)
        if apply:
            self.adam['D'].apply(self.w['D'], avg)
        return values, avg

    def Gs_update(self):
        self.w['Gs'] = OO.ema(self.w['Gs'], self.w['G'], self.Gs_beta)         # network.py:341-351
