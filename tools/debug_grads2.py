import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from inclusivegan_amd import hip_ops
from inclusivegan_amd.dnnlib import tflib
from inclusivegan_amd.dnnlib.tflib import tfutil
from oracle import networks_stylegan2 as ON
from oracle.misc import Tape

dev = torch.device('cuda', 0)
RES, FMAP = 32, 1024
kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
z = torch.randn(3, 512, device=dev); lab = torch.zeros(3, 0, device=dev); wnoise = torch.randn(3, 3, RES, RES, device=dev)
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
rec = tfutil.RecordingRandom()
with tfutil.use_random(rec):
    img, dl = G.get_output_for(z, lab, is_training=True, return_dlatents=True)
gp = {n: v.detach().double().cpu() for n, v in G.vars.items()}
for n in G.trainables:
    gp[n].requires_grad_(True)
img_o, dl_o = ON.G_main(gp, z.double().cpu(), Tape(rec.entries, torch.float64), RES, fmap_base=FMAP, architecture='skip', is_training=True, return_dlatents=True)
(go,) = torch.autograd.grad((img_o * wnoise.double().cpu()).sum(), [dl_o])
(ga,) = torch.autograd.grad((img * wnoise).sum(), [dl], retain_graph=True)
print('A fused first-order pl_grads err', rel(ga, go))
(gb,) = torch.autograd.grad((img * wnoise).sum(), [dl], create_graph=True)
print('B create_graph pl_grads err', rel(gb, go))
# C: force the composite branch for a forward built only from composite ops
orig = hip_ops.ModConv2dFn.apply
class Comp:
    @staticmethod
    def apply(x, w, s, d, geom, out_hw):
        return hip_ops.modconv_composite(x, w, s, d, geom, out_hw)
hip_ops.ModConv2dFn.apply = Comp.apply
with tfutil.use_random(tfutil.RandomTape(rec.entries)):
    img2, dl2 = G.get_output_for(z, lab, is_training=True, return_dlatents=True)
print('C composite forward img err', rel(img2, img_o))
(gc,) = torch.autograd.grad((img2 * wnoise).sum(), [dl2], retain_graph=True)
print('C composite first-order pl_grads err', rel(gc, go))
(gd,) = torch.autograd.grad((img2 * wnoise).sum(), [dl2], create_graph=True)
print('D composite create_graph pl_grads err', rel(gd, go))
