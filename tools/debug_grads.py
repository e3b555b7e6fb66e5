"""Debug helper: per-parameter gradient error of the HIP G path vs the fp64 oracle (not a test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from inclusivegan_amd.dnnlib import tflib
from inclusivegan_amd.dnnlib.tflib import tfutil
from oracle import networks_stylegan2 as ON
from oracle.misc import Tape

print('cpu_count', os.cpu_count())
dev = torch.device('cuda', 0)
RES, FMAP = 32, 1024
kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
rng = np.random.RandomState(0)
with torch.no_grad():
    for n, v in G.vars.items():
        if n.endswith('bias') or n.endswith('noise_strength'):
            v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(dev).reshape(v.shape))
z = torch.randn(3, 512, device=dev)
lab = torch.zeros(3, 0, device=dev)
wnoise = torch.randn(3, 3, RES, RES, device=dev)

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

for mode in ('first', 'create_graph'):
    G.zero_grad()
    rec = tfutil.RecordingRandom()
    with tfutil.use_random(rec):
        img, dl = G.get_output_for(z, lab, is_training=True, return_dlatents=True)
    gp = {n: v.detach().double().cpu() for n, v in G.vars.items()}
    for n in G.trainables:
        gp[n].requires_grad_(True)
    img_o, dl_o = ON.G_main(gp, z.double().cpu(), Tape(rec.entries, torch.float64), RES, fmap_base=FMAP, architecture='skip', is_training=True, return_dlatents=True)
    print(mode, 'img err', rel(img, img_o))
    if mode == 'first':
        (img * wnoise).sum().backward()
        (img_o * wnoise.double().cpu()).sum().backward()
    else:
        (g,) = torch.autograd.grad((img * wnoise).sum(), [dl], create_graph=True)
        (go,) = torch.autograd.grad((img_o * wnoise.double().cpu()).sum(), [dl_o], create_graph=True)
        print('  pl_grads err', rel(g, go))
        for li in range(g.shape[1]):
            print('   layer %d err %.2e' % (li, rel(g[:, li], go[:, li])))
        (g * g).sum().backward()
        (go * go).sum().backward()
    errs = {n: rel(v.grad, gp[n].grad) for n, v in G.trainables.items() if gp[n].grad is not None}
    for n, e in sorted(errs.items(), key=lambda kv: -kv[1])[:14]:
        extra = ' hip %.5e oracle %.5e' % (float(G.trainables[n].grad), float(gp[n].grad)) if G.trainables[n].numel() == 1 else ''
        print('  %-44s %.2e%s' % (n, e, extra))
