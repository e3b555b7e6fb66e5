import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from inclusivegan_amd.dnnlib import tflib
from inclusivegan_amd.dnnlib.tflib import tfutil
from inclusivegan_amd.training import loss as PL
from inclusivegan_amd.training.dataset import SyntheticDataset
from oracle import loss as OL
from oracle.misc import Tape
dev = torch.device('cuda', 0)
RES, FMAP, B = 32, 1024, 6
kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=12, **kw)
ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
cfg = dict(resolution=RES, num_channels=3, fmap_base=FMAP, G_arch='skip', D_arch='resnet')
lab = torch.zeros(B, 0, device=dev)
z = torch.zeros(B, 512, device=dev)
for seed in range(12):
    torch.manual_seed(seed)
    G.zero_grad(); D.requires_grad_(False)
    G.pl_mean_var = torch.zeros((), device=dev)
    rec = tfutil.RecordingRandom()
    with tfutil.use_random(rec):
        _, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, None, ts, B, None, lab, z, None, lab, z, NN_rec_lpips_weight=2.5, phase='reg')
    torch.autograd.backward((reg * 4).mean(), inputs=list(G.trainables.values()))
    D.requires_grad_(True)
    gp = {n: v.detach().double().cpu() for n, v in G.vars.items()}
    for n in G.trainables: gp[n].requires_grad_(True)
    dp = {n: v.detach().double().cpu() for n, v in D.vars.items()}
    _, ro, _ = OL.G_loss(gp, dp, {}, cfg, Tape(rec.entries, torch.float64), B, None, z.double().cpu(), None, z.double().cpu(), 2.5, phase='reg', state={})
    (ro * 4).mean().backward()
    num = den = 0.0; worst = ('', 0.0)
    mnum = mden = 0.0
    for n, v in G.trainables.items():
        go = gp[n].grad
        if go is None: continue
        gh = v.grad.detach().double().cpu()
        e = float((gh - go).norm() / (go.norm() + 1e-30))
        num += float((gh - go).norm() ** 2); den += float(go.norm() ** 2)
        if 'G_mapping' in n: mnum += float((gh - go).norm() ** 2); mden += float(go.norm() ** 2)
        if e > worst[1]: worst = (n, e)
    u, r = float(rec.entries[2][1]), int(rec.entries[3][1])
    print('seed %2d coin %.3f cutoff %d | value rel %.2e | grad rel L2 all %.2e mapping %.2e worst %s %.2e' % (
        seed, u, r, abs(float(reg.mean()) - float(ro.mean())) / float(ro.mean()), (num / den) ** .5, (mnum / mden) ** .5, worst[0], worst[1]), flush=True)
