import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from inclusivegan_amd import hip_ops
from inclusivegan_amd.dnnlib import tflib
from inclusivegan_amd.dnnlib.tflib import tfutil
from inclusivegan_amd.training import loss as PL
from inclusivegan_amd.training.dataset import SyntheticDataset
from oracle import loss as OL
from oracle.misc import Tape
dev = torch.device('cuda', 0)
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
for shape in [(12, 512, 8, 8), (12, 256, 16, 16), (12, 128, 32, 32), (6, 512, 4, 4), (24, 64, 16, 16)]:
    x = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)
    db = hip_ops.bias_grad_raw(x, shape[1], 1)
    print('bias_grad', shape, rel(db, x.double().sum(dim=(0, 2, 3))))
RES, FMAP = 32, 1024
kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=11, **kw)
D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=12, **kw)
rng = np.random.RandomState(0)
with torch.no_grad():
    for net in (G, D):
        for n, v in net.vars.items():
            if n.endswith('bias') or n.endswith('noise_strength'):
                v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(dev).reshape(v.shape))
ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
B = 6
g = torch.Generator().manual_seed(5)
reals = torch.rand(2 * B, 3, RES, RES, generator=g) * 2 - 1
cfg = dict(resolution=RES, num_channels=3, fmap_base=FMAP, G_arch='skip', D_arch='resnet')
rec = tfutil.RecordingRandom()
gp = {n: v.detach().double().cpu() for n, v in G.vars.items()}
dp = {n: v.detach().double().cpu().requires_grad_(n in D.trainables) for n, v in D.vars.items()}
with tfutil.use_random(rec):
    loss, _ = PL.D_logistic_r1(G, D, ts, B, reals.to(dev).contiguous(memory_format=torch.channels_last), torch.zeros(2 * B, 0, device=dev), gamma=100, phase='loss')
D.zero_grad()
torch.autograd.backward(loss.mean(), inputs=list(D.trainables.values()))
lo, _, _ = OL.D_loss(gp, dp, cfg, Tape(rec.entries, torch.float64), B, reals.double(), gamma=100, phase='loss', state={})
lo.mean().backward()
print('loss err', rel(loss, lo))
errs = {n: rel(v.grad, dp[n].grad) for n, v in D.trainables.items()}
for n, e in sorted(errs.items(), key=lambda kv: -kv[1])[:12]:
    print('  %-30s %.2e  max|g| %.3e' % (n, e, float(dp[n].grad.abs().max())))
