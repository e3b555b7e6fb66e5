#!/usr/bin/env python3
"""Where does the fp32 path take the OTHER branch of a leaky ReLU than fp64?  (Round 6: on some states of the config-2 run every arithmetic form of the HIP path sits 2e-3 ... 9e-3
from the fp64 oracle on the same variables to three digits while every single kernel call is at 1e-7 of fp64 from its own inputs -- profiles/r06_second_order.txt section 7.)
A pre-activation within fp32 rounding of zero lands on the other side of the kink than its fp64 twin; the derivative of that ONE unit changes from gain to 0.2 gain, and with it one
summand of every gradient below.  This tool runs D's forward pass of the first-order D step of a state (tools/reg_forms.py --keep-state) on the HIP path and on the fp64 oracle with
the same draws, and lists the feature-map entries (D_stylegan2_feature's features_out: the input image, FromRGB, every block, 4x4/Conv, Dense0, Output) whose SIGN differs.
usage: python tools/kink_flips.py <state.npz>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import inclusivegan_amd  # noqa: E402,F401
from tests import reg_forms as RF  # noqa: E402


def main():
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from oracle import networks_stylegan2 as ON
    from oracle.misc import Tape
    state = RF.load_state(sys.argv[1])
    cfg = state['cfg']
    res, fmap, B = cfg['res'], cfg['fmap'], cfg['B']
    dev = torch.device('cuda', 0)
    G, D = RF.make_nets(dev, res, fmap)
    RF._assign(G, state['G']); RF._assign(D, state['D'])
    reals = torch.from_numpy(state['reals']).to(dev).contiguous(memory_format=torch.channels_last)
    n = int(reals.shape[0])
    lab = torch.zeros(n, 0, device=dev)
    with torch.no_grad(), tfutil.use_random(tfutil.RandomTape(state['tape_Dloss'])):
        z = tfutil.random_normal([n, 512], dev)                       # training/loss.py:98
        fakes = G.get_output_for(z, lab, is_training=True)
        _, f_fake = D.get_output_for(fakes, lab, is_training=True, return_features=True)
        _, f_real = D.get_output_for(reals, lab, is_training=True, return_features=True)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    gp = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in state['G'].items()}
    dp = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in state['D'].items()}
    tape = Tape(state['tape_Dloss'], torch.float64)
    with torch.no_grad():
        zo = tape.normal([n, 512])
        fo = ON.G_main(gp, zo, tape, res, fmap_base=fmap, architecture='skip', is_training=True, state=dict(dlatent_avg=gp['dlatent_avg']))
        _, o_fake = ON.D_stylegan2_feature(dp, fo, res, fmap_base=fmap, architecture='resnet')
        _, o_real = ON.D_stylegan2_feature(dp, torch.from_numpy(state['reals']).double(), res, fmap_base=fmap, architecture='resnet')
    # segments of features_out in the order D_stylegan2_feature appends them (resnet: one FromRGB)
    nf = lambda stage: int(np.clip(int(fmap / (2.0 ** stage)), 1, 512))
    r2 = int(np.log2(res))
    segs = [('input image', 3 * res * res), ('%dx%d/FromRGB' % (res, res), nf(r2 - 1) * res * res)]
    for r in range(r2, 2, -1):
        segs.append(('%dx%d block (Conv0, Conv1_down + Skip)' % (2 ** r, 2 ** r), nf(r - 2) * (2 ** (r - 1)) ** 2))
    segs += [('4x4/Conv', nf(1) * 16), ('4x4/Dense0', nf(0)), ('Output', 1)]
    assert sum(s for _, s in segs) == int(f_fake.shape[1]) == int(o_fake.shape[1]), (sum(s for _, s in segs), f_fake.shape, o_fake.shape)
    print("# sign flips between the HIP path (fp32) and the fp64 oracle in D's feature maps, %d fakes + %d reals; a flipped entry is listed with its value relative to the rms of its segment" % (n, n))
    for name, fh, fo_ in (('fakes', f_fake, o_fake), ('reals', f_real, o_real)):
        fh = fh.double().cpu()
        off = 0
        for seg, size in segs:
            a, b = fh[:, off:off + size], fo_[:, off:off + size]
            flip = (torch.sign(a) != torch.sign(b)) & (a != 0) & (b != 0)
            rms = float(b.pow(2).mean().sqrt())
            rel = float((a - b).norm() / b.norm())
            line = '%-6s %-44s %8d entries  rel L2 %.2e  sign flips %d' % (name, seg, a.numel(), rel, int(flip.sum()))
            if int(flip.sum()):
                idx = flip.nonzero()[:6]
                line += '   ' + '; '.join('sample %d entry %d: hip %.2e oracle %.2e (rms %.2e)' % (int(i), int(j), float(a[i, j]), float(b[i, j]), rms) for i, j in idx)
            print(line)
            off += size


if __name__ == '__main__':
    main()
