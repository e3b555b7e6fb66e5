"""CPU: the plumbing of tests/reg_forms.py (state files, the oracle evaluation of the three ops, result files, deviations, table) on a tiny network -- the GPU test
tests/test_gpu_reg_forms.py and tools/reg_forms.py stand on it.  The "implementation under test" here is the oracle's own code evaluated in fp32 by PyTorch's CPU kernels
(the tool's `cpu32` variant): it must sit within fp32 rounding of the fp64 evaluation, and a planted error must show in the variable it was planted in."""
import numpy as np
import torch

from oracle import loss as OL
from oracle.misc import SeededRandom
from tests import reg_forms as RF


class _Rec(SeededRandom):
    def __init__(self, seed):
        super().__init__(seed, dtype=torch.float64)
        self.entries = []

    def normal(self, shape):
        t = super().normal(shape); self.entries.append(('normal', t.numpy())); return t

    def uniform(self, shape):
        t = super().uniform(shape); self.entries.append(('uniform', t.numpy())); return t

    def randint(self, lo, hi):
        v = super().randint(lo, hi); self.entries.append(('randint', np.asarray(v))); return v


def test_state_round_trip_oracle_ops_and_deviation_table(tmp_path):
    res, fmap, B = 16, 256, 3          # 2 B = 6 reals: one minibatch-stddev group
    G, D = RF.make_nets('cpu', res, fmap)
    state = dict(cfg=dict(res=res, fmap=fmap, B=B), G={n: v.detach().numpy() for n, v in G.vars.items()}, D={n: v.detach().numpy() for n, v in D.vars.items()},
                 pl_means=[0.0, 0.7], tape_G=[], tape_D=[], reals=(np.random.RandomState(0).rand(2 * B, 3, res, res).astype(np.float32) * 2 - 1))
    ocfg = dict(resolution=res, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
    gp = {n: torch.from_numpy(np.asarray(v, np.float64)).requires_grad_(n in G.trainables) for n, v in state['G'].items()}
    dp = {n: torch.from_numpy(np.asarray(v, np.float64)) for n, v in state['D'].items()}
    z = torch.zeros(B, 512, dtype=torch.float64)
    r = _Rec(0)
    OL.G_loss(gp, dp, {}, ocfg, r, B, None, z, None, z, 2.5, phase='reg', state={})          # records the path-length step's draws in the reference's order
    state['tape_G'] = r.entries
    r = _Rec(1)
    OL.D_loss(gp, dp, ocfg, r, B, torch.from_numpy(state['reals']).double(), gamma=100, phase='loss', state=dict(dlatent_avg=gp['dlatent_avg']))
    state['tape_Dloss'] = r.entries
    path = str(tmp_path / 'state.npz')
    RF.save_state_dict(path, state)
    back = RF.load_state(path)
    assert back['cfg'] == state['cfg'] and back['pl_means'] == [0.0, 0.7] and len(back['tape_G']) == len(state['tape_G']) and len(back['tape_Dloss']) == len(state['tape_Dloss'])
    assert all(np.array_equal(back['G'][n], state['G'][n]) for n in state['G']) and np.array_equal(back['reals'], state['reals'])
    names = dict(G=list(G.trainables), D=list(D.trainables))
    ops = ('G_reg', 'D_reg', 'D_loss')
    ora = RF.oracle_ops_of_state(back, ops=ops, trainables=names)
    f32 = RF.oracle_ops_of_state(back, ops=ops, trainables=names, dtype=torch.float32)
    assert set(ora) == {'G_reg@0', 'G_reg@1', 'D_reg', 'D_loss'}
    RF.save_result(str(tmp_path / 'f32.npz'), f32)
    f32 = RF.load_result(str(tmp_path / 'f32.npz'))
    devs = {'cpu32': {op: RF.deviations(f32[op], ora[op]) for op in ora}}
    for op in ora:
        assert max(devs['cpu32'][op]['errs'].values()) < 1e-3 and devs['cpu32'][op]['value'] < 1e-4, (op, devs['cpu32'][op])
    assert float(ora['G_reg@1']['value'].mean()) != float(ora['G_reg@0']['value'].mean())          # the second moving average is another problem
    # a planted error shows where it was planted
    bad = {op: dict(r, grads=dict(r['grads'])) for op, r in f32.items()}
    bad['D_loss']['grads']['4x4/Conv/weight'] = bad['D_loss']['grads']['4x4/Conv/weight'] * 1.01
    e = RF.deviations(bad['D_loss'], ora['D_loss'])['errs']
    assert max(e, key=e.get) == '4x4/Conv/weight' and 0.009 < e['4x4/Conv/weight'] < 0.011
    text = RF.table(devs, ['cpu32'])
    assert 'G_reg@1' in text and 'WORST' in text and 'D_loss' in text
