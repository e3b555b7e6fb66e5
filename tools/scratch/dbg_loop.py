import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_loop_parity import record_loop, loop_kwargs, replay_into_oracle
from tests.util import gloss_tape_in_reference_order
from oracle.train_ops import TrainOps
fmap = int(os.environ.get('FMAP', '1024'))
for mode in os.environ.get('MODES', '1,0').split(','):
    os.environ['IGAN_HIP_GRAPHS'] = mode
    log = record_loop(int(os.environ.get("ITERS", "6")), loop_kwargs(fmap, 6, data_size=48))
    init = log['init']
    cfg = dict(resolution=32, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
    ops = TrainOps(init['G'], init['D'], init['G_layout'], init['D_layout'], init['lpips'], cfg, world=1, minibatch_gpu=6)
    for r in log['ops']:
        name = r['name']
        if name in ('G', 'G_reg'):
            t = dict(r, tape=gloss_tape_in_reference_order(r['tape'], 6) if name == 'G' else r['tape'])
            v = ops.G_op([t], 'loss' if name == 'G' else 'reg')
        else:
            v = ops.D_op([r], 'loss' if name == 'D' else 'reg')
        if name == 'D':
            ops.Gs_update()
        print('graphs', mode, name, 'hip', r['value'], 'oracle', v[0], 'ntape', len(r['tape']), flush=True)
    print('pl_mean hip', log['final']['pl_mean'], 'oracle', float(ops.state[0]['pl_mean']))
