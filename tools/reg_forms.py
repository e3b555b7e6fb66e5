#!/usr/bin/env python3
"""The second-order steps (path-length regulariser, R1) of ONE state under several builds of the convolution arithmetic, against the fp64 oracle
(tests/reg_forms.py does the work; tests/test_gpu_reg_forms.py is the suite's fixed instance of it at the bench configuration).

    python tools/reg_forms.py --res 128 --fmap 8192 --B 6 --state init --pl-fracs 0,0.9,0.98 --variants "0;1;2"
    python tools/reg_forms.py --res 32 --fmap 8192 --B 6 --state loop:4,16 --variants "0;1;2;2:1024"

--state init          random initialisation, pl_mean = fraction x the batch's mean path length (one G_reg evaluation per fraction)
--state loop:i,j,...  the state training_loop() had BEFORE the path-length step of (0-based) iterations i, j, ... of a run under the default form
                      (weights, dlatent_avg, pl_mean and that op's own draws; tests/test_gpu_loop_parity.record_loop) -- every variant then
                      evaluates THAT state, unlike a comparison of separate runs, whose weights differ from the fourth iteration on (beta1 = 0)
--variants            ';'-separated "form[:min_rows[:wgrad_min_rows]]": IGAN_CONV_PLANES and the two row thresholds of the piece forms;
                      "cpu32" = the oracle's own restatement evaluated in fp32 by PyTorch's CPU kernels (a second, unrelated fp32 implementation:
                      what fp32 arithmetic as such costs on this state)
Prints per op the worst variables of every variant, SHA-1 digests of the gradient buckets (which variants are bit-identical), and writes a JSON."""
import argparse
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import inclusivegan_amd  # noqa: E402,F401
from tests import reg_forms as RF  # noqa: E402


def loop_states(res, fmap, B, iterations, op='G_reg'):
    """Pre-op states of the `op` ops ('G_reg' or 'D': the first-order D step) of the given iterations from one run of the real loop (graphs on, default form)."""
    from tests.test_gpu_loop_parity import record_loop, loop_kwargs, flat_of
    want = set(iterations)
    states = {}
    box = {}

    class Capture:
        def start(self, init):
            box['init'] = init
            box['pre'] = dict(G=flat_of(init, 'G'), D=flat_of(init, 'D'))
            box['pl_mean'] = 0.0
            box['dlatent_avg'] = init['G']['dlatent_avg'].copy()

        def __call__(self, rec):
            init = box['init']
            if rec['name'] == op and rec['it'] in want:
                def named(which, flat):
                    out = {n: np.asarray(v).copy() for n, v in init[which].items()}
                    for n, (o, c, shape) in init[which + '_layout'].items():
                        out[n] = flat[o:o + c].reshape(shape).copy()
                    return out
                Gv = named('G', box['pre']['G'])
                Gv['dlatent_avg'] = box['dlatent_avg'].copy()
                if op == 'G_reg':
                    states[rec['it']] = dict(cfg=dict(res=res, fmap=fmap, B=B), G=Gv, D=named('D', box['pre']['D']), pl_means=[box['pl_mean']],
                                             tape_G=rec['tape'], tape_D=[], reals=np.zeros((2 * B, 3, res, res), np.float32), hip_value=rec['value'])
                else:       # 'D': reals as the loop fed them (dataset range 0..255 -> [-1, 1], process_reals at lod 0 without mirroring)
                    from oracle.misc import adjust_dynamic_range
                    reals = np.ascontiguousarray(adjust_dynamic_range(rec['reals'].astype(np.float32), [0, 255], [-1, 1]), dtype=np.float32)
                    states[rec['it']] = dict(cfg=dict(res=res, fmap=fmap, B=B), G=Gv, D=named('D', box['pre']['D']), pl_means=[box['pl_mean']],
                                             tape_G=[], tape_D=[], tape_Dloss=rec['tape'], reals=reals, hip_value=rec['value'])
            key = 'G' if rec['name'].startswith('G') else 'D'
            box['pre'][key] = rec['post']['w' + key][:box['pre'][key].size].copy()
            box['pl_mean'] = rec['pl_mean']
            box['dlatent_avg'] = rec['dlatent_avg'].copy()

    record_loop(max(iterations) + 1, loop_kwargs(fmap, B, data_size=48, res=res), consumer=Capture())
    return states


def digest(result, op):
    h = hashlib.sha1()
    for n in sorted(result[op]['grads']):
        h.update(np.ascontiguousarray(result[op]['grads'][n]).tobytes())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--fmap', type=int, default=8192)
    ap.add_argument('--B', type=int, default=6)
    ap.add_argument('--state', default='init')
    ap.add_argument('--pl-fracs', default='0,0.9')
    ap.add_argument('--variants', default='0;1;2')
    ap.add_argument('--ops', default='G_reg,D_reg')
    ap.add_argument('--out', default=None)
    ap.add_argument('--keep-state', default=None, help='directory: keep every evaluated state file there as state_<k>.npz (for tools/conv_audit.py --state)')
    ap.add_argument('--loop-op', default='G_reg', help="with --state loop: which op's pre-state to take: G_reg or D (the first-order D step)")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    ops = tuple(a.ops.split(','))
    variants = []
    for v in a.variants.split(';'):
        if v == 'cpu32':
            variants.append((v, None))
            continue
        f = v.split(':')
        env = dict(IGAN_CONV_PLANES=f[0])
        if len(f) > 1 and f[1]:
            env['IGAN_PLANES_MIN_ROWS'] = f[1]
        if len(f) > 2 and f[2]:
            env['IGAN_WGRAD_PLANES_MIN_ROWS'] = f[2]
        elif len(f) > 1 and f[1]:
            env['IGAN_WGRAD_PLANES_MIN_ROWS'] = f[1]
        variants.append((v, env))
    if a.state == 'init':
        state, names = RF.init_state(dev, a.res, a.fmap, a.B, [float(x) for x in a.pl_fracs.split(',')])
        todo = [('init (pl_mean fractions %s of the mean path length %.5g)' % (a.pl_fracs, state['mean_path_length']), state)]
    else:
        its = [int(x) for x in a.state.split(':')[1].split(',')]
        st = loop_states(a.res, a.fmap, a.B, its, op=a.loop_op)
        G, D = RF.make_nets('cpu', a.res, a.fmap)
        names = dict(G=list(G.trainables), D=list(D.trainables))
        todo = [('loop state before %s of iteration %d (0-based), pl_mean %.6g' % (a.loop_op, it, st[it]['pl_means'][0]), st[it]) for it in its]
        ops = ('G_reg',) if a.loop_op == 'G_reg' else ('D_loss',)
    torch.cuda.empty_cache()
    record = []
    tmp = tempfile.mkdtemp(prefix='reg_forms_')
    for k, (title, state) in enumerate(todo):
        spath = os.path.join(tmp, 'state.npz')
        RF.save_state_dict(spath, state)
        if a.keep_state:
            os.makedirs(a.keep_state, exist_ok=True)
            RF.save_state_dict(os.path.join(a.keep_state, 'state_%d.npz' % k), state)
        hip, digs = {}, {}
        for label, env in variants:
            if env is None:
                hip[label] = RF.oracle_ops_of_state(state, ops=ops, trainables=names, dtype=torch.float32)
                digs[label] = {op: '-' for op in hip[label]}
                continue
            opath = os.path.join(tmp, 'out.npz')
            info = RF.run_child(spath, opath, env, ops=ops)
            hip[label] = RF.load_result(opath)
            digs[label] = {op: digest(hip[label], op) for op in hip[label]}
            os.remove(opath)
        ora = RF.oracle_ops_of_state(state, ops=ops, trainables=names)
        labels = [l for l, _ in variants]
        devs = {l: {op: RF.deviations(hip[l][op], ora[op]) for op in ora} for l in labels}
        print('## %s  (res %d, fmap %d, minibatch_gpu %d; variants = IGAN_CONV_PLANES[:row thresholds])' % (title, a.res, a.fmap, a.B))
        print(RF.table(devs, labels))
        for op in ora:
            print('  gradient digests %s: %s' % (op, '  '.join('%s %s' % (l, digs[l][op]) for l in labels)))
        sys.stdout.flush()
        record.append(dict(title=title, deviations=devs, digests=digs))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(record, f, indent=1)


if __name__ == '__main__':
    main()
