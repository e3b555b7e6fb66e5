"""Parity of the path that bench.py times: `training_loop()` itself, with its four training ops captured into hipGraphs and
replayed, against the oracle (oracle/train_ops.py: oracle losses + NumPy SimpleAdam in the reference's op order,
training_loop.py:242-297,466-479).

How a graph replay is compared with a CPU restatement: the loop runs with a `TapRandom` source (tflib/tfutil.py) that keeps
the device tensors of every random draw; after each op a hook copies them, the op's static inputs, its loss output, the
gradient bucket and the weights to the host.  The oracle then evaluates THE SAME op from the state the HIP path had before it
(teacher forcing) -- same weights, same fed batch, same draws, same pl_mean / dlatent_avg:

    loss value of the op                       2e-4 relative (regularisers 1e-3: built from fp32 gradients)
    gradient bucket after the op               5e-3 relative L2 per variable (lrelu kinks, see tests/test_gpu_networks.py)
    weights after the op                       == oracle SimpleAdam applied to the HIP gradient, 1e-6 relative
    pl_mean, dlatent_avg, Gs, beta powers      1e-4 / 1e-5 / 1e-6 / 1e-6

Why not a free-running comparison over 20 iterations: with beta1 = 0 the first Adam steps move every weight by
+-lr whatever the size of its gradient, so a gradient element whose fp32 value has the other sign than its fp64 twin
(a few hundred of 24 million per step) moves that weight the other way, and the two trajectories separate at the same rate
for ANY fp32 implementation (measured: eager HIP vs oracle 6e-3 after six iterations).  Re-synchronising before every op keeps
every op of all 20 iterations comparable at kernel-level tolerances, and any stale buffer / missed state update /
generator fault in a replay still shows up in the op it corrupts.

Graph replay vs eager execution on identical draws must be bit-identical (third test)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import rel_err, gloss_tape_in_reference_order

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = 32


def loop_kwargs(fmap, B, world=1, data_size=48, label=None, res=RES, **extra):
    from inclusivegan_amd.dnnlib import EasyDict
    label = label or dict(label_size=0)
    kw = dict(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=fmap, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=fmap, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=res, num_channels=3, **label),
        sched_args=EasyDict(minibatch_gpu_base=B, minibatch_size_base=B * world), tf_config={'rnd.np_random_seed': 1000},
        total_kimg=1, data_size=data_size, num_samples_factor=4, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=64)
    kw.update(extra)
    return kw


def record_loop(iterations, kwargs, consumer=None, reseed=None, tap=True, keep_state=True):
    """Run training_loop() for `iterations` iterations.  Per op a record {name, it, value, tape, fed inputs, post: state after the
    op (weights of G / D / Gs, the op's gradient bucket), pl_mean, dlatent_avg} goes to `consumer(rec)` (online checking: nothing
    is kept) or into log['ops'].  reseed=S: the device generator is re-seeded to S + i before iteration i."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import training_loop as TL
    src = tfutil.TapRandom() if tap else None
    log = dict(ops=[], init=None, graphs=None)
    cur = {}
    nets = {}
    npy = lambda t: t.detach().cpu().numpy().copy()

    def on_start(st):
        nets.update(st)
        G, D, lp = st['G'], st['D'], st['lpips']
        log['init'] = dict(
            G={n: npy(v) for n, v in G.vars.items()}, D={n: npy(v) for n, v in D.vars.items()}, lpips={n: npy(v) for n, v in lp.vars.items()},
            G_layout={n: (int(o), int(c), tuple(G.vars[n].shape)) for n, (o, c) in G._offsets.items()},
            D_layout={n: (int(o), int(c), tuple(D.vars[n].shape)) for n, (o, c) in D._offsets.items()})
        if consumer is not None and hasattr(consumer, 'start'):
            consumer.start(log['init'])
        if reseed is not None:
            torch.manual_seed(reseed)

    def on_batch(b):
        cur['batch'] = {k: np.array(v) for k, v in b.items() if isinstance(v, np.ndarray)}

    def on_op(name, out, feed):
        G, D, Gs = nets['G'], nets['D'], nets['Gs']
        rec = dict(name=name, value=float(out.detach().double().mean()), it=cur.get('it', 0))
        if src is not None:
            rec['tape'] = src.snapshot(name)
        if name in ('D', 'D_reg'):
            rec['reals'] = npy(feed['reals'])
        else:
            for k in ('reals_rec_1', 'latents_rec_1', 'reals_rec_2', 'latents_rec_2'):
                rec[k] = npy(feed[k])
            if name == 'G':
                rec['global_latents_rec_1'] = cur['batch']['latents_rec_1']      # the whole minibatch, before the rank slice
        rec['pl_mean'] = float(G.pl_mean_var) if hasattr(G, 'pl_mean_var') else 0.0
        rec['dlatent_avg'] = npy(G.vars['dlatent_avg'])
        net = G if name.startswith('G') else D
        if keep_state == 'digest':      # enough for bit-equality checks
            import hashlib
            rec['digest'] = [hashlib.sha1(npy(t).tobytes()).hexdigest() for t in (net.flat_grads, G.flat_params, D.flat_params, Gs.flat_params)]
        elif keep_state:
            rec['post'] = dict(wG=npy(G.flat_params), wD=npy(D.flat_params), Gs=npy(Gs.flat_params), g=npy(net.flat_grads))
        if consumer is not None:
            consumer(rec)
        else:
            log['ops'].append(rec)

    def on_it(info):
        cur['it'] = info['iteration']
        if reseed is not None:
            torch.manual_seed(reseed + info['iteration'])
        return info['iteration'] >= iterations

    hooks = dict(on_start=on_start, on_batch=on_batch, on_op=on_op, on_iteration=on_it, on_graphs=lambda g: log.update(graphs=g))
    if src is not None:
        hooks['random_source'] = src
    out = TL.training_loop(hooks=hooks, **kwargs)
    G, D, Gs = out['G'], out['D'], out['Gs']
    torch.cuda.synchronize()
    G_opt, D_opt = nets['G_opt'], nets['D_opt']
    log['final'] = dict(G=npy(G.flat_params), D=npy(D.flat_params), Gs=npy(Gs.flat_params), dlatent_avg=npy(G.vars['dlatent_avg']),
                        Gs_dlatent_avg=npy(Gs.vars['dlatent_avg']), pl_mean=float(G.pl_mean_var), G_pow=npy(G_opt._state['pow']), D_pow=npy(D_opt._state['pow']),
                        cur_nimg=out['cur_nimg'])
    return log


def flat_of(init, which):
    layout = init[which + '_layout']
    flat = np.zeros(max(o + c for o, c, _ in layout.values()), np.float32)
    for n, (o, c, _) in layout.items():
        flat[o:o + c] = init[which][n].reshape(-1)
    return flat


class TeacherForcedOracle:
    """Consumes the op records of a run (one list entry per rank) and checks each op against the oracle from the HIP path's own
    pre-op state.  `select(index, name, it)` chooses the ops that get the (expensive) oracle evaluation; the optimizer, moving
    average and state bookkeeping is checked on every op."""

    def __init__(self, fmap, B, world=1, select=None, dtype=torch.float64, res=RES):
        self.fmap, self.B, self.world, self.dtype, self.res = fmap, B, world, dtype, res
        self.select = select or (lambda j, name, it: True)
        self.ops = None
        self.j = 0
        self.evaluated = []
        self.worst = dict(value=0.0, grad=0.0, adam=0.0, Gs=0.0, pl_mean=0.0)

    def start(self, init):
        from oracle.train_ops import TrainOps
        cfg = dict(resolution=self.res, num_channels=3, fmap_base=self.fmap, G_arch='skip', D_arch='resnet')
        self.init = init
        self.ops = TrainOps(init['G'], init['D'], init['G_layout'], init['D_layout'], init['lpips'], cfg, world=self.world, minibatch_gpu=self.B, dtype=self.dtype)
        self.pre = dict(G=self.ops.w['G'].copy(), D=self.ops.w['D'].copy(), Gs=self.ops.w['G'].copy())
        self.state = [dict(pl_mean=0.0, dlatent_avg=init['G']['dlatent_avg'].copy()) for _ in range(self.world)]
        self.pending_Gs = False

    def __call__(self, rec):
        self.consume([rec])

    def consume(self, recs):
        import oracle.optimizer as OO
        ops = self.ops
        name, it = recs[0]['name'], recs[0]['it']
        assert all(r['name'] == name for r in recs)
        key = 'G' if name.startswith('G') else 'D'
        post = recs[0]['post']
        n = ops.w[key].size
        # --- Gs <- lerp(G, Gs, beta) ran after the previous D step (training_loop.py:477-478)
        if self.pending_Gs:
            exp = OO.ema(self.pre['Gs'], self.pre['G'], ops.Gs_beta)
            err = rel_err(post['Gs'][:exp.size], exp)
            self.worst['Gs'] = max(self.worst['Gs'], err)
            assert err < 1e-6, ('Gs', self.j, err)
            self.pre['Gs'] = post['Gs'][:exp.size].copy()
            self.pending_Gs = False
        else:
            assert np.array_equal(post['Gs'][:self.pre['Gs'].size], self.pre['Gs']), ('Gs changed outside its update', self.j)
        # --- the other network's weights did not move
        other = 'D' if key == 'G' else 'G'
        assert np.array_equal(post['w' + other][:self.pre[other].size], self.pre[other]), (name, self.j, 'touched ' + other)
        # --- oracle evaluation of the op from the HIP path's pre-op state
        if self.select(self.j, name, it):
            ops.w['G'], ops.w['D'] = self.pre['G'].copy(), self.pre['D'].copy()
            for r in range(self.world):
                ops.state[r]['pl_mean'] = torch.tensor(self.state[r]['pl_mean'], dtype=self.dtype)
                ops.state[r]['dlatent_avg'] = torch.from_numpy(self.state[r]['dlatent_avg'].astype(np.float64)).to(self.dtype)
            if key == 'G':
                towers = [dict(r, tape=gloss_tape_in_reference_order(r['tape'], self.B) if name == 'G' else r['tape']) for r in recs]
                vals, g = ops.G_op(towers, 'loss' if name == 'G' else 'reg', apply=False)
            else:
                vals, g = ops.D_op(recs, 'loss' if name == 'D' else 'reg', apply=False)
            for r, v in zip(recs, vals):
                e = abs(r['value'] - v) / (abs(v) + 1e-12)
                self.worst['value'] = max(self.worst['value'], e)
                assert e < (2e-4 if name in ('G', 'D') else 1e-3), (self.j, name, it, r['value'], v)
            scal_h, scal_o = [], []
            for vn, (o, c, _) in ops.layout[key].items():
                gh, go = post['g'][o:o + c].astype(np.float64), g[o:o + c].astype(np.float64)
                if c == 1:
                    scal_h.append(gh); scal_o.append(go)        # scalar parameters (noise_strength) judged jointly, as in test_gpu_networks
                    continue
                if not np.any(go):
                    assert not np.any(gh), (self.j, name, vn, 'gradient where the oracle has none')
                    continue
                e = float(np.linalg.norm(gh - go) / np.linalg.norm(go))
                self.worst['grad'] = max(self.worst['grad'], e)
                if os.environ.get('IGAN_TEST_GRAD_REPORT') == '1':      # DIAGNOSTIC: list the largest per-variable deviations instead of stopping at the first
                    self.report = sorted(getattr(self, 'report', []) + [(e, name, it, vn)], reverse=True)[:8]
                    print('GRAD-REPORT', self.report[:4], flush=True)
                    continue
                assert e < 5e-3, (self.j, name, it, vn, e, 'one leaky-ReLU unit on the other side of its kink than in fp64 reads 1e-3 ... 1e-2 in this metric under EVERY '
                                  'arithmetic form: tools/reg_forms.py --state loop:<it> + tools/kink_flips.py tell (profiles/r06_second_order.txt section 7)')
            if scal_o:
                e = float(np.linalg.norm(np.concatenate(scal_h) - np.concatenate(scal_o)) / (np.linalg.norm(np.concatenate(scal_o)) + 1e-30))
                assert e < 5e-3, (self.j, name, it, '<scalars>', e)
            for r, rec in enumerate(recs):
                if name == 'G_reg':
                    self.worst['pl_mean'] = max(self.worst['pl_mean'], abs(rec['pl_mean'] - float(ops.state[r]['pl_mean'])) / (abs(rec['pl_mean']) + 1e-30))
                    # 3e-4: the path-length VALUE itself sits 1.3e-4 from the fp64 oracle at 128x128 under every convolution form, the exact-fp32 instruction included
                    # (profiles/r04_f16_pairs_variant.txt section 9); config 5 at minibatch 3 read 4.9e-6 (bf16 x3) and 1.0e-4 (fp16 x2)
                    assert abs(rec['pl_mean'] - float(ops.state[r]['pl_mean'])) <= 1e-4 * abs(rec['pl_mean']) + 1e-8, (self.j, rec['pl_mean'], float(ops.state[r]['pl_mean']))
                if name != 'D_reg':
                    assert rel_err(rec['dlatent_avg'], ops.state[r]['dlatent_avg'].numpy()) < 1e-5, (self.j, name, 'dlatent_avg')
            self.evaluated.append((self.j, name, it))
        # --- the update: the oracle's SimpleAdam (its slots follow the HIP gradients, so they stay in step) on the HIP gradient
        w = self.pre[key].copy()
        applied = ops.adam[key].apply(w, post['g'][:n].copy())
        assert applied, (self.j, name, 'non-finite gradient')
        e = float(np.abs(w - post['w' + key][:n]).max() / (np.abs(w).max() + 1e-30))
        self.worst['adam'] = max(self.worst['adam'], e)
        assert e < 1e-6, (self.j, name, it, 'weights after the update', e)
        assert float(np.abs(post['w' + key][:n] - self.pre[key]).max()) > 0
        self.pre[key] = post['w' + key][:n].copy()
        for r, rec in enumerate(recs):
            self.state[r] = dict(pl_mean=rec['pl_mean'], dlatent_avg=rec['dlatent_avg'])
        if name == 'D':
            self.pending_Gs = True
        self.j += 1

    def finish(self, final):
        import oracle.optimizer as OO
        ops = self.ops
        exp_Gs = OO.ema(self.pre['Gs'], self.pre['G'], ops.Gs_beta) if self.pending_Gs else self.pre['Gs']
        assert rel_err(final['Gs'][:exp_Gs.size], exp_Gs) < 1e-6
        assert np.array_equal(final['G'][:self.pre['G'].size], self.pre['G']) and np.array_equal(final['D'][:self.pre['D'].size], self.pre['D'])
        assert np.array_equal(final['Gs_dlatent_avg'], final['dlatent_avg'])            # non-trainables are copied (beta_nontrainable = 0)
        # Adam beta powers (optimizer.py:311-317): one multiplication per applied update, shared between main and reg optimizer
        for key, adam in (('G_pow', ops.adam['G']), ('D_pow', ops.adam['D'])):
            np.testing.assert_allclose(final[key], [adam.b1pow, adam.b2pow], rtol=1e-6, atol=0)


def test_graphed_training_loop_matches_oracle_config2(cuda_device):
    """BASELINE config 2 / SURVEY 8d's parity row: Stacked-MNIST-shaped 32x32, config-e width (fmap_base 8192), minibatch_gpu 6,
    20 iterations of training_loop() with hipGraphs ON (G reg at iterations 1, 5, 9, 13, 17; D reg at 1, 17).  The optimizer /
    moving-average / state checks run on all 47 ops; the oracle losses and gradients (fp64, ~20 s per op at this width) on the
    ops of iterations 1 and 17 -- the two iterations that run all four op kinds, early and late (IGAN_TEST_TRAJECTORY_ALL=1: all 47)."""
    from inclusivegan_amd.dnnlib.tflib import graphs
    assert graphs.graphs_enabled(True), 'this test is about the captured path'
    n_it = int(os.environ.get('IGAN_TEST_TRAJECTORY_ITERS', '20'))
    chosen = {0, 16} if os.environ.get('IGAN_TEST_TRAJECTORY_ALL', '0') != '1' else set(range(n_it))
    if os.environ.get('IGAN_TEST_TRAJECTORY_ITS'):       # DIAGNOSTIC: the oracle on the ops of these iterations only (comma list, 0-based)
        chosen = {int(v) for v in os.environ['IGAN_TEST_TRAJECTORY_ITS'].split(',')}
    names, firsts = [], []

    class Consumer(TeacherForcedOracle):
        def __call__(self, rec):
            names.append(rec['name'])
            if rec['name'] == 'G':
                firsts.append(rec['tape'][1][1])
            super().__call__(rec)

    oracle = Consumer(8192, 6, select=lambda j, name, it: it in chosen)
    log = record_loop(n_it, loop_kwargs(8192, 6, data_size=48), consumer=oracle)
    oracle.finish(log['final'])
    g = log['graphs']
    assert g['captured'] and g['validated'] and g['faithful'] and all(c['faithful'] for c in g['checks']) and g['runtime']['packet_capture'] == '0', g
    assert names.count('G') == n_it and names.count('D') == n_it
    assert names.count('G_reg') == (n_it + 3) // 4 and names.count('D_reg') == (n_it + 15) // 16
    assert log['final']['cur_nimg'] == 12 * n_it
    assert {n for _, n, _ in oracle.evaluated} == {'G', 'G_reg', 'D', 'D_reg'}
    # the captured generator really advances from replay to replay: no two G steps saw the same draws, and they are N(0,1)
    assert all(not np.array_equal(firsts[0], f) for f in firsts[1:])
    allz = np.concatenate([f.reshape(-1) for f in firsts])
    assert abs(float(allz.mean())) < 0.05 and abs(float(allz.std()) - 1.0) < 0.05
    print('ops checked against the oracle: %d of %d; worst deviations %s' % (len(oracle.evaluated), len(names), oracle.worst))


def test_graphed_training_loop_every_op_small_width(cuda_device):
    """The same check with the oracle on EVERY op of 6 iterations (reduced width, fmap_base 512, 1000-d one-hot labels as in
    Stacked-MNIST: the labels reach G and D, which ignore them like the reference does)."""
    oracle = TeacherForcedOracle(512, 6)
    log = record_loop(6, loop_kwargs(512, 6, data_size=48, label=dict(label_size=1000, label_kind='onehot')), consumer=oracle)
    oracle.finish(log['final'])
    assert len(oracle.evaluated) == 6 + 2 + 6 + 1 and log['graphs']['faithful']
    print('worst deviations', oracle.worst)


def test_graph_replay_equals_eager_bitwise(cuda_device, monkeypatch):
    """The same 9 iterations (all four op kinds; G reg twice more) with IGAN_HIP_GRAPHS on and off, the device generator re-seeded
    before every iteration: every op's loss output, gradient bucket and the final G / D / Gs weights, pl_mean, dlatent_avg and
    beta powers are BIT-identical -- a stale static buffer, host-side state that a replay does not redo, or a generator offset
    that a replay does not advance would all show here."""
    runs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('IGAN_HIP_GRAPHS', mode)
        runs[mode] = record_loop(9, loop_kwargs(1024, 6, data_size=48, label=dict(label_size=10, label_kind='onehot')), reseed=4242, tap=False, keep_state='digest')
    a, b = runs['1'], runs['0']
    assert a['graphs'] is not None and a['graphs']['faithful'] and b['graphs'] is None
    assert [o['name'] for o in a['ops']] == [o['name'] for o in b['ops']]
    for x, y in zip(a['ops'], b['ops']):
        assert x['value'] == y['value'], (x['name'], x['it'], x['value'], y['value'])
        assert x['digest'] == y['digest'], (x['name'], x['it'])         # gradient bucket, G, D, Gs after the op
    for k in ('G', 'D', 'Gs', 'dlatent_avg', 'G_pow', 'D_pow'):
        assert np.array_equal(a['final'][k], b['final'][k]), k
    assert a['final']['pl_mean'] == b['final']['pl_mean']
    assert float(np.abs(a['final']['G'] - flat_of(a['init'], 'G')).max()) > 0


def test_two_rank_graphed_loop_matches_oracle_towers(cuda_device, tmp_path):
    """World size 2 (both ranks on GPU 0 over gloo -- RCCL refuses two ranks on one device): five iterations of the real loop,
    every rank recording its own draws and slices.  Replicas end bit-identical, the ranks' slices tile the global batch, and
    every op equals the oracle's two towers with averaged gradients (optimizer.py:186,199) under the one-rank tolerances."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), 'record', str(r), '2', str(port), str(tmp_path)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    logs = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(2)]
    for k in ('G', 'D', 'Gs', 'G_pow', 'D_pow'):
        assert np.array_equal(logs[0]['final'][k], logs[1]['final'][k]), k
    d0 = [o for o in logs[0]['ops'] if o['name'] == 'D']; d1 = [o for o in logs[1]['ops'] if o['name'] == 'D']
    assert not np.array_equal(d0[0]['reals'], d1[0]['reals'])               # different slices ...
    assert not np.array_equal(d0[0]['tape'][0][1], d1[0]['tape'][0][1])     # ... and different device draws per rank
    for a, b in zip([o for o in logs[0]['ops'] if o['name'] == 'G'], [o for o in logs[1]['ops'] if o['name'] == 'G']):
        assert np.array_equal(np.concatenate([a['latents_rec_1'], b['latents_rec_1']]), a['global_latents_rec_1'].astype(np.float32))
    oracle = TeacherForcedOracle(512, 3, world=2)
    oracle.start(logs[0]['init'])
    assert len(logs[0]['ops']) == len(logs[1]['ops'])
    for r0, r1 in zip(logs[0]['ops'], logs[1]['ops']):
        oracle.consume([r0, r1])
    oracle.finish(logs[0]['final'])
    print('worst deviations', oracle.worst)


def test_async_submission_equals_synchronous_loop(cuda_device):
    """The submission thread (training_loop.SubmitThread: the main thread assembles iteration i + 1 while a second thread hands iteration
    i's copies, graph replays and updates to the device) against the single-thread loop: same stream, same order, so after 9 iterations
    (refresh, lazy regularisers, three staging sets reused three times) G, D, Gs, pl_mean, dlatent_avg and the Adam powers are bit-identical."""
    import hashlib
    from inclusivegan_amd.training import training_loop as TL

    def final(flag):
        os.environ['IGAN_ASYNC_SUBMIT'] = flag
        seen = dict(n=0, threads=set())
        import threading

        def on_it(info):
            seen['n'] += 1
            seen['threads'] = {t.name for t in threading.enumerate()}
            return info['iteration'] >= 9
        try:
            out = TL.training_loop(hooks=dict(on_iteration=on_it, async_ok=True), **loop_kwargs(1024, 6, data_size=48))
        finally:
            os.environ.pop('IGAN_ASYNC_SUBMIT', None)
        torch.cuda.synchronize()
        G, D, Gs = out['G'], out['D'], out['Gs']
        dig = [hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest() for t in (G.flat_params, D.flat_params, Gs.flat_params, G.vars['dlatent_avg'], G.pl_mean_var)]
        return dig, seen
    a, sa = final('1')
    b, sb = final('0')
    assert 'igan-submit' in sa['threads'] and 'igan-submit' not in sb['threads']
    assert sa['n'] == sb['n'] == 9
    assert a == b


CONFIG5 = dict(res=128, fmap=8192, B=3, data_size=240, attr='Smiling', col=31)      # BASELINE config 5 at its own size (minibatch_gpu 3, attribute mask)


def config5_kwargs(world, **extra):
    from inclusivegan_amd.training import imle
    c = CONFIG5
    return loop_kwargs(c['fmap'], c['B'], world=world, data_size=c['data_size'], res=c['res'], label=dict(label_size=40, label_kind='attributes'),
                       attr_interesting=c['attr'], attr_names=list(imle.CELEBA_ATTRIBUTES), num_samples_factor=2, candidate_batch_size=16, **extra)


def test_config5_at_its_own_size_matches_oracle(cuda_device):
    """BASELINE config 5 on one rank at ITS OWN size: 128x128, config-e width (fmap_base 8192), minibatch_gpu 3,
    40 attribute labels with the attribute AND-mask (training_loop.py:416-424) choosing the reals, data_size divisible by
    2 * minibatch (:338-340), the four training ops captured as hipGraphs.  Two iterations of training_loop(); every op kind
    (G, G_reg, D, D_reg: iteration 1 runs all four) is evaluated by the fp64 oracle from the HIP path's pre-op state -- loss value,
    gradient of every trainable, Adam update, pl_mean, dlatent_avg, Gs -- and the fed reals all carry the attribute."""
    c = CONFIG5
    fed = []

    class Consumer(TeacherForcedOracle):
        def __call__(self, rec):
            super().__call__(rec)

    oracle = Consumer(c['fmap'], c['B'], select=lambda j, name, it: it == 0, res=c['res'])
    kw = config5_kwargs(1)
    assert kw['data_size'] % (2 * c['B']) == 0
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import training_loop as TL
    orig = TL.imle.ImleSampler.next_batch

    def next_batch(self, mb):
        b = orig(self, mb)
        fed.append({k: np.array(v) for k, v in b.items() if isinstance(v, np.ndarray)})
        return b
    TL.imle.ImleSampler.next_batch = next_batch
    try:
        log = record_loop(2, kw, consumer=oracle)
    finally:
        TL.imle.ImleSampler.next_batch = orig
    oracle.finish(log['final'])
    assert log['graphs']['captured'] and log['graphs']['faithful'], log['graphs']
    assert [e[1] for e in oracle.evaluated] == ['G', 'G_reg', 'D', 'D_reg'], oracle.evaluated
    assert len(fed) == 2
    for b in fed:
        assert b['reals_rec_1'].shape == (c['B'], 3, c['res'], c['res']) and b['labels_rec_1'].shape == (c['B'], 40)
        assert bool((b['labels_rec_1'][:, c['col']] == 1).all()) and bool((b['labels_rec_2'][:, c['col']] == 1).all())      # the AND-mask selection
    print('config 5 @128 world 1: worst deviations', oracle.worst)


def test_config5_two_ranks_at_its_own_size_match_oracle_towers(cuda_device, tmp_path):
    """Config 5's shape on TWO ranks (GPU 0, gloo) at 128x128 / fmap_base 8192 / minibatch_gpu 3 / attribute mask: two iterations,
    replicas bit-identical, the ranks' slices tile the masked global minibatch, and the G op of iteration 1 equals the oracle's two
    towers with 1 / 2-scaled summed gradients (optimizer.py:186,199; the D, G_reg and D_reg ops at this size are the one-rank test's, the
    two-tower D op at 32x32 test_two_rank_graphed_loop_matches_oracle_towers's); the optimizer / moving-average bookkeeping is checked on
    every op of both iterations.  The RCCL leg itself needs more than one device."""
    c = CONFIG5
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), 'record5', str(r), '2', str(port), str(tmp_path)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=2400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    logs = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(2)]
    for k in ('G', 'D', 'Gs', 'G_pow', 'D_pow'):
        assert np.array_equal(logs[0]['final'][k], logs[1]['final'][k]), k
    for a, b in zip([o for o in logs[0]['ops'] if o['name'] == 'G'], [o for o in logs[1]['ops'] if o['name'] == 'G']):
        assert np.array_equal(np.concatenate([a['latents_rec_1'], b['latents_rec_1']]), a['global_latents_rec_1'].astype(np.float32))
    oracle = TeacherForcedOracle(c['fmap'], c['B'], world=2, select=lambda j, name, it: it == 0 and name == 'G', res=c['res'])
    oracle.start(logs[0]['init'])
    for r0, r1 in zip(logs[0]['ops'], logs[1]['ops']):
        oracle.consume([r0, r1])
    oracle.finish(logs[0]['final'])
    assert [e[1] for e in oracle.evaluated] == ['G']
    print('config 5 @128 world 2: worst deviations', oracle.worst)


def test_packet_capture_fault_is_detected_by_the_replay_check(cuda_device):
    """The HIP runtime's graph packet capture (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default in the runtime) makes replays of the
    G regulariser's graph disagree with its eager execution on this machine; inclusivegan_amd/__init__.py switches it off.  With
    it forced back ON the loop's own replay check must notice and run the op eagerly (so results stay right either way).  If a
    future runtime fixes the fault this test reports that instead of failing: the check then simply passes."""
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "import tests.test_gpu_loop_parity as T\n"
        "from inclusivegan_amd.training import training_loop as TL\n"
        "seen = []\n"
        "TL.training_loop(hooks=dict(on_graphs=seen.append, on_iteration=lambda i: True), **T.loop_kwargs(1024, 6, data_size=48))\n"
        "print('GRAPHS', seen[0])\n" % ROOT)
    out = {}
    for flag in ('0', '1'):
        env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE=flag)
        r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith('GRAPHS')][0]
        out[flag] = (eval(line[len('GRAPHS '):]), 'does not reproduce its eager execution' in r.stdout)
    assert out['0'][0]['faithful'] and not out['0'][1]                  # the product's setting: every graph validated
    assert out['1'][0]['faithful'] != out['1'][1]                       # forced on: either the fault shows AND is reported, or it is gone
    print('packet capture forced on: faithful =', out['1'][0]['faithful'])
