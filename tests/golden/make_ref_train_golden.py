"""Golden vectors for the TRAINING-side statements of the hot path, produced by EXECUTING the reference's own Python.

Round 3 pinned the op / layer / network arithmetic this way (make_ref_ops_golden.py).  This script does the same for the last
functions that were restated on both sides of every test:

    training/loss.py                    G_logistic_ns_rec_interp_arb_pathreg :19-91 and D_logistic_r1 :93-113 -- imported from
                                        /root/reference and called unchanged: which image pairs reach LPIPS, the 0.5 / 0.4 factors, the
                                        slerp / lerp argument order, softplus signs, the path-length statistics (pl_lengths, pl_mean
                                        update, penalty), the R1 reduction and gamma / 2
    training/training_loop.py           process_reals :40-60 (cut out of the file's syntax tree: the module imports sklearn / DCI at the
                                        top) incl. the mirror branch and the level-of-detail fade / upscale; the optimizer set-up :242-255
                                        (lazy-regularisation learning rate and beta exponents), Gs_beta :222, and the registration block
                                        :283-291 (reduce_mean(reg * interval); non-lazy: loss += reg), executed from the syntax tree
    dnnlib/tflib/optimizer.py           Optimizer.register_gradients / apply_updates :114-265 (1 / num_devices scaling :186, all_sum over
                                        the devices :193-201, the gradient-accumulation branch with multiplier 1 :208-233, the finite
                                        gate :236-239) driving SimpleAdam.apply_gradients :303-336, several steps on two devices
    dnnlib/tflib/network.py             Network.setup_as_moving_average_of :341-351

What is a stand-in (everything else runs from the reference's files): TensorFlow's primitives (tests/golden/np_tf.py, float64);
`tf.gradients` (cannot be executed: the hook returns the fp64 autograd gradient of oracle/networks_stylegan2.py, whose forward the
round-3 goldens pin, after checking that the oracle's forward equals the reference's at that point); G / D objects (a scope plus a call
of the reference's build function, as in make_ref_ops_golden.py); LPIPS (oracle/lpips.py on seeded weights -- the pickle is absent);
nccl_ops.all_sum (the sum); autosummary (records the value it is given); random draws (seeded, recorded).

Output: tests/golden/ref_train_golden.npz.  tests/test_ref_train_golden.py (CPU) requires oracle/ to reproduce it;
tests/test_gpu_ref_golden.py runs the HIP path against it.

Run from the repo root:  python tests/golden/make_ref_train_golden.py   (needs /root/reference)
"""
import ast
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from tests.golden import np_tf  # noqa: E402
from tests.golden import make_ref_ops_golden as OPS  # noqa: E402
from tests.util import lpips_params_from_seed  # noqa: E402

tf = np_tf
RES, FMAP, B = 16, 64, 6
SMALL = dict(latent_size=32, dlatent_size=48, mapping_fmaps=40)
LPIPS_SEED = 77


def T64(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


def put(out, prefix, d):
    for k, v in d.items():
        a = np.asarray(v)
        out[prefix + k.replace('/', '.')] = a.astype(np.float32) if a.dtype == np.float64 and np.array_equal(a.astype(np.float32).astype(np.float64), a) else a


def cut_function(path, name):
    mod = ast.parse(open(path).read(), filename=path)
    fn = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    return compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec')


# ---------------------------------------------------------------------------------------------------------------
# losses

class RefNet:
    """G / D as the loss functions see them: get_output_for = the reference's build function under the network's scope."""

    def __init__(self, N, name, func, static, input_shapes, output_shape):
        self.N, self.name, self.func, self.static = N, name, func, static
        self.input_shapes, self.output_shape = input_shapes, output_shape

    def get_output_for(self, *inputs, **dyn):
        kw = dict(self.static)
        kw.update(dyn)
        if self.name == 'G':
            kw['components'] = self.N.dnnlib.EasyDict()
        with tf.variable_scope(tf.VariableScope([self.name])):
            return self.func(*inputs, **kw)


class RefLpips:
    def __init__(self, params):
        self.params = {k: T64(v) for k, v in params.items()}
        self.calls = []

    def get_output_for(self, a, b):
        from oracle import lpips as OLP
        self.calls.append((np.array(a.v), np.array(b.v)))
        return tf.Tensor(OLP.lpips(self.params, T64(a.v), T64(b.v)).numpy())


class RefTrainingSet:
    @staticmethod
    def get_random_labels_tf(n):
        return tf.Tensor(np.zeros((int(np_tf._val(n)), 0)))


def loss_goldens(out, N, L):
    from oracle import networks_stylegan2 as ON
    from oracle.misc import Tape
    from inclusivegan_amd.dnnlib import tflib as P

    def net_params(kind, arch, seed):
        fn = 'inclusivegan_amd.training.networks_stylegan2.' + ('G_main' if kind == 'G' else 'D_stylegan2_feature')
        net = P.Network(kind, func_name=fn, num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, architecture=arch, device='cpu', seed=seed, **(SMALL if kind == 'G' else {}))
        prng = np.random.RandomState(seed)
        vals = {}
        for name, v in net.vars.items():
            a = v.detach().numpy().astype(np.float64)
            if name.endswith('bias') or name.endswith('noise_strength') or name == 'dlatent_avg':
                a = np.asarray(prng.randn(*a.shape) * 0.2).astype(np.float32).astype(np.float64)
            vals[name] = a
        return vals

    gp, dp = net_params('G', 'skip', 301), net_params('D', 'resnet', 302)
    put(out, 'loss_Gparam.', gp)
    put(out, 'loss_Dparam.', dp)
    lpips_params = lpips_params_from_seed(LPIPS_SEED)
    rng = np.random.RandomState(20261004)
    # the networks in one injected variable dict: 'G/...' and 'D/...'
    params = {'G/' + k: v for k, v in gp.items()}
    params.update({'D/' + k: v for k, v in dp.items()})
    common = dict(resolution=RES, fmap_base=FMAP, num_channels=3, label_size=0)
    G = RefNet(N, 'G', N.G_main, dict(architecture='skip', **common, **SMALL), [[None, SMALL['latent_size']], [None, 0]], [None, 3, RES, RES])
    D = RefNet(N, 'D', N.D_stylegan2_feature, dict(architecture='resnet', **common), [[None, 3, RES, RES], [None, 0]], [None])

    class RefNetwork:        # tflib.Network inside G_main (components.synthesis / .mapping), as in make_ref_ops_golden.py
        def __init__(self, name, func_name=None, **static):
            self.name, self.func, self.static, self.vars = name, func_name, static, {}
            res_log2 = int(np.log2(static.get('resolution', 1024)))
            self.input_shape = [None, res_log2 * 2 - 2, static.get('dlatent_size', 512)]

        def get_output_for(self, *inputs, **dyn):
            kw = dict(self.static); kw.update(dyn)
            with tf.variable_scope(tf.VariableScope(['G', self.name])):
                return self.func(*inputs, **kw)
    N.tflib.Network = RefNetwork

    summaries = {}
    L.autosummary = lambda name, value, **kw: summaries.setdefault(name, []).append(np.array(value.v)) or value

    okw = dict(fmap_base=FMAP, architecture='skip', is_training=True, dlatent_size=SMALL['dlatent_size'], mapping_fmaps=SMALL['mapping_fmaps'])
    gpt = {k: T64(v) for k, v in gp.items()}
    dpt = {k: T64(v) for k, v in dp.items()}
    reals1 = rng.uniform(-1, 1, (B, 3, RES, RES)).astype(np.float32).astype(np.float64)
    reals2 = rng.uniform(-1, 1, (B, 3, RES, RES)).astype(np.float32).astype(np.float64)
    z1, z2 = rng.randn(B, SMALL['latent_size']), rng.randn(B, SMALL['latent_size'])
    z1 = (z1 / np.linalg.norm(z1, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    z2 = (z2 / np.linalg.norm(z2, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    lab = np.zeros((B, 0))
    out.update(loss_reals_rec_1=reals1, loss_reals_rec_2=reals2, loss_latents_rec_1=z1, loss_latents_rec_2=z2)
    out['loss_cfg'] = np.array([RES, FMAP, B, SMALL['latent_size'], SMALL['dlatent_size'], SMALL['mapping_fmaps'], LPIPS_SEED])
    pl_mean_before = 0.37

    for w in (2.5, 0.0):
        rec = OPS.Recorder(500 + int(w * 10))
        marks = {}

        def grad_hook(ys, xs):
            """tf.gradients(sum(images * noise), [dlatents]) (loss.py:65): oracle autograd at the same point."""
            dl = xs[0]
            assert len(xs) == 1 and dl.v.ndim == 3
            entries = rec.entries[marks['pl_start']:]
            zpl, noise = entries[0][1], entries[-1][1]
            tape = entries[1:-1]
            coin = [i for i, (k, v) in enumerate(tape) if k == 'uniform' and np.ndim(v) == 0][0]
            if tape[coin + 1][0] != 'randint':
                tape = tape[:coin + 1] + [('randint', np.int64(1))] + tape[coin + 1:]
            # latents -> dlatents -> images through the oracle's generator on the same draws (the latents carry the graph)
            img, dl_o = ON.G_main(gpt, T64(zpl).requires_grad_(True), Tape(tape, torch.float64), RES, return_dlatents=True, state={}, **okw)
            assert float((dl_o.detach() - T64(dl.v)).abs().max()) < 1e-10
            pn = T64(noise) / np.sqrt(RES * RES)
            s = torch.sum(img * pn)
            assert abs(float(s.detach()) - float(ys.v)) < 1e-9 * max(1.0, abs(float(ys.v))), (float(s.detach()), float(ys.v))
            marks['pl_grads'] = torch.autograd.grad(s, [dl_o])[0].numpy()
            return [marks['pl_grads']]

        store = [tf.Variable(pl_mean_before, dtype=tf.float32, name='pl_mean', trainable=False)]
        lp = RefLpips(lpips_params)
        summaries.clear()
        orig_normal = rec.normal

        def normal(shape, _orig=orig_normal):      # mark the start of the path-length section: its latents are [B // 2, latent]
            if list(shape) == [B // 2, SMALL['latent_size']] and 'pl_start' not in marks:
                marks['pl_start'] = len(rec.entries)
            return _orig(shape)
        rec.normal = normal
        with tf.session(params, rec, grad_hook=grad_hook) as st, tf.variable_replay(store):
            loss, reg = L.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, RefTrainingSet, B, tf.Tensor(reals1), tf.Tensor(lab), tf.Tensor(z1),
                                                               tf.Tensor(reals2), tf.Tensor(lab), tf.Tensor(z2), NN_rec_lpips_weight=w)
        entries = list(rec.entries)
        main, plsec = entries[:marks['pl_start']], entries[marks['pl_start']:]

        def fix(tape):       # the cutoff draw only happens when the coin says "mix"; restatements draw it always
            tape = list(tape)
            i = 0
            while i < len(tape):
                if tape[i][0] == 'uniform' and np.ndim(tape[i][1]) == 0 and (i + 1 >= len(tape) or tape[i + 1][0] != 'randint'):
                    tape.insert(i + 1, ('randint', np.int64(1)))
                i += 1
            return tape
        p = 'Gloss_w%d_' % int(w * 10)
        OPS.pack_tape(out, p + 'main_', fix(main))
        OPS.pack_tape(out, p + 'pl_', fix(plsec))
        out.update({p + 'loss': loss.v, p + 'reg': reg.v, p + 'pl_mean_before': np.float64(pl_mean_before), p + 'pl_mean_after': np.array(store[0].v),
                    p + 'pl_grads': marks['pl_grads'], p + 'weight': np.float64(w)})
        for name, vals in summaries.items():
            assert len(vals) == 1
            out[p + 'term_' + name.split('/')[1]] = vals[0]
        assert len(lp.calls) == 4
        # which images were compared with which reals (loss.py:31,41): recorded as the index of the real set
        which = []
        for a, b in lp.calls:
            which.append(1 if np.array_equal(b, (reals1 + 1) * (255 / 2)) else (2 if np.array_equal(b, (reals2 + 1) * (255 / 2)) else 0))
        out[p + 'lpips_real_sets'] = np.array(which)

    # ---- D loss + R1
    reals = rng.uniform(-1, 1, (2 * B, 3, RES, RES)).astype(np.float32).astype(np.float64)
    rec = OPS.Recorder(611)
    marks = {}

    def grad_hook_d(ys, xs):
        """tf.gradients(sum(real_scores), [reals]) (loss.py:108)."""
        assert len(xs) == 1 and xs[0].v.ndim == 4
        x = T64(xs[0].v).requires_grad_(True)
        s, _ = ON.D_stylegan2_feature(dpt, x, RES, fmap_base=FMAP, architecture='resnet')
        assert abs(float(s.sum().detach()) - float(ys.v)) < 1e-9 * max(1.0, abs(float(ys.v)))
        marks['real_grads'] = torch.autograd.grad(s.sum(), [x])[0].numpy()
        return [marks['real_grads']]
    summaries.clear()
    with tf.session(params, rec, grad_hook=grad_hook_d):
        loss, reg = L.D_logistic_r1(G, D, RefTrainingSet, B, tf.Tensor(reals), tf.Tensor(np.zeros((2 * B, 0))), gamma=100.0)
    tape = list(rec.entries)
    coin = [i for i, (k, v) in enumerate(tape) if k == 'uniform' and np.ndim(v) == 0][0]
    if tape[coin + 1][0] != 'randint':
        tape.insert(coin + 1, ('randint', np.int64(1)))
    OPS.pack_tape(out, 'Dloss_', tape)
    out.update(Dloss_reals=reals, Dloss_loss=loss.v, Dloss_reg=reg.v, Dloss_real_grads=marks['real_grads'], Dloss_gamma=np.float64(100.0))
    for name, vals in summaries.items():
        out['Dloss_term_' + name.split('/')[1]] = vals[0]


# ---------------------------------------------------------------------------------------------------------------
# process_reals

def process_reals_goldens(out, T):
    from training import misc as ref_misc
    code = cut_function(os.path.join(REF, 'training', 'training_loop.py'), 'process_reals')
    ns = dict(tf=tf, misc=ref_misc, tflib=T, np=np)
    exec(code, ns)
    process_reals = ns['process_reals']
    rng = np.random.RandomState(31)
    cases = []
    for j, (n, hw, lod, mirror, drange_data) in enumerate([(6, 8, 0.0, False, [0, 255]), (6, 8, 0.0, True, [0, 255]), (5, 8, 0.0, True, [0, 1]),
                                                            (4, 8, 1.25, True, [0, 255]), (4, 8, 2.0, False, [0, 255]), (3, 16, 0.5, False, [0, 255])]):
        x = rng.randint(0, 256, size=(n, 3, hw, hw)).astype(np.uint8)
        if drange_data == [0, 1]:
            x = (x / 255.0).astype(np.float32)
        lab = rng.rand(n, 3).astype(np.float32)
        rec = OPS.Recorder(40 + j)
        with tf.session({}, rec):
            y, lab_out = process_reals(tf.Tensor(x, tf.uint8 if x.dtype == np.uint8 else tf.float32), tf.Tensor(lab), tf.Tensor(np.float64(lod)), mirror, drange_data, [-1, 1])
        assert np.array_equal(lab_out.v, lab.astype(np.float64))
        p = 'preals_%d_' % j
        coins = [v for k, v in rec.entries]
        assert len(coins) == (1 if mirror else 0)
        out.update({p + 'x': x, p + 'y': y.v, p + 'cfg': np.array([lod, float(mirror), drange_data[0], drange_data[1]], np.float64),
                    p + 'coin': coins[0] if coins else np.zeros(0)})
        cases.append(p)
    out['preals_cases'] = np.array(cases)


# ---------------------------------------------------------------------------------------------------------------
# Optimizer + SimpleAdam, EMA

def optimizer_goldens(out, O):
    O.autosummary.autosummary = lambda name, value, **kw: value
    O.tfutil.run = lambda *a, **k: None
    O.tfutil.init_uninitialized_vars = lambda *a, **k: None

    class Nccl:
        @staticmethod
        def all_sum(tensors):
            total = sum(np.asarray(t.v) for t in tensors)
            return [tf.Tensor(total.copy(), t.dtype) for t in tensors]
    O.nccl_ops = Nccl

    rng = np.random.RandomState(99)
    shapes = [(3, 3, 4, 5), (5,), (7, 2), ()]
    cases = []
    for name, devices, multiplier, steps, nan_step, hp in [
            ('one_device', 1, None, 4, None, dict(learning_rate=0.002, beta1=0.0, beta2=0.99, epsilon=1e-8)),
            ('two_devices', 2, None, 4, 2, dict(learning_rate=0.0016, beta1=0.0, beta2=0.99 ** 0.8, epsilon=1e-8)),
            ('two_devices_multiplier_1', 2, 1, 3, None, dict(learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8))]:
        w0 = [np.asarray(rng.randn(*s)).astype(np.float32).astype(np.float64) for s in shapes]
        dev_names = ['/gpu:%d' % d for d in range(devices)]
        dev_vars = []
        for d in dev_names:
            with tf.device(d):
                dev_vars.append([tf.Variable(w.copy(), dtype=tf.float32, name='w%d' % i) for i, w in enumerate(w0)])
        store = []
        grads_all, weights_all = [], []
        for step in range(steps):
            grads = [[np.asarray(rng.randn(*s) * (10.0 ** rng.randint(-3, 2))).astype(np.float32).astype(np.float64) for s in shapes] for _ in dev_names]
            if nan_step is not None and step == nan_step:
                grads[1][2][3, 1] = np.inf
            table = {}

            def hook(ys, xs, _t=table):       # compute_gradients(loss, var_list) -> the gradients injected for that device's variables
                return _t[id(xs[0])]
            np_tf._GRAPH.names.clear()
            with tf.session({}, grad_hook=hook), tf.variable_replay(store):
                opt = O.Optimizer(name='TrainG', tf_optimizer='dnnlib.tflib.optimizer.SimpleAdam', minibatch_multiplier=None if multiplier is None else tf.Tensor(np.int64(multiplier)), **hp)
                for d, vs, gs in zip(dev_names, dev_vars, grads):
                    with tf.device(d):
                        table[id(vs[0])] = gs
                        opt.register_gradients(tf.Tensor(np.float64(1.0)), vs)
                opt.apply_updates()
            for d in range(1, devices):          # every device applied the same summed gradient to its copy
                for a, b in zip(dev_vars[0], dev_vars[d]):
                    assert np.array_equal(a.v, b.v)
            grads_all.append(grads)
            weights_all.append([v.v.copy() for v in dev_vars[0]])
        p = 'opt_%s_' % name
        out[p + 'hp'] = np.array([hp['learning_rate'], hp['beta1'], hp['beta2'], hp['epsilon'], devices, -1 if multiplier is None else multiplier, steps], np.float64)
        for i, w in enumerate(w0):
            out['%sw0_%d' % (p, i)] = w
        for s in range(steps):
            for d in range(devices):
                for i in range(len(shapes)):
                    out['%sgrad_s%d_d%d_%d' % (p, s, d, i)] = grads_all[s][d][i]
            for i in range(len(shapes)):
                out['%sw_s%d_%d' % (p, s, i)] = weights_all[s][i]
        cases.append(p)
    out['opt_cases'] = np.array(cases)
    out['opt_num_vars'] = np.array(len(shapes))


def ema_goldens(out, NW, T):
    rng = np.random.RandomState(5)
    names = ['a/weight', 'a/bias', 'dlatent_avg', 'lod', 'only_in_dst']
    shapes = [(4, 3), (3,), (6,), (), (2,)]
    trainable = {'a/weight', 'a/bias'}
    src = {n: rng.randn(*s) for n, s in zip(names[:4], shapes[:4])}
    dst = {n: rng.randn(*s) for n, s in zip(names, shapes)}
    cases = []
    for j, (beta, beta_nt) in enumerate([(0.99, 0.0), (0.5 ** (12 / 10000.0), 0.0), (0.9, 0.25)]):
        class Net:
            scope = 'Gs'
        me, other = Net(), Net()
        with tf.session({}):
            me.vars = {n: tf.Variable(np.array(v), dtype=tf.float32, name=n) for n, v in dst.items()}
            other.vars = {n: tf.Variable(np.array(v), dtype=tf.float32, name=n) for n, v in src.items()}
            me.trainables = {n: v for n, v in me.vars.items() if n in trainable}
            NW.Network.setup_as_moving_average_of(me, other, beta=beta, beta_nontrainable=beta_nt)
        p = 'ema_%d_' % j
        out[p + 'betas'] = np.array([beta, beta_nt])
        for n in names:
            out[p + 'after.' + n.replace('/', '.')] = me.vars[n].v
        cases.append(p)
    for n in names:
        out['ema_dst.' + n.replace('/', '.')] = dst[n]
        if n in src:
            out['ema_src.' + n.replace('/', '.')] = src[n]
    out['ema_trainable'] = np.array(sorted(trainable))
    out['ema_cases'] = np.array(cases)


# ---------------------------------------------------------------------------------------------------------------
# training_loop.py: optimizer set-up, Gs_beta, registration block -- statements executed from the syntax tree

def loop_setup_goldens(out):
    path = os.path.join(REF, 'training', 'training_loop.py')
    mod = ast.parse(open(path).read(), filename=path)
    tl = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name == 'training_loop'][0]

    def targets(n):
        return [getattr(t, 'id', None) for t in getattr(n, 'targets', [])]
    body = tl.body
    i0 = [i for i, n in enumerate(body) if isinstance(n, ast.Assign) and 'G_opt_args' in targets(n)][0]
    i1 = [i for i, n in enumerate(body) if isinstance(n, ast.Assign) and 'D_reg_opt' in targets(n)][0]
    setup = compile(ast.fix_missing_locations(ast.Module(body=body[i0:i1 + 1], type_ignores=[])), path, 'exec')
    inputs = [n for n in body if isinstance(n, ast.With) and any(isinstance(s, ast.Assign) and 'Gs_beta' in targets(s) for s in n.body)][0]
    gs_beta = [s for s in inputs.body if isinstance(s, ast.Assign) and 'Gs_beta' in targets(s)][0]
    gs_code = compile(ast.fix_missing_locations(ast.Module(body=[gs_beta], type_ignores=[])), path, 'exec')
    gpu_loop = [n for n in body if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'gpu'][0]
    with_gpu = gpu_loop.body[0]
    i2 = [i for i, n in enumerate(with_gpu.body) if isinstance(n, ast.If) and isinstance(n.test, ast.UnaryOp)][0]      # if not lazy_regularization:
    reg_code = compile(ast.fix_missing_locations(ast.Module(body=with_gpu.body[i2:i2 + 3], type_ignores=[])), path, 'exec')

    class RecOpt:
        made = []

        def __init__(self, name, share=None, **kw):
            self.name, self.share, self.kw, self.registered = name, share, kw, []
            RecOpt.made.append(self)

        def register_gradients(self, loss, trainables):
            self.registered.append((np.array(loss.v), trainables))
    rng = np.random.RandomState(17)
    cases = []
    for j, (lazy, gi, di, lrate) in enumerate([(True, 4, 16, 0.002), (False, 4, 16, 0.002), (True, 8, 2, 0.0025)]):
        RecOpt.made = []
        ns = dict(tflib=type('T', (), dict(Optimizer=RecOpt)), tf=tf, G_opt_args=dict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=dict(beta1=0.0, beta2=0.99, epsilon=1e-8),
                  G_reg_interval=gi, D_reg_interval=di, lazy_regularization=lazy, lrate_in=tf.Tensor(np.float64(lrate)), minibatch_multiplier=None)
        exec(setup, ns)
        p = 'setup_%d_' % j
        out[p + 'cfg'] = np.array([float(lazy), gi, di, lrate])
        for o in RecOpt.made:
            out[p + o.name] = np.array([float(np_tf._val(o.kw['learning_rate'])), o.kw['beta1'], o.kw['beta2'], o.kw['epsilon'], float(o.share is not None)])
        assert ns['G_reg_opt'].share is ns['G_opt'] and ns['D_reg_opt'].share is ns['D_opt']
        # registration block with tensors standing for the four loss outputs (equal lengths so that the non-lazy sum is defined)
        vals = dict(G_loss=rng.randn(6), G_reg=rng.randn(6), D_loss=rng.randn(6), D_reg=rng.randn(6))
        ns.update({k: tf.Tensor(v) for k, v in vals.items()})
        ns.update(G_gpu=type('N', (), dict(trainables='G.trainables')), D_gpu=type('N', (), dict(trainables='D.trainables')))
        exec(reg_code, ns)
        for k, v in vals.items():
            out[p + 'in_' + k] = v
        for o in RecOpt.made:
            out[p + o.name + '_registered'] = np.array([r[0] for r in o.registered])
            assert all(r[1] == o.name[-1] + '.trainables' for r in o.registered)
        cases.append(p)
    out['setup_cases'] = np.array(cases)
    betas = []
    for mb, kimg in [(12, 10.0), (32, 10.0), (48, 5.0), (12, 0.0)]:
        ns = dict(tf=tf, minibatch_size_in=tf.Tensor(np.int64(mb), tf.int32), G_smoothing_kimg=kimg)
        exec(gs_code, ns)
        betas.append([mb, kimg, float(np_tf._val(ns['Gs_beta']))])
    out['gs_beta'] = np.array(betas)


def main():
    U, F, N, T, calls = OPS.load_reference()
    L = importlib.import_module('training.loss')
    O = importlib.import_module('dnnlib.tflib.optimizer')
    NW = importlib.import_module('dnnlib.tflib.network')
    assert L.__file__.startswith(REF) and O.__file__.startswith(REF) and NW.__file__.startswith(REF)
    out = {}
    loss_goldens(out, N, L)
    process_reals_goldens(out, T)
    optimizer_goldens(out, O)
    ema_goldens(out, NW, T)
    loop_setup_goldens(out)
    path = os.path.join(HERE, 'ref_train_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.2f MB' % (path, len(out), os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    main()
