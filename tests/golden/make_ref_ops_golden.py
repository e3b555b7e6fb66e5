"""Golden vectors for the op / layer / network arithmetic, produced by EXECUTING the reference's own Python.

The reference's `dnnlib/tflib/ops/upfirdn_2d.py`, `dnnlib/tflib/ops/fused_bias_act.py` and `training/networks_stylegan2.py` are
imported from /root/reference where they lie (read at generation time only; nothing of them is stored in the repo) with
`tests/golden/np_tf.py` standing in for TensorFlow: a NumPy restatement of the ~40 TF primitives those files call (its header
lists the semantics assumed for each).  So every statement of the reference that COMPOSES primitives runs unchanged:

    upfirdn_2d.py       _setup_kernel; _upfirdn_2d_ref (zero insertion, pad / crop, flipped-kernel VALID convolution, decimation);
                        _upfirdn_2d_cuda's gradient definition (flipped kernel, up <-> down, gpad*), evaluated through the same
                        reference implementation (the file declares 'ref' and 'cuda' to be one function) for 1st and 2nd order;
                        upsample_2d / downsample_2d / filter_2d / upsample_conv_2d / conv_downsample_2d (kernel gains, pad
                        arithmetic, weight flip + transposed convolution)
    fused_bias_act.py   the activation table (def_alpha, def_gain, cuda_idx, ref, zero_2nd_grad) and _fused_bias_act_ref
    networks_stylegan2  get_weight / dense_layer / conv2d_layer / apply_bias_act / modulated_conv2d_layer (fused AND non-fused,
                        plain / up / down, with and without demodulation) / minibatch_stddev_layer / G_mapping /
                        G_synthesis_stylegan2 / D_stylegan2_feature / G_main (training: dlatent_avg update + style mixing;
                        validation: truncation), architectures orig / skip / resnet

Stand-ins besides the TF primitives: the two CUDA plugin ops (forward calls are answered by the reference's own '*_ref'
implementation of the same op), `tflib.Network` inside G_main (a scope + a call of the build function), random draws (recorded and
stored, TF's generator cannot be reproduced), variable values (seeded NumPy, stored).

Output: tests/golden/ref_ops_golden.npz (inputs, parameters, random draws and the reference's outputs).  tests/test_ref_ops_golden.py
requires oracle/ AND the product's host helpers to reproduce them; tests/test_gpu_ref_golden.py runs the HIP path against them.

Run from the repo root:  python tests/golden/make_ref_ops_golden.py   (needs /root/reference)
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from tests.golden import np_tf  # noqa: E402


def load_reference():
    np_tf.install()
    sys.path.insert(0, REF)
    U = importlib.import_module('dnnlib.tflib.ops.upfirdn_2d')
    F = importlib.import_module('dnnlib.tflib.ops.fused_bias_act')
    N = importlib.import_module('training.networks_stylegan2')
    T = importlib.import_module('dnnlib.tflib.tfutil')
    assert U.__file__.startswith(REF) and N.__file__.startswith(REF)

    calls = dict(upfirdn=[], simple=[])

    class UpfirdnPlugin:        # the UpFirDn2D op (upfirdn_2d.cu), answered by the reference's own reference implementation
        @staticmethod
        def up_fir_dn2d(x, k, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
            calls['upfirdn'].append(dict(k=np.array(np_tf._val(k)), upx=upx, upy=upy, downx=downx, downy=downy, padx0=padx0, padx1=padx1, pady0=pady0, pady1=pady1))
            return U._upfirdn_2d_ref(x=x, k=np_tf._val(k), upx=upx, upy=upy, downx=downx, downy=downy, padx0=padx0, padx1=padx1, pady0=pady0, pady1=pady1)
    U._get_plugin = lambda: UpfirdnPlugin

    by_idx = {spec.cuda_idx: name for name, spec in F.activation_funcs.items()}

    class BiasActPlugin:        # the FusedBiasAct op (fused_bias_act.cu): forward only (grad=0), answered by _fused_bias_act_ref
        @staticmethod
        def fused_bias_act(x, b, ref, grad, axis, act, alpha, gain):
            assert grad == 0, 'the derivative tables live in the .cu file and are not executable here'
            return F._fused_bias_act_ref(x=x, b=b, axis=axis, act=by_idx[act], alpha=alpha, gain=gain)
    F._get_plugin = lambda: BiasActPlugin

    simple = U._simple_upfirdn_2d
    def recording_simple(x, k, up=1, down=1, pad0=0, pad1=0, data_format='NCHW', impl='cuda'):
        calls['simple'].append(dict(k=np.array(k), up=up, down=down, pad0=pad0, pad1=pad1))
        return simple(x, k, up=up, down=down, pad0=pad0, pad1=pad1, data_format=data_format, impl=impl)
    U._simple_upfirdn_2d = recording_simple
    return U, F, N, T, calls


class Recorder:
    """Random source for the stand-in: seeded draws, recorded in call order as [(kind, array)]."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.entries = []

    def normal(self, shape):
        v = self.rng.standard_normal(shape)
        self.entries.append(('normal', v))
        return v

    def uniform(self, shape):
        v = self.rng.uniform(size=shape)
        self.entries.append(('uniform', v))
        return v

    def randint(self, lo, hi):
        v = int(self.rng.randint(lo, hi))
        self.entries.append(('randint', np.int64(v)))
        return v


def pack_tape(out, prefix, entries):
    out[prefix + 'tape_kinds'] = np.array([k for k, _ in entries])
    for i, (_, v) in enumerate(entries):
        out['%stape_%03d' % (prefix, i)] = np.asarray(v)


UPFIRDN_CASES = [   # (H, W, C, kh, kw, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
    (8, 8, 3, 4, 4, 1, 1, 1, 1, 2, 1, 2, 1),
    (8, 8, 2, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1),        # upsample_2d
    (8, 8, 2, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1),        # downsample_2d
    (9, 9, 4, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1),        # after a transposed 3x3 conv (G Conv0_up)
    (8, 8, 4, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2),        # in front of a strided 3x3 conv (D Conv1_down)
    (7, 5, 3, 3, 2, 2, 3, 3, 2, 1, 2, 0, 3),        # anisotropic everything
    (6, 6, 1, 4, 4, 1, 1, 1, 1, -1, 2, 1, -1),      # negative padding = cropping
    (5, 4, 2, 1, 1, 2, 2, 1, 1, 0, 1, 0, 1),        # 1x1 kernel
]


def main():
    U, F, N, T, calls = load_reference()
    tf = np_tf
    out = {}
    rng = np.random.RandomState(20261003)

    # ---- _setup_kernel
    for i, k in enumerate([[1, 3, 3, 1], [1, 1], [1, 2, 1], [[1, 2], [3, 4]], [1, 4, 6, 4, 1]]):
        out['setup_kernel_%d_in' % i] = np.asarray(k, np.float64)
        out['setup_kernel_%d_out' % i] = np.array(U._setup_kernel(k), np.float64)
        assert U._setup_kernel(k).dtype == np.float32

    # ---- upfirdn_2d: reference implementation, and the gradient the op defines (1st + 2nd order)
    for i, (H, W, C, kh, kw, upx, upy, downx, downy, px0, px1, py0, py1) in enumerate(UPFIRDN_CASES):
        x = rng.randn(2, H, W, C)
        k = rng.randn(kh, kw).astype(np.float32)
        kw_ = dict(upx=upx, upy=upy, downx=downx, downy=downy, padx0=px0, padx1=px1, pady0=py0, pady1=py1)
        with tf.session({}):
            y = U.upfirdn_2d(tf.Tensor(x), k, impl='ref', **kw_)
            calls['upfirdn'].clear()
            yc = U.upfirdn_2d(tf.Tensor(x), k, impl='cuda', **kw_)
            assert np.array_equal(y.v, yc.v)
            dy = rng.randn(*y.v.shape)
            dx = yc.grad_fn(tf.Tensor(dy))                   # grad(dy) of _upfirdn_2d_cuda.func
            ddx = rng.randn(*x.shape)
            d_dy = dx.grad_fn(tf.Tensor(ddx))                # its gradient: func again
            fwd, bwd, fwd2 = calls['upfirdn']
        assert dx.v.shape == x.shape and d_dy.v.shape == y.v.shape and fwd2['padx0'] == px0 and np.array_equal(fwd2['k'], fwd['k'])
        p = 'upfirdn_%d_' % i
        out.update({p + 'x': x, p + 'k': k.astype(np.float64), p + 'params': np.array([upx, upy, downx, downy, px0, px1, py0, py1]), p + 'y': y.v,
                    p + 'dy': dy, p + 'dx': dx.v, p + 'ddx': ddx, p + 'd_dy': d_dy.v, p + 'grad_k': bwd['k'].astype(np.float64),
                    p + 'grad_params': np.array([bwd[n] for n in ('upx', 'upy', 'downx', 'downy', 'padx0', 'padx1', 'pady0', 'pady1')])})
    out['upfirdn_cases'] = np.array(len(UPFIRDN_CASES))

    # ---- the wrappers: kernel gain + pad arithmetic (recorded) and values
    wrappers = []
    x = rng.randn(2, 3, 8, 8)
    for name, fn, kws in [('upsample_2d', U.upsample_2d, [dict(k=[1, 3, 3, 1]), dict(k=None), dict(k=[1, 3, 3, 1], factor=2, gain=2.0), dict(k=[1, 2, 1], factor=3)]),
                          ('downsample_2d', U.downsample_2d, [dict(k=[1, 3, 3, 1]), dict(k=None), dict(k=[1, 2, 1], factor=2, gain=0.5), dict(k=[1, 3, 3, 1], factor=4)]),
                          ('filter_2d', U.filter_2d, [dict(k=[1, 3, 3, 1]), dict(k=[1, 2, 1], gain=3.0)])]:
        for j, kw in enumerate(kws):
            with tf.session({}):
                calls['simple'].clear()
                y = fn(tf.Tensor(x), impl='ref', **kw)
                c = calls['simple'][0]
            p = '%s_%d_' % (name, j)
            out.update({p + 'y': y.v, p + 'k': c['k'].astype(np.float64), p + 'call': np.array([c['up'], c['down'], c['pad0'], c['pad1']]),
                        p + 'k_in': np.asarray(kw['k'] if kw.get('k') is not None else [], np.float64), p + 'factor_gain': np.array([kw.get('factor', 2 if name != 'filter_2d' else 1), kw.get('gain', 1.0)])})
            wrappers.append(p)
    out['wrapper_x'] = x
    for name, fn in [('upsample_conv_2d', U.upsample_conv_2d), ('conv_downsample_2d', U.conv_downsample_2d)]:
        for j, (ksz, kk, factor, gain) in enumerate([(3, [1, 3, 3, 1], 2, 1), (1, [1, 3, 3, 1], 2, 1), (3, None, 2, 1), (3, [1, 2, 1], 2, 1.5)]):
            w = rng.randn(ksz, ksz, 3, 5)
            with tf.session({}):
                calls['simple'].clear()
                y = fn(tf.Tensor(x), tf.Tensor(w), k=kk, factor=factor, gain=gain, impl='ref')
                c = calls['simple'][0]
            p = '%s_%d_' % (name, j)
            out.update({p + 'w': w, p + 'y': y.v, p + 'k': c['k'].astype(np.float64), p + 'call': np.array([c['up'], c['down'], c['pad0'], c['pad1']]),
                        p + 'k_in': np.asarray(kk if kk is not None else [], np.float64), p + 'factor_gain': np.array([factor, gain], np.float64)})
            wrappers.append(p)
    out['wrapper_cases'] = np.array(wrappers)

    # ---- fused_bias_act: table + reference implementation
    names = list(F.activation_funcs)
    out['act_names'] = np.array(names)
    out['act_def_alpha'] = np.array([np.nan if F.activation_funcs[n].def_alpha is None else F.activation_funcs[n].def_alpha for n in names])
    out['act_def_gain'] = np.array([F.activation_funcs[n].def_gain for n in names], np.float64)
    out['act_cuda_idx'] = np.array([F.activation_funcs[n].cuda_idx for n in names])
    out['act_ref'] = np.array([F.activation_funcs[n].ref for n in names])
    out['act_zero_2nd_grad'] = np.array([F.activation_funcs[n].zero_2nd_grad for n in names])
    xa = rng.randn(3, 5, 4, 4) * 2
    ba = rng.randn(5)
    out['act_x'], out['act_b'] = xa, ba
    for n in names:
        with tf.session({}):
            out['act_%s_default' % n] = F.fused_bias_act(tf.Tensor(xa), b=tf.Tensor(ba), act=n, impl='ref').v
            out['act_%s_custom' % n] = F.fused_bias_act(tf.Tensor(xa), b=tf.Tensor(ba), act=n, alpha=0.3, gain=0.7, impl='ref').v
            out['act_%s_nobias_axis3' % n] = F.fused_bias_act(tf.Tensor(xa), b=None, axis=3, act=n, impl='cuda').v
    with tf.session({}):
        out['act_lrelu_axis3'] = F.fused_bias_act(tf.Tensor(xa), b=tf.Tensor(rng.randn(4)), axis=3, act='lrelu', impl='ref').v
    out['act_b_axis3'] = np.array(np_tf.STATE.params.get('unused', 0)) if False else None
    del out['act_b_axis3']

    # ---- layers with injected variables
    def put(prefix, d):
        for k, v in d.items():
            a = np.asarray(v)
            out[prefix + k.replace('/', '.')] = a.astype(np.float32) if a.dtype == np.float64 and np.array_equal(a.astype(np.float32).astype(np.float64), a) else a

    layer_cases = []
    for j, (kernel, up, down, demod, fused, cin, cout, hw) in enumerate([
            (3, False, False, True, True, 6, 5, 8), (3, False, False, True, False, 6, 5, 8), (3, True, False, True, True, 6, 5, 4),
            (3, True, False, True, False, 6, 5, 4), (3, False, True, True, False, 6, 5, 8), (1, False, False, False, True, 6, 3, 8),
            (1, False, False, False, False, 6, 3, 8), (3, False, True, True, True, 4, 7, 8)]):
        params = {'L/weight': rng.randn(kernel, kernel, cin, cout), 'L/mod_weight': rng.randn(12, cin), 'L/mod_bias': rng.randn(cin) * 0.3}
        xin, yin = rng.randn(3, cin, hw, hw), rng.randn(3, 12)
        with tf.session(params):
            with tf.variable_scope('L'):
                y = N.modulated_conv2d_layer(tf.Tensor(xin), tf.Tensor(yin), fmaps=cout, kernel=kernel, up=up, down=down, demodulate=demod,
                                             resample_kernel=[1, 3, 3, 1], fused_modconv=fused)
        p = 'modconv_%d_' % j
        put(p + 'param.', params)
        out.update({p + 'x': xin, p + 'y_in': yin, p + 'out': y.v, p + 'cfg': np.array([kernel, up, down, demod, fused, cout])})
        layer_cases.append(p)
    out['modconv_cases'] = np.array(layer_cases)

    for j, (kernel, up, down, gain, lrmul, use_wscale) in enumerate([(3, False, False, 1, 1, True), (3, True, False, 1, 1, True), (3, False, True, np.sqrt(2), 1, True),
                                                                       (1, False, True, 1, 0.5, True), (1, True, False, 1, 1, False)]):
        params = {'C/weight': rng.randn(kernel, kernel, 4, 6), 'C/bias': rng.randn(6)}
        xin = rng.randn(2, 4, 8, 8)
        with tf.session(params):
            with tf.variable_scope('C'):
                y = N.conv2d_layer(tf.Tensor(xin), fmaps=6, kernel=kernel, up=up, down=down, resample_kernel=[1, 3, 3, 1], gain=gain, lrmul=lrmul, use_wscale=use_wscale)
                z = N.apply_bias_act(y, act='lrelu', lrmul=lrmul)
        p = 'conv_%d_' % j
        put(p + 'param.', params)
        out.update({p + 'x': xin, p + 'out': y.v, p + 'act': z.v, p + 'cfg': np.array([kernel, up, down, gain, lrmul, use_wscale], np.float64)})
    out['conv_cases'] = np.array(5)

    params = {'D/weight': rng.randn(2 * 3 * 3, 7), 'D/bias': rng.randn(7)}
    xin = rng.randn(4, 2, 3, 3)
    with tf.session(params):
        with tf.variable_scope('D'):
            y = N.apply_bias_act(N.dense_layer(tf.Tensor(xin), fmaps=7, gain=np.sqrt(2), lrmul=0.01), act='lrelu', lrmul=0.01)
    put('dense_param.', params)
    out.update(dense_x=xin, dense_out=y.v)

    for j, (n, g) in enumerate([(6, 6), (12, 6), (4, 6), (8, 4)]):
        xin = rng.randn(n, 5, 4, 4)
        with tf.session({}):
            out['mbstd_%d_out' % j] = N.minibatch_stddev_layer(tf.Tensor(xin), group_size=g, num_new_features=1).v
        out['mbstd_%d_x' % j] = xin
        out['mbstd_%d_cfg' % j] = np.array([n, g])
    out['mbstd_cases'] = np.array(4)

    # ---- whole networks: variable values from the product's Network on the CPU (same names, same shapes as the reference's)
    from inclusivegan_amd.dnnlib import tflib as P
    RES, FMAP = 16, 64
    SMALL = dict(latent_size=32, dlatent_size=48, mapping_fmaps=40)     # keeps the fixture small; every code path is the same

    def net_params(kind, arch, seed):
        fn = 'inclusivegan_amd.training.networks_stylegan2.' + ('G_main' if kind == 'G' else 'D_stylegan2_feature')
        net = P.Network(kind, func_name=fn, num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, architecture=arch, device='cpu', seed=seed, **(SMALL if kind == 'G' else {}))
        prng = np.random.RandomState(seed)
        vals = {}
        for name, v in net.vars.items():
            a = v.detach().numpy().astype(np.float64)
            if name.endswith('bias') or name.endswith('noise_strength') or name == 'dlatent_avg':
                a = np.asarray(prng.randn(*a.shape) * 0.2).astype(np.float32).astype(np.float64)           # zero-initialised in training: make them matter
            vals[name] = a
        return vals

    class RefNetwork:        # tflib.Network as G_main uses it: components with a scope, input_shape, vars, get_output_for
        def __init__(self, name, func_name=None, **static):
            self.name, self.func, self.static, self.vars = name, func_name, static, {}
            res_log2 = int(np.log2(static.get('resolution', 1024)))
            self.input_shape = [None, res_log2 * 2 - 2, static.get('dlatent_size', 512)]

        def get_output_for(self, *inputs, **dyn):
            kw = dict(self.static); kw.update(dyn)
            with tf.variable_scope(tf.VariableScope([self.name])):       # absolute scope, like Network.get_output_for (network.py:216)
                return self.func(*inputs, **kw)
    N.tflib.Network = RefNetwork

    net_cases = []
    for arch in ('skip', 'resnet', 'orig'):
        gp = net_params('G', arch, 100 + len(net_cases))
        z = rng.randn(3, SMALL['latent_size'])
        lab = np.zeros((3, 0))
        for mode, kw in [('train', dict(is_training=True)), ('val', dict(is_validation=True, truncation_psi_val=0.7, truncation_cutoff_val=4)),
                         ('plain', dict(truncation_psi=0.5, randomize_noise=False))]:
            for fused in ((True, False) if mode == 'train' and arch == 'skip' else (True,)):
                rec = Recorder(7 + len(net_cases))
                with tf.session(gp, rec) as st:
                    img, dl = N.G_main(tf.Tensor(z), tf.Tensor(lab), components=N.dnnlib.EasyDict(), return_dlatents=True, resolution=RES, fmap_base=FMAP,
                                       architecture=arch, num_channels=3, fused_modconv=fused, **SMALL, **kw)
                    assigned = dict(st.assigned)
                entries = list(rec.entries)
                if mode == 'train':       # the cutoff draw only happens when the coin says "mix" (tf.cond); restatements draw it always:
                    coin = [i for i, (k, v) in enumerate(entries) if k == 'uniform' and np.ndim(v) == 0][0]
                    if not (entries[coin + 1][0] == 'randint'):
                        entries.insert(coin + 1, ('randint', np.int64(1)))          # a placeholder the reference never looked at
                p = 'G_%s_%s_%d_' % (arch, mode, int(fused))
                out.update({p + 'z': z, p + 'img': img.v, p + 'dlatents': dl.v, p + 'dlatent_avg_after': assigned.get('dlatent_avg', gp['dlatent_avg'])})
                pack_tape(out, p, entries)
                net_cases.append(p)
        put('Gparam_%s.' % arch, gp)
        dp = net_params('D', arch, 200 + len(net_cases))
        xin = rng.randn(6, 3, RES, RES)
        with tf.session(dp):
            s, f = N.D_stylegan2_feature(tf.Tensor(xin), tf.Tensor(np.zeros((6, 0))), resolution=RES, fmap_base=FMAP, architecture=arch, num_channels=3)
        put('Dparam_%s.' % arch, dp)
        out.update({'D_%s_x' % arch: xin, 'D_%s_scores' % arch: s.v, 'D_%s_features' % arch: f.v})
    out['G_cases'] = np.array(net_cases)
    out['net_cfg'] = np.array([RES, FMAP, SMALL['latent_size'], SMALL['dlatent_size'], SMALL['mapping_fmaps']])

    # ---- tfutil: lerp / slerp on tensors (tfutil.py:62-87)
    a, b, t = rng.randn(5, 16), rng.randn(5, 16), rng.rand(5, 1)
    out.update(tf_lerp_a=a, tf_lerp_b=b, tf_lerp_t=t, tf_lerp_out=T.lerp(tf.Tensor(a), tf.Tensor(b), tf.Tensor(t)).v, tf_slerp_out=T.slerp(tf.Tensor(a), tf.Tensor(b), tf.Tensor(t)).v)

    path = os.path.join(HERE, 'ref_ops_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.2f MB' % (path, len(out), os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    main()
