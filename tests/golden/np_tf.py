"""A NumPy stand-in for the handful of TensorFlow 1.x primitives the reference's network / op code calls, so that the
reference's OWN Python (dnnlib/tflib/ops/upfirdn_2d.py, fused_bias_act.py, training/networks_stylegan2.py -- imported from
/root/reference where it lies, by tests/golden/make_ref_ops_golden.py) can be EXECUTED in this container, which has no TensorFlow.

What this pins and what it does not: every statement of the reference that composes primitives -- pad amounts, kernel flips,
reshapes / transposes, strides, gains, the order of scale / bias / noise / activation, the layer -> dlatent index map, nf() --
runs unchanged.  The primitives themselves are restated here from TensorFlow's documented semantics:

    tf.nn.conv2d            cross-correlation, filter HWIO, NCHW / NHWC, strides, 'VALID' or 'SAME' (SAME: total padding
                            max((ceil(in / s) - 1) * s + k - in, 0), the smaller half in front)
    tf.nn.conv2d_transpose  the gradient of conv2d w.r.t. its input: filter [h, w, out_channels, in_channels],
                            out[n, o, i*s + ky, j*s + kx] += x[n, c, i, j] * w[ky, kx, o, c], 'VALID', explicit output_shape
    tf.pad                  zero padding;  tf.tile, tf.reshape, tf.transpose, tf.concat, tf.where, tf.matmul, reductions: NumPy's
    tf.nn.leaky_relu(x, a)  max(x, a * x) ... (activations: the textbook formulas)
    tf.random_normal / random_uniform     the next entry of an injected tape (TF's Philox streams cannot be reproduced)
    tf.get_variable         looked up by '<variable scopes>/<name>' in an injected dict (the reference's own variable names)

Round 4 adds what the TRAINING statements call (training/loss.py, training_loop.process_reals, dnnlib/tflib/optimizer.py,
Network.setup_as_moving_average_of; driven by tests/golden/make_ref_train_golden.py):

    tf.Variable             a mutable Tensor; under `variable_replay(store)` the k-th creation of a run returns the k-th
                            variable of `store` (so re-running graph-BUILDING code once per step acts on persistent state:
                            every read of a variable in those files precedes its assign through a data dependency, so the
                            eager order is the order a session.run would produce)
    tf.assign / assign_sub / Variable.assign    in-place update of the variable's value
    tf.gradients            cannot be executed: answered by an injected hook `session(..., grad_hook=f)`; f(ys, xs) -> list
    tf.where                TF1 semantics: a rank-1 condition selects whole rows of higher-rank operands
    tf.cond / group / no_op / stack / add_n / reduce_all / is_finite / zeros / zeros_like / floor / reverse / split / div
    tf.device               recorded on the tensors created inside (Optimizer keys its per-device state on it)

All arithmetic is float64 (dtype labels are kept and compared, values are never rounded to float32), so a restatement that
follows the same formulas agrees to ~1e-13.  TEST INFRASTRUCTURE ONLY -- never imported by the package."""
import contextlib
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np


# ---------------------------------------------------------------------------------------------------------------
# dtypes, shapes, tensors

class DType:
    def __init__(self, name):
        self.name = name
        self.base_dtype = self
        self.is_floating = name.startswith('float')
        self.is_integer = name.startswith('int') or name.startswith('uint')

    @property
    def as_numpy_dtype(self):
        return np.float64 if self.is_floating else np.dtype(self.name).type

    def __eq__(self, other):
        return as_dtype(other).name == self.name if other is not None else False

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return 'tf.' + self.name


_DTYPES = {n: DType(n) for n in ('float16', 'float32', 'float64', 'int32', 'int64', 'uint8', 'bool')}
float16, float32, float64, int32, int64, uint8 = (_DTYPES[n] for n in ('float16', 'float32', 'float64', 'int32', 'int64', 'uint8'))


def as_dtype(d):
    if isinstance(d, DType):
        return d
    if isinstance(d, str):
        return _DTYPES[d]
    return _DTYPES[np.dtype(d).name]


class Dimension:
    def __init__(self, v):
        self.value = None if v is None else int(v)

    def __int__(self):
        return self.value

    __index__ = __int__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dimension) else o)

    def __hash__(self):
        return hash(self.value)

    def __floordiv__(self, o):
        return Dimension(self.value // int(o))

    def __mul__(self, o):
        return Dimension(self.value * int(o))

    __rmul__ = __mul__

    def __repr__(self):
        return 'Dimension(%r)' % self.value


class TensorShape:
    def __init__(self, dims):
        self.dims = [d if isinstance(d, Dimension) else Dimension(d) for d in dims]
        self.rank = self.ndims = len(self.dims)

    def as_list(self):
        return [d.value for d in self.dims]

    def __getitem__(self, i):
        return TensorShape(self.dims[i]) if isinstance(i, slice) else self.dims[i]

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __eq__(self, o):
        return self.as_list() == (o.as_list() if isinstance(o, TensorShape) else list(o))

    def __repr__(self):
        return 'TensorShape(%r)' % self.as_list()


def _val(x):
    """-> NumPy value of a Tensor / Dimension / scalar / nested list."""
    if isinstance(x, Tensor):
        return x.v
    if isinstance(x, Dimension):
        return x.value
    if isinstance(x, (list, tuple)):
        return [_val(e) for e in x]
    return x


def _ints(seq):
    return [int(_val(e)) for e in seq]


class Tensor:
    __array_priority__ = 1000       # ndarray <op> Tensor -> Tensor.__r<op>__

    def __init__(self, v, dtype=None):
        v = np.asarray(_val(v))
        if dtype is None:
            dtype = float32 if v.dtype.kind == 'f' else as_dtype(v.dtype)
        self.dtype = as_dtype(dtype)
        self.v = v.astype(np.float64) if self.dtype.is_floating else v
        self.name = 'tensor'
        self.device = STATE.device if 'STATE' in globals() else ''

    @property
    def shape(self):
        return TensorShape(self.v.shape)

    def get_shape(self):
        return self.shape

    def set_shape(self, shape):
        want = [None if s is None else int(_val(s)) for s in (shape.as_list() if isinstance(shape, TensorShape) else shape)]
        assert len(want) == self.v.ndim and all(w is None or w == h for w, h in zip(want, self.v.shape)), (want, self.v.shape)

    def _bin(self, o, f, r=False):
        a, b = self.v, np.asarray(_val(o))
        out = f(b, a) if r else f(a, b)
        return Tensor(out, self.dtype if out.dtype.kind == 'f' else None)

    __add__ = lambda s, o: s._bin(o, np.add);            __radd__ = lambda s, o: s._bin(o, np.add, True)
    __sub__ = lambda s, o: s._bin(o, np.subtract);       __rsub__ = lambda s, o: s._bin(o, np.subtract, True)
    __mul__ = lambda s, o: s._bin(o, np.multiply);       __rmul__ = lambda s, o: s._bin(o, np.multiply, True)
    __truediv__ = lambda s, o: s._bin(o, np.divide);     __rtruediv__ = lambda s, o: s._bin(o, np.divide, True)
    __lt__ = lambda s, o: s._bin(o, np.less);            __le__ = lambda s, o: s._bin(o, np.less_equal)
    __gt__ = lambda s, o: s._bin(o, np.greater);         __ge__ = lambda s, o: s._bin(o, np.greater_equal)
    __neg__ = lambda s: Tensor(-s.v, s.dtype)
    __pow__ = lambda s, o: s._bin(o, np.power);          __rpow__ = lambda s, o: s._bin(o, np.power, True)
    __floordiv__ = lambda s, o: s._bin(o, np.floor_divide); __rfloordiv__ = lambda s, o: s._bin(o, np.floor_divide, True)
    __mod__ = lambda s, o: s._bin(o, np.mod)

    def __int__(self):
        return int(self.v)

    __index__ = __int__

    def __float__(self):
        return float(self.v)

    def __hash__(self):
        return id(self)

    def __eq__(self, o):          # identity, like TF1 tensors (they are dictionary keys in Optimizer)
        return self is o

    def __getitem__(self, idx):
        return Tensor(self.v[idx], self.dtype)

    def __bool__(self):
        return bool(self.v)

    def __repr__(self):
        return 'Tensor(shape=%s, dtype=%s)' % (self.v.shape, self.dtype.name)


class Variable(Tensor):
    """tf.Variable(initial_value, trainable, name, dtype).  Under `variable_replay` creation returns persistent objects."""

    def __new__(cls, initial_value=None, *a, **k):
        st = globals().get('STATE')
        if st is not None and st.var_store is not None:
            if st.var_cursor < len(st.var_store):
                v = st.var_store[st.var_cursor]
                st.var_cursor += 1
                return v
            v = super().__new__(cls)
            st.var_store.append(v)
            st.var_cursor += 1
            return v
        return super().__new__(cls)

    def __init__(self, initial_value=None, dtype=None, name=None, trainable=True, **_kw):
        if getattr(self, '_made', False):
            return                    # replayed: keep the current value
        if isinstance(dtype, bool):   # positional (initial_value, trainable) is never used by the reference; guard anyway
            trainable, dtype = dtype, None
        Tensor.__init__(self, _val(initial_value), dtype if dtype is not None else (initial_value.dtype if isinstance(initial_value, Tensor) else None))
        self.name = name or 'Variable'
        self.trainable = trainable
        self.initializer = ('init', self)
        self._made = True

    def assign(self, value):
        return assign(self, value)


# ---------------------------------------------------------------------------------------------------------------
# graph-side state: variable scopes, injected variables and random draws

class State:
    def __init__(self):
        self.scopes = []
        self.params = {}
        self.tape = None            # object with .normal(shape) / .uniform(shape) / .randint(lo, hi) returning arrays
        self.assigned = {}
        self.created = []
        self.device = ''
        self.var_store = None       # list of Variables when replaying (see Variable.__new__)
        self.var_cursor = 0
        self.grad_hook = None

STATE = State()


@contextlib.contextmanager
def session(params, tape=None, grad_hook=None):
    """Run reference code against `params` (full variable name -> array), `tape` (random draws, in call order) and
    `grad_hook` (the answer to tf.gradients)."""
    old = (STATE.params, STATE.tape, STATE.scopes, STATE.assigned, STATE.created, STATE.grad_hook)
    STATE.params, STATE.tape, STATE.scopes, STATE.assigned, STATE.created, STATE.grad_hook = dict(params), tape, [], {}, [], grad_hook
    try:
        yield STATE
    finally:
        STATE.params, STATE.tape, STATE.scopes, STATE.assigned, STATE.created, STATE.grad_hook = old


@contextlib.contextmanager
def variable_replay(store):
    """Inside, the k-th `tf.Variable(...)` creation returns store[k] (created and appended on first use)."""
    old = (STATE.var_store, STATE.var_cursor)
    STATE.var_store, STATE.var_cursor = store, 0
    try:
        yield store
    finally:
        STATE.var_store, STATE.var_cursor = old


@contextlib.contextmanager
def device(name):
    old, STATE.device = STATE.device, (name or '')
    try:
        yield
    finally:
        STATE.device = old


@contextlib.contextmanager
def variable_scope(name=None, reuse=None, **_kw):
    if isinstance(name, VariableScope):
        saved, STATE.scopes = STATE.scopes, list(name.path)
        try:
            yield name
        finally:
            STATE.scopes = saved
        return
    STATE.scopes.append(name)
    try:
        yield VariableScope(STATE.scopes)
    finally:
        STATE.scopes.pop()


class VariableScope:
    def __init__(self, path):
        self.path = list(path)
        self.name = '/'.join(path)


@contextlib.contextmanager
def name_scope(name=None, *a, **k):
    yield name


@contextlib.contextmanager
def control_dependencies(deps):
    yield


def get_variable(name, shape=None, initializer=None, trainable=True, dtype=None, **_kw):
    full = '/'.join(STATE.scopes + [name])
    STATE.created.append(full)
    if full not in STATE.params:
        raise KeyError('np_tf.get_variable: no value injected for %r' % full)
    saved, STATE.var_store = STATE.var_store, None       # a named variable is looked up, never replayed by creation order
    try:
        v = Variable(np.asarray(STATE.params[full]), float32)
    finally:
        STATE.var_store = saved
    v.name = full
    if shape is not None:
        assert list(v.v.shape) == _ints(shape), (full, v.v.shape, _ints(shape))
    return v


def assign(var, value):
    new = np.array(_val(value), dtype=np.float64 if var.dtype.is_floating else None)
    assert new.shape == var.v.shape, (var.name, new.shape, var.v.shape)
    var.v = new
    STATE.assigned[var.name] = new
    if var.name in STATE.params:
        STATE.params[var.name] = new
    return Tensor(new, var.dtype)


def assign_sub(var, value):
    return assign(var, var.v - np.asarray(_val(value)))


def assign_add(var, value):
    return assign(var, var.v + np.asarray(_val(value)))


def gradients(ys, xs, *a, **k):
    assert STATE.grad_hook is not None, 'np_tf.gradients: tf.gradients cannot be executed; inject session(grad_hook=...)'
    out = STATE.grad_hook(ys, xs)
    return [g if g is None or isinstance(g, Tensor) else Tensor(g, float32) for g in out]


class initializers:
    random_normal = staticmethod(lambda *a, **k: ('random_normal', a, k))
    zeros = staticmethod(lambda *a, **k: ('zeros',))
    ones = staticmethod(lambda *a, **k: ('ones',))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, **_kw):
    shape = _ints(np.asarray(_val(shape)).reshape(-1).tolist()) if not isinstance(shape, (list, tuple)) else _ints(shape)
    out = np.asarray(STATE.tape.normal(shape), dtype=np.float64)
    assert list(out.shape) == shape, (out.shape, shape)
    return Tensor(out * stddev + mean, dtype)


def random_uniform(shape, minval=0, maxval=None, dtype=float32, **_kw):
    dtype = as_dtype(dtype)
    if dtype.is_integer:
        return Tensor(np.int64(STATE.tape.randint(int(_val(minval)), int(_val(maxval)))), dtype)
    u = np.asarray(STATE.tape.uniform(_ints(shape)), dtype=np.float64)
    maxval = 1.0 if maxval is None else maxval
    return Tensor(u * (maxval - minval) + minval, dtype)


# ---------------------------------------------------------------------------------------------------------------
# element-wise / shape primitives

def convert_to_tensor(x, dtype=None, **_kw):
    return x if isinstance(x, Tensor) and dtype is None else Tensor(x, dtype if dtype is not None else (x.dtype if isinstance(x, Tensor) else None))


def constant(v, dtype=None, shape=None, **_kw):
    return Tensor(np.asarray(_val(v), dtype=np.float64 if dtype is None or as_dtype(dtype).is_floating else None), dtype)


def cast(x, dtype):
    d = as_dtype(dtype)
    v = np.asarray(_val(x))
    return Tensor(v.astype(np.int64) if d.is_integer else v, d)


def identity(x, name=None):
    return x


def shape(x):
    return Tensor(np.array(np.asarray(_val(x)).shape, dtype=np.int64), int32)


def reshape(x, shp):
    return Tensor(np.reshape(_val(x), _ints(shp)), x.dtype)


def transpose(x, perm=None):
    return Tensor(np.transpose(_val(x), perm), x.dtype)


def pad(x, paddings, **_kw):
    return Tensor(np.pad(_val(x), [tuple(_ints(p)) for p in paddings]), x.dtype)


def tile(x, multiples):
    return Tensor(np.tile(_val(x), _ints(multiples)), x.dtype)


def concat(values, axis):
    return Tensor(np.concatenate([_val(v) for v in values], axis=axis), values[0].dtype)


def squeeze(x, axis=None, **_kw):
    return Tensor(np.squeeze(_val(x), axis=None if axis is None else tuple(np.atleast_1d(axis))), x.dtype)


def broadcast_to(x, shp):
    return Tensor(np.broadcast_to(_val(x), _ints(np.asarray(_val(shp)).tolist())), x.dtype if isinstance(x, Tensor) else None)


def where(cond, a, b):
    c, av, bv = np.asarray(_val(cond)), np.asarray(_val(a)), np.asarray(_val(b))
    if c.ndim == 1 and av.ndim > 1:       # TF1: a vector condition picks rows
        assert c.shape[0] == av.shape[0]
        c = c.reshape((-1,) + (1,) * (av.ndim - 1))
    else:
        assert c.shape == av.shape or c.ndim == 0, (c.shape, av.shape)
    return Tensor(np.where(c, av, bv), a.dtype if isinstance(a, Tensor) else (b.dtype if isinstance(b, Tensor) else None))


def cond(pred, true_fn, false_fn):
    return true_fn() if bool(_val(pred)) else false_fn()


class Operation:
    def __init__(self, parts=()):
        self.parts = list(parts)
        self.device = STATE.device


def group(*ops, name=None):
    return Operation(ops)


def no_op(name=None):
    return Operation()


def stack(values, axis=0):
    return Tensor(np.stack([np.asarray(_val(v)) for v in values], axis=axis))


def add_n(values):
    out = np.asarray(_val(values[0])).copy()
    for v in values[1:]:
        out = out + np.asarray(_val(v))
    return Tensor(out, values[0].dtype)


def zeros(shape, dtype=None, **_kw):
    shp = shape.as_list() if isinstance(shape, TensorShape) else _ints(shape)
    return Tensor(np.zeros(shp), dtype if dtype is not None else float32)


def zeros_like(x, **_kw):
    return Tensor(np.zeros_like(np.asarray(_val(x))), x.dtype if isinstance(x, Tensor) else None)


def is_finite(x):
    return Tensor(np.isfinite(np.asarray(_val(x))))


def reduce_all(x, axis=None, **_kw):
    return Tensor(np.all(np.asarray(_val(x)), axis=axis))


def floor(x):
    return Tensor(np.floor(np.asarray(_val(x), dtype=np.float64)), x.dtype if isinstance(x, Tensor) else float32)


def reverse(x, axis):
    return Tensor(np.flip(_val(x), axis=tuple(_ints(axis))), x.dtype)


def split(x, num, axis=0):
    return [Tensor(p, x.dtype) for p in np.split(_val(x), int(_val(num)), axis=axis)]


def div(a, b):
    return Tensor(np.asarray(_val(a)) / np.asarray(_val(b)), a.dtype if isinstance(a, Tensor) else float32)


class _Graph:
    def __init__(self):
        self.names = {}

    def unique_name(self, name):
        n = self.names.get(name, 0)
        self.names[name] = n + 1
        return name if n == 0 else '%s_%d' % (name, n)


_GRAPH = _Graph()


def get_default_graph():
    return _GRAPH


def get_default_session():
    return _GRAPH           # anything that is not None: "a session exists" (tfutil.assert_tf_initialized)


def minimum(a, b):
    out = np.minimum(_val(a), _val(b))
    return Tensor(out, a.dtype if isinstance(a, Tensor) else (b.dtype if isinstance(b, Tensor) else None))


def _red(f):
    def g(x, axis=None, keepdims=False, **_kw):
        ax = None if axis is None else tuple(np.atleast_1d(axis).tolist())
        return Tensor(f(_val(x), axis=ax, keepdims=keepdims), x.dtype)
    return g


reduce_sum, reduce_mean = _red(np.sum), _red(np.mean)


def _un(f):
    return lambda x, *a, **k: Tensor(f(np.asarray(_val(x), dtype=np.float64)), x.dtype if isinstance(x, Tensor) else float32)


square, sqrt, exp, log = _un(np.square), _un(np.sqrt), _un(np.exp), _un(np.log)
rsqrt = _un(lambda v: 1.0 / np.sqrt(v))


def matmul(a, b):
    return Tensor(np.matmul(_val(a), _val(b)), a.dtype)


def clip_by_value(x, lo, hi):
    return Tensor(np.clip(_val(x), lo, hi), x.dtype)


def custom_gradient(f):
    """y = f(x)[0]; the gradient closure the reference attaches is kept on the result (`.grad_fn`) so that it can be
    called explicitly (the reference defines its ops' gradients through these closures)."""
    def wrapped(*args):
        y, grad = f(*args)
        if isinstance(y, Tensor):
            y = Tensor(y.v, y.dtype)
            y.grad_fn = grad
        return y
    return wrapped


class math:
    sin, cos, acos = _un(np.sin), _un(np.cos), _un(np.arccos)


# ---------------------------------------------------------------------------------------------------------------
# tf.nn

def _conv2d_nchw(x, w, sy, sx, padding):
    n, c, h, wd = x.shape
    kh, kw, ci, co = w.shape
    # grouped convolution (TF: filter in_channels divides the input's; group g = input channels [g*ci, (g+1)*ci) ->
    # output channels [g*co/groups, (g+1)*co/groups)) -- the fused modulated convolution uses it (networks_stylegan2.py:108-110)
    assert c % ci == 0 and co % (c // ci) == 0, (x.shape, w.shape)
    groups = c // ci
    cog = co // groups
    if padding == 'SAME':
        oh, ow = -(-h // sy), -(-wd // sx)
        ph, pw = max((oh - 1) * sy + kh - h, 0), max((ow - 1) * sx + kw - wd, 0)
        x = np.pad(x, [(0, 0), (0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)])
    else:
        assert padding == 'VALID'
    oh, ow = (x.shape[2] - kh) // sy + 1, (x.shape[3] - kw) // sx + 1
    out = np.zeros((n, co, oh, ow))
    for g in range(groups):
        for ky in range(kh):
            for kx in range(kw):
                patch = x[:, g * ci:(g + 1) * ci, ky:ky + (oh - 1) * sy + 1:sy, kx:kx + (ow - 1) * sx + 1:sx]
                out[:, g * cog:(g + 1) * cog] += np.einsum('nchw,co->nohw', patch, w[ky, kx, :, g * cog:(g + 1) * cog])
    return out


class nn:
    @staticmethod
    def conv2d(x, w, strides=None, padding=None, data_format='NHWC', **_kw):
        xv, wv = np.asarray(_val(x), np.float64), np.asarray(_val(w), np.float64)
        if data_format == 'NCHW':
            assert strides[0] == strides[1] == 1
            return Tensor(_conv2d_nchw(xv, wv, strides[2], strides[3], padding), x.dtype)
        assert strides[0] == strides[3] == 1
        return Tensor(_conv2d_nchw(xv.transpose(0, 3, 1, 2), wv, strides[1], strides[2], padding).transpose(0, 2, 3, 1), x.dtype)

    @staticmethod
    def conv2d_transpose(x, w, output_shape=None, strides=None, padding='VALID', data_format='NHWC', **_kw):
        assert padding == 'VALID'
        xv, wv = np.asarray(_val(x), np.float64), np.asarray(_val(w), np.float64)
        oshape = _ints(output_shape)
        if data_format != 'NCHW':
            xv = xv.transpose(0, 3, 1, 2)
            oshape = [oshape[0], oshape[3], oshape[1], oshape[2]]
            sy, sx = strides[1], strides[2]
        else:
            sy, sx = strides[2], strides[3]
        n, c, h, wd = xv.shape
        kh, kw, cog, ci = wv.shape
        # grouped: the filter's output dimension is per group (upfirdn_2d.py:283-286 builds it that way), its input dimension total
        assert ci == c and oshape[1] % cog == 0, (xv.shape, wv.shape, oshape)
        groups = oshape[1] // cog
        cig = c // groups
        out = np.zeros((n, oshape[1], (h - 1) * sy + kh, (wd - 1) * sx + kw))
        assert list(out.shape) == [n] + oshape[1:], (out.shape, oshape)
        for g in range(groups):
            for ky in range(kh):
                for kx in range(kw):
                    out[:, g * cog:(g + 1) * cog, ky:ky + (h - 1) * sy + 1:sy, kx:kx + (wd - 1) * sx + 1:sx] += \
                        np.einsum('nchw,oc->nohw', xv[:, g * cig:(g + 1) * cig], wv[ky, kx, :, g * cig:(g + 1) * cig])
        return Tensor(out if data_format == 'NCHW' else out.transpose(0, 2, 3, 1), x.dtype)

    relu = staticmethod(_un(lambda v: np.maximum(v, 0.0)))
    tanh = staticmethod(_un(np.tanh))
    sigmoid = staticmethod(_un(lambda v: 1.0 / (1.0 + np.exp(-v))))
    softplus = staticmethod(_un(lambda v: np.logaddexp(v, 0.0)))
    elu = staticmethod(_un(lambda v: np.where(v > 0, v, np.expm1(v))))
    selu = staticmethod(_un(lambda v: 1.0507009873554804934193349852946 * np.where(v > 0, v, 1.6732632423543772848170429916717 * np.expm1(v))))

    @staticmethod
    def leaky_relu(x, alpha=0.2, **_kw):
        v = np.asarray(_val(x), np.float64)
        return Tensor(np.maximum(v, v * float(alpha)), x.dtype if isinstance(x, Tensor) else float32)


# ---------------------------------------------------------------------------------------------------------------
# installing the stand-in as `tensorflow` (and a hollow `tensorboard`) for the import of the reference's modules

class _Hollow(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        m = _Hollow(self.__name__ + '.' + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Hollow('call')


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in ('tensorflow', 'tensorboard'):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        if spec.name == 'tensorflow':
            m = _Hollow('tensorflow')
            me = sys.modules[__name__]
            for k in dir(me):
                if not k.startswith('_') and k not in ('sys', 'types', 'np', 'contextlib', 'importlib'):
                    setattr(m, k, getattr(me, k))
            m.Tensor, m.Variable, m.Operation, m.Dimension, m.VariableScope = Tensor, Variable, Operation, Dimension, VariableScope
            return m
        return _Hollow(spec.name)

    def exec_module(self, module):
        pass


def install():
    """Make `import tensorflow` resolve to this stand-in (idempotent).  Only for processes that generate golden vectors."""
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
