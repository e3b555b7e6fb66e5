"""`Network`: the object model the reference's training code programs against
(dnnlib/tflib/network.py:35-590), restated for eager PyTorch-ROCm.

Kept semantics (the parts the hot path touches):
  * construction from a build function given by object or dotted name (`func_name`), static kwargs
    remembered and merged with per-call dynamic kwargs (network.py:84-123,199-232);
  * `vars` / `trainables`: ordered dicts keyed by the reference's *local variable names*
    (e.g. 'G_synthesis/64x64/Conv0_up/mod_weight', network.py:181-185) in creation order;
  * sub-networks created inside a build function become `components` whose variables appear in the
    parent under '<component name>/...' (network.py:125-185);
  * `get_output_for`, `clone`, `copy_vars_from`, `setup_as_moving_average_of`, `run`,
    `input_shapes` / `output_shapes`.
Dropped: TF graph/session mechanics, pickling of build-module source, weight histograms.

MI355X-specific design: all trainables of a top-level network are views into ONE contiguous fp32
bucket (`flat_params`, every tensor 16-byte aligned) with a matching gradient bucket
(`flat_grads`), so that the data-parallel exchange is a single RCCL all-reduce and Adam / EMA /
finite-check are single streaming kernels (optimizer.py) instead of the reference's one NCCL call
and one Adam chain per variable (dnnlib/tflib/optimizer.py:193-201,237-239).
"""
import inspect
from collections import OrderedDict

import numpy as np
import torch

from .. import util
from ..util import EasyDict
from . import tfutil
from ... import hip_ops

_ALIGN = 4  # floats (16 bytes)


def _resolve(func_name):
    if callable(func_name):
        return func_name
    return util.get_obj_by_name(func_name)


class Network:
    def __init__(self, name=None, func_name=None, device=None, seed=0, **static_kwargs):
        assert isinstance(name, str) or name is None
        assert func_name is not None
        self.name = name or 'net'
        self.static_kwargs = EasyDict(static_kwargs)
        self._build_func = _resolve(func_name)
        self._build_func_name = func_name if isinstance(func_name, str) else getattr(func_name, '__name__', str(func_name))
        self.components = EasyDict()
        self.seed = seed

        # Nested construction (inside a parent's build function) => component of that parent.
        parent = tfutil._stack().store
        if parent is not None:
            self._root = parent._root
            self._prefix = parent._prefix + self.name + '/'
            self.device = self._root.device
        else:
            self._root = self
            self._prefix = ''
            if device is None:
                if not torch.cuda.is_available():
                    raise RuntimeError('inclusivegan_amd.Network needs a ROCm device (or device="meta"/"cpu" for shape-only / host-side use)')
                device = torch.device('cuda', torch.cuda.current_device())
            self.device = torch.device(device)
            self._specs = OrderedDict()   # full name -> (shape, initializer, trainable)
            self._store = OrderedDict()   # full name -> tensor (after materialisation)
            self._templating = True
            self.flat_params = None
            self.flat_grads = None
            self._derived = {}            # per-training-op cache of tensors derived from variables only
            self._derived_depth = 0       # > 0 while inside derived_scope()

        # Template pass on the meta device: creates variable specs, infers shapes.
        self.input_names = [p.name for p in inspect.signature(self._build_func).parameters.values()
                            if p.kind == p.POSITIONAL_OR_KEYWORD and p.default is p.empty]
        templ = [torch.empty(self._template_shape(n), device='meta') for n in self.input_names]
        was_templating = self._root._templating
        self._root._templating = True
        with tfutil.variable_store(self), torch.no_grad():
            out = self._build_func(*templ, is_template_graph=True, components=self.components, **self.static_kwargs)
        self._root._templating = was_templating if parent is not None else False
        outs = list(out) if isinstance(out, (tuple, list)) else [out]
        self.input_shapes = [[None] + list(t.shape[1:]) for t in templ]
        self.output_shapes = [([None] + list(t.shape[1:])) if t is not None else None for t in outs]
        self.input_shape = self.input_shapes[0]
        self.output_shape = self.output_shapes[0]
        self.num_inputs = len(self.input_shapes)
        self.num_outputs = len(self.output_shapes)

        if parent is None:
            self._materialize()

    # ------------------------------------------------------------------
    def _template_shape(self, arg_name):
        kw = self.static_kwargs
        if arg_name == 'latents_in':
            return (1, kw.get('latent_size', 512))
        if arg_name == 'labels_in':
            return (1, kw.get('label_size', 0))
        if arg_name == 'images_in':
            r = kw.get('resolution', 1024)
            return (1, kw.get('num_channels', 3), r, r)
        if arg_name == 'dlatents_in':
            r = kw.get('resolution', 1024)
            return (1, int(np.log2(r)) * 2 - 2, kw.get('dlatent_size', 512))
        if arg_name in ('images_a', 'images_b'):
            r = kw.get('resolution', 64)
            return (1, 3, r, r)
        raise ValueError('Network: do not know the template shape of input %r' % arg_name)

    # variable access (called through tfutil.get_variable) -----------------
    def _get_variable(self, local_name, shape, initializer, trainable):
        root = self._root
        full = self._prefix + local_name
        if full in root._specs:
            if root._templating or root._store.get(full) is None:
                return torch.empty(root._specs[full][0], device='meta')
            return root._store[full]
        if not root._templating:
            raise KeyError('variable %r does not exist in network %r' % (full, root.name))
        shape = tuple(int(s) for s in (shape if shape is not None else ()))
        root._specs[full] = (shape, initializer or ('zeros',), bool(trainable))
        return torch.empty(shape, device='meta')

    def derived(self, key, fn):
        """Tensors that depend on the network's variables only (the equalised-LR weight w * runtime_coef,
        networks_stylegan2.py:36; sum_k w^2 for demodulation).  Inside a `derived_scope()` they are computed
        once and shared by every forward pass of that scope (the G loss evaluates G on unchanged weights
        several times); outside a scope nothing is cached."""
        root = self._root
        if root._derived_depth == 0 or root._templating or root.device.type == 'meta':
            return fn()
        k = (self._prefix, key, torch.is_grad_enabled())
        v = root._derived.get(k)
        if v is None:
            v = fn()
            root._derived[k] = v
        return v

    def derived_many(self, keys, fn_many):
        """`derived` for several keys at once: `fn_many(indices)` computes the missing ones together (one grouped
        launch for the sum-of-squares matrices of all modulated layers)."""
        root = self._root
        if root._derived_depth == 0 or root._templating or root.device.type == 'meta':
            return fn_many(list(range(len(keys))))
        ks = [(self._prefix, k, torch.is_grad_enabled()) for k in keys]
        missing = [i for i, k in enumerate(ks) if root._derived.get(k) is None]
        if missing:
            for i, v in zip(missing, fn_many(missing)):
                root._derived[ks[i]] = v
        return [root._derived[k] for k in ks]

    def derived_scope(self):
        """Context manager delimiting one training op: variables must not change inside it."""
        import contextlib
        root = self._root

        @contextlib.contextmanager
        def scope():
            if root._derived_depth == 0:
                root._derived.clear()
            root._derived_depth += 1
            try:
                yield
            finally:
                root._derived_depth -= 1
                if root._derived_depth == 0:
                    root._derived.clear()
        return scope()

    def invalidate_derived(self):
        self._root._derived.clear()

    def _materialize(self):
        """Allocate the flat buckets and initialise every variable (seeded NumPy stream, so all
        ranks / devices start bit-identical, like the reference's towers: optimizer.py:204-239)."""
        dev = self.device
        offsets = OrderedDict()
        total = 0
        for name, (shape, init, trainable) in self._specs.items():
            if trainable:
                n = int(np.prod(shape)) if len(shape) else 1
                offsets[name] = (total, n)
                total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self._offsets = offsets
        self._flat_size = total
        if dev.type == 'meta':
            self.flat_params = torch.empty((total,), device='meta')
            self.flat_grads = torch.empty((total,), device='meta')
            for name, (shape, init, trainable) in self._specs.items():
                self._store[name] = torch.empty(shape, device='meta')
            self._index()
            return
        rng = np.random.RandomState(self.seed)
        flat = np.zeros((total,), dtype=np.float32)
        others = OrderedDict()
        for name, (shape, init, trainable) in self._specs.items():
            kind = init[0]
            if kind == 'zeros':
                val = np.zeros(shape, dtype=np.float32)
            elif kind == 'normal':
                val = (rng.standard_normal(shape) * init[1]).astype(np.float32)
            elif kind == 'const':
                val = np.full(shape, init[1], dtype=np.float32)
            else:
                raise ValueError('unknown initializer %r' % (init,))
            if trainable:
                off, n = offsets[name]
                flat[off:off + n] = val.reshape(-1)
            else:
                others[name] = val
        self.flat_params = torch.from_numpy(flat).to(dev)
        self.flat_grads = torch.zeros_like(self.flat_params)
        for name, (shape, init, trainable) in self._specs.items():
            if trainable:
                off, n = offsets[name]
                p = self.flat_params[off:off + n].view(shape)
                p.requires_grad_(True)
                p.grad = self.flat_grads[off:off + n].view(shape)
                self._store[name] = p
            else:
                self._store[name] = torch.from_numpy(others[name]).to(dev)
        self._index()

    def _index(self):
        self.vars = OrderedDict(self._store)
        self.trainables = OrderedDict((n, v) for n, v in self._store.items() if self._specs[n][2])
        self._index_components(self)

    def _index_components(self, net):
        for comp in net.components.values():
            if isinstance(comp, Network):
                pre = comp._prefix
                comp.vars = OrderedDict((n[len(pre):], v) for n, v in self._store.items() if n.startswith(pre))
                comp.trainables = OrderedDict((n[len(pre):], v) for n, v in self._store.items()
                                              if n.startswith(pre) and self._specs[n][2])
                self._index_components(comp)

    # ------------------------------------------------------------------
    def get_output_for(self, *in_expr, return_as_list=False, **dynamic_kwargs):
        """Run the build function on the given tensors (network.py:199-232)."""
        assert len(in_expr) == self.num_inputs
        build_kwargs = dict(self.static_kwargs)
        build_kwargs.update(dynamic_kwargs)
        build_kwargs['is_template_graph'] = False
        build_kwargs['components'] = self.components
        with tfutil.variable_store(self):
            out = self._build_func(*in_expr, **build_kwargs)
        if isinstance(out, (tuple, list)):
            return list(out) if return_as_list else tuple(out)
        return [out] if return_as_list else out

    def requires_grad_(self, flag):
        """Freeze / unfreeze every trainable (the reference gets the same effect by registering
        gradients only w.r.t. one network's trainables, training_loop.py:288-291)."""
        assert self._root is self
        for v in self.trainables.values():
            v.requires_grad_(flag)
        self._derived.clear()     # cached derived tensors carry the old requires_grad state
        return self

    def get_var_local_name(self, var_or_global_name):
        """Local name of a variable given the tensor itself or its name (network.py:236-241)."""
        if torch.is_tensor(var_or_global_name):
            for name, v in self.vars.items():
                if v is var_or_global_name:
                    return name
            raise KeyError('tensor is not a variable of network %r' % self.name)
        name = str(var_or_global_name)
        prefix = self.name + '/'
        return name[len(prefix):] if name.startswith(prefix) else name

    def find_var(self, var_or_local_name):
        """The variable tensor for a local name (or the tensor itself) (network.py:243-246)."""
        return var_or_local_name if torch.is_tensor(var_or_local_name) else self.vars[self.get_var_local_name(var_or_local_name)]

    def get_var(self, var_or_local_name):
        """Value of a variable as a NumPy array (network.py:248-250 evaluates the variable)."""
        return self.find_var(var_or_local_name).detach().cpu().numpy()

    def set_var(self, var_or_local_name, new_value):
        """Overwrite a variable in place (network.py:252-255); cached weight-derived tensors are dropped."""
        var = self.find_var(var_or_local_name)
        with torch.no_grad():
            var.copy_(torch.as_tensor(np.asarray(new_value), dtype=var.dtype).to(var.device).reshape(var.shape))
        self.invalidate_derived()

    def _reset(self, names):
        """Re-run the initialisers of the given variables from the network's seed (network.py:223-233)."""
        rng = np.random.RandomState(self.seed)
        with torch.no_grad():
            for name, (shape, init, trainable) in self._specs.items():
                # draw for every variable so that a partial reset reproduces the values of a full one
                val = (rng.standard_normal(shape) * init[1]).astype(np.float32) if init[0] == 'normal' else \
                      np.full(shape, init[1] if init[0] == 'const' else 0.0, dtype=np.float32)
                if name in names:
                    self.vars[name].copy_(torch.from_numpy(val).to(self.vars[name].device).reshape(self.vars[name].shape))
        self.invalidate_derived()

    def reset_own_vars(self):
        self._reset(set(self.vars))      # components are part of this network's bucket: same as reset_vars here

    def reset_vars(self):
        self._reset(set(self.vars))

    def reset_trainables(self):
        self._reset(set(self.trainables))

    def num_params(self):
        return sum(int(np.prod(v.shape)) if v.dim() else 1 for v in self.trainables.values())

    def zero_grad(self):
        self.flat_grads.zero_()

    def copy_vars_from(self, src_net):
        src = src_net
        assert self._root is self and src._root is src
        same = list(self._offsets.items()) == list(src._offsets.items())
        self._derived.clear()
        with torch.no_grad():
            if same:
                self.flat_params.copy_(src.flat_params)
            for name, var in self.vars.items():
                if name in src.vars and (not same or not self._specs[name][2]):
                    var.copy_(src.vars[name])

    def copy_own_vars_from(self, src_net):
        """network.py:316-319 (components live in the same bucket here: all shared names are copied)."""
        self.copy_vars_from(src_net)

    def copy_trainables_from(self, src_net):
        """Copy the values of all trainables present in both networks (network.py:326-329)."""
        self._derived.clear()
        with torch.no_grad():
            for name, var in self.trainables.items():
                if name in src_net.trainables:
                    var.copy_(src_net.trainables[name])

    def clone(self, name=None, **new_static_kwargs):
        """network.py:301-314: same build function, same variable values."""
        static_kwargs = dict(self.static_kwargs)
        static_kwargs.update(new_static_kwargs)
        net = Network(name=name or self.name, func_name=self._build_func, device=self.device, seed=self.seed, **static_kwargs)
        net.copy_vars_from(self)
        return net

    def setup_as_moving_average_of(self, src_net, beta=0.99, beta_nontrainable=0.0):
        """Returns a callable that moves this network's variables towards `src_net`'s:
        var <- lerp(src, var, beta) for trainables, beta_nontrainable otherwise (network.py:341-351).
        `beta` may be a float or a zero-arg callable evaluated at each call."""
        assert self._root is self and src_net._root is src_net
        assert list(self._offsets) == list(src_net._offsets)

        def update_op():
            b = beta() if callable(beta) else beta
            self._derived.clear()
            with torch.no_grad():
                hip_ops.ema_raw(self.flat_params, src_net.flat_params, b)
                for name, var in self.vars.items():
                    if name in src_net.vars and not self._specs[name][2]:
                        if beta_nontrainable == 0.0:
                            var.copy_(src_net.vars[name])
                        else:
                            var.copy_(tfutil.lerp(src_net.vars[name], var, beta_nontrainable))
        return update_op

    def run(self, *in_arrays, minibatch_size=None, num_gpus=1, return_as_list=False, **dynamic_kwargs):
        """NumPy in, NumPy out, evaluated in minibatches without gradients (network.py:353-453)."""
        assert len(in_arrays) == self.num_inputs
        num_items = in_arrays[0].shape[0]
        if minibatch_size is None:
            minibatch_size = num_items
        outs = None
        with torch.no_grad():
            for begin in range(0, num_items, minibatch_size):
                end = min(begin + minibatch_size, num_items)
                ins = [torch.as_tensor(np.asarray(a[begin:end], dtype=np.float32)).to(self.device) for a in in_arrays]
                mb = self.get_output_for(*ins, return_as_list=True, **dynamic_kwargs)
                mb = [t.contiguous().cpu().numpy() for t in mb]
                if outs is None:
                    outs = [np.empty([num_items] + list(o.shape[1:]), dtype=o.dtype) for o in mb]
                for dst, o in zip(outs, mb):
                    dst[begin:end] = o
        if not return_as_list:
            outs = outs[0] if len(outs) == 1 else tuple(outs)
        return outs

    # ------------------------------------------------------------------
    # Pickle export / import in the layout of the reference's Network.__getstate__ / __setstate__ (network.py:255-299):
    #   version 4, name, static_kwargs, components {key: Network}, build_module_src, build_func_name,
    #   variables = [(local name, ndarray)] of the network's OWN variables (component variables travel with the component).
    def own_var_names(self):
        """Local names of the variables that belong to this network itself, not to one of its components (network.py:181-185)."""
        comp_prefixes = [c._prefix[len(self._prefix):] for c in self.components.values() if isinstance(c, Network)]
        return [n for n in self.vars if not any(n.startswith(cp) for cp in comp_prefixes)]

    def state_v4(self, build_module_src='', build_func_name=None):
        """The reference's pickle state for this network (recursively for components).  `build_module_src` is the source
        text of the reference module that defines the build function (training/networks_stylegan2.py of a reference
        checkout) when the pickle is meant to be opened BY the reference; '' when it only has to come back here."""
        fn = build_func_name or self._build_func_name.rsplit('.', 1)[-1]
        return dict(version=4, name=self.name, static_kwargs=dict(self.static_kwargs),
                    components={k: c for k, c in self.components.items() if isinstance(c, Network)},
                    build_module_src=build_module_src, build_func_name=fn,
                    variables=[(n, self.vars[n].detach().cpu().numpy().copy()) for n in self.own_var_names()])

    def __getstate__(self):
        return self.state_v4(build_module_src=getattr(self._root, '_export_module_src', ''))

    def __reduce__(self):
        return (_rebuild_network, (self.__getstate__(), str(self.device)))

    def load_state_v4(self, state, strict=True):
        """Copy the variables of a reference-layout state (this network's own + its components', recursively) into this
        network by local name.  strict: every variable of this network must be present with the same shape."""
        found = set()

        def walk(st, prefix):
            for local, value in st['variables']:
                found.add(prefix + local)
                if prefix + local in self.vars:
                    self.set_var(prefix + local, value)
                elif strict:
                    raise KeyError('pickled variable %r does not exist in network %r' % (prefix + local, self.name))
            for comp in dict(st.get('components', {})).values():
                cst = comp if isinstance(comp, dict) else comp.__dict__.get('_state') or comp.state_v4()
                walk(cst, prefix + cst['name'] + '/')
        walk(state, '')
        if strict:
            missing = [n for n in self.vars if n not in found]
            if missing:
                raise KeyError('network %r: variables missing from the pickle: %s' % (self.name, missing[:5]))
        return self

    def print_layers(self, title=None):
        rows = [[title or self.name, 'Params', 'Shape']]
        total = 0
        for name, v in self.trainables.items():
            n = int(np.prod(v.shape)) if v.dim() else 1
            total += n
            rows.append([name, str(n), str(tuple(v.shape))])
        rows.append(['Total', str(total), ''])
        widths = [max(len(r[i]) for r in rows) for i in range(3)]
        print()
        for r in rows:
            print('  '.join(c + ' ' * (w - len(c)) for c, w in zip(r, widths)))
        print()


class PickledNetwork:
    """What a reference pickle's `dnnlib.tflib.network.Network` object becomes when it is opened here (training/misc.py
    load_pkl): just its state dict.  `to_network(device)` builds the live Network."""

    def __setstate__(self, state):
        self._state = state

    def to_network(self, device=None, func_module='inclusivegan_amd.training.networks_stylegan2', **override_static_kwargs):
        return network_from_state(self._state, device=device, func_module=func_module, **override_static_kwargs)


def network_from_state(state, device=None, func_module='inclusivegan_amd.training.networks_stylegan2', **override_static_kwargs):
    """Reference-layout state (version 2-4) -> live Network on `device`: the build function is looked up BY NAME in this
    engine's counterpart module (the pickled source text is the reference's TensorFlow code and is not executed)."""
    assert state['version'] in (2, 3, 4)
    kw = dict(state['static_kwargs'])
    kw.update(override_static_kwargs)
    kw.pop('func_name', None)
    fn = state['build_func_name']
    func_name = fn if '.' in fn else func_module + '.' + fn
    if func_name.startswith('training.') or func_name.startswith('metrics.'):
        func_name = 'inclusivegan_amd.' + func_name
    net = Network(state['name'], func_name=func_name, device=device, **kw)
    net.load_state_v4(state)
    return net


def _rebuild_network(state, device):
    dev = device
    if dev.startswith('cuda') and not torch.cuda.is_available():
        dev = 'cpu'
    return network_from_state(state, device=dev)
