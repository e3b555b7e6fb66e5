"""`Optimizer`: gradient averaging across GPUs, non-finite-gradient skipping and Adam, with the
semantics of the reference's `dnnlib/tflib/optimizer.py:28-287` (+ `SimpleAdam` :290-336):

  * `register_gradients(loss, trainables)` -- differentiate the (already batch-meaned) loss w.r.t.
    one network's trainables (:114-154); several registrations before an update are summed and
    divided by their count (:169-186);
  * `apply_updates()` -- scale by 1/num_devices, sum across devices (:186,193-201), skip the whole
    update if any gradient is non-finite (:237), otherwise Adam with
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m, v, w -= lr_t*m/(sqrt(v)+eps) (:318-332);
  * `share=`: a regularisation optimizer shares the main optimizer's Adam slots and beta powers
    (:45,77-82,103-106; training_loop.py:252-255);
  * `reset_optimizer_state()` (:266-269).

MI355X design instead of per-variable NCCL/Adam ops: one process per GPU; the network's trainables
and gradients are single flat fp32 buckets (network.py), so an update is
    [RCCL all-reduce of the bucket] -> finite-check stream -> Adam stream,
a handful of launches regardless of the number of variables.  The skip decision and the beta powers stay
on the device (no host sync; hipGraph-capturable).

Gradient exchange overlapped with backward (`GradientExchange`): the bucket is cut into a few chunks of
consecutive variables; a tensor hook on every trainable places its (pre-scaled) gradient in the bucket the
moment autograd produces it, and when the last variable of a chunk has arrived that chunk's all-reduce is issued
asynchronously -- the process group runs it on its own stream, ordered after the producing kernels by an event,
while the backward pass carries on with the layers that are still to come.  (xGMI is point-to-point: a ring
all-reduce of the 94 MB bucket over 8 GPUs is ~1.1 ms of link time per step; D's 4x4..32x32 layers hold 90 % of
its parameters and finish first in backward, so their exchange hides completely behind the 64x64 / 128x128
layers; G's bulk finishes last and only its high-resolution tail overlaps.)  The waits are issued after
backward, still inside `differentiate()`, so under RCCL the whole exchange is part of the captured hipGraph.

Loss scaling / gradient accumulation
(`use_loss_scaling`, `minibatch_multiplier`) are fp16 / large-batch features the fp32 configs never
enable; they are not offered.
"""
import torch

from ... import hip_ops


def allreduce_mean_(flat_grads, num_registered=1, group=None):
    """Gradient averaging of optimizer.py:169-201 on the flat bucket, in place: scale by
    1 / (registrations * world size), then ONE all-reduce(sum) over the data-parallel group (the reference
    issues one nccl all_sum per variable).  Backend-agnostic (RCCL on GPUs, gloo in the CPU tests)."""
    world = 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
    scale = 1.0 / num_registered / world    # optimizer.py:186
    if scale != 1.0:
        flat_grads.mul_(scale)
    if world > 1:
        torch.distributed.all_reduce(flat_grads, op=torch.distributed.ReduceOp.SUM, group=group)  # :199
    return flat_grads


def _dist_world(group=None):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size(group)
    return 1


_capture_probe = {}


def _rccl_capture_probe(group):
    """Capture and replay a tiny all-reduce on this process group; True when the replayed result is right."""
    import os
    if os.environ.get('IGAN_GRAPH_COLLECTIVES', '1') == '0' or torch.distributed.get_backend(group) != 'nccl':
        return False
    try:
        t = torch.ones(1024, device=torch.device('cuda', torch.cuda.current_device()))
        torch.distributed.all_reduce(t, group=group)            # communicator set up outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            w = torch.distributed.all_reduce(t, group=group, async_op=True)
            w.wait()
        t.fill_(1.0)
        g.replay()
        torch.cuda.synchronize()
        return bool((t == float(_dist_world(group))).all())
    except Exception:   # noqa: BLE001 -- any failure means: keep collectives outside the graphs
        return False


def collectives_capturable(group=None, probe=None):
    """Can this process group's all-reduce be captured into a hipGraph?  RCCL: yes, verified once per group by capturing
    and replaying a tiny all-reduce (a failure here is cheap and leaves nothing behind; a failure in the middle of a
    training op's capture would not be).  gloo stages through the host: never.  IGAN_GRAPH_COLLECTIVES=0 forces 'no'.
    EVERY rank takes the same path: the ranks' answers are combined with an all-reduce(MIN), so one rank whose probe fails
    (`probe`: the per-rank check, injectable for the tests) takes every rank to the exchange-after-replay form."""
    if _dist_world(group) == 1:
        return True
    key = id(group)
    if key not in _capture_probe or probe is not None:
        ok = bool((probe or _rccl_capture_probe)(group))
        on_device = torch.cuda.is_available() and torch.distributed.get_backend(group) == 'nccl'
        flag = torch.tensor([1 if ok else 0], device=torch.device('cuda', torch.cuda.current_device()) if on_device else 'cpu')
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
        if probe is not None:
            return bool(flag.item())
        _capture_probe[key] = bool(flag.item())
    return _capture_probe[key]


class GradientExchange:
    """Chunked, asynchronous averaging of one network's flat gradient bucket across the data-parallel group, driven by
    tensor hooks during backward (see the module docstring).  Arithmetic = optimizer.py:186,199 of the reference: every
    gradient is multiplied by 1 / num_devices, then summed over the devices."""

    def __init__(self, net, group=None, num_chunks=4):
        self.net = net
        self.group = group
        self.world = _dist_world(group)
        names = [n for n, p in net.trainables.items() if p.requires_grad]
        params = [net.trainables[n] for n in names]
        base = net.flat_grads.data_ptr()
        offs = [(p.grad.data_ptr() - base) // 4 for p in params]
        order = sorted(range(len(params)), key=lambda i: offs[i])
        total = net.flat_grads.numel()
        # chunk boundaries at variable boundaries, ~equal element counts (16-byte aligned slots: network.py)
        bounds, acc = [0], 0
        for j, i in enumerate(order):
            acc = offs[i] + params[i].numel()
            if len(bounds) < num_chunks and acc >= total * len(bounds) / num_chunks and j + 1 < len(order):
                bounds.append(offs[order[j + 1]])
        bounds.append(total)
        self.chunks = [net.flat_grads[bounds[k]:bounds[k + 1]] for k in range(len(bounds) - 1)]
        self.chunk_of = {}
        self.chunk_size = [0] * len(self.chunks)
        for i in order:
            k = max(c for c in range(len(self.chunks)) if bounds[c] <= offs[i])
            self.chunk_of[id(params[i])] = k
            self.chunk_size[k] += 1
        self.active = False
        self.scale = 1.0
        self.pending = []
        self.works = []
        self.handles = [p.register_hook(self._make_hook(p)) for p in params]

    def _make_hook(self, p):
        def hook(g):
            if not self.active:
                return None
            view = p.grad
            with torch.no_grad():
                torch.mul(g.reshape(view.shape) if g.shape != view.shape else g, self.scale, out=view)    # placed + pre-scaled in one pass
            k = self.chunk_of[id(p)]
            self.pending[k] -= 1
            if self.pending[k] == 0 and self.world > 1:
                self.works.append(torch.distributed.all_reduce(self.chunks[k], op=torch.distributed.ReduceOp.SUM, group=self.group, async_op=True))
            return None
        return hook

    def begin(self, num_registered=1):
        self.scale = 1.0 / num_registered / self.world
        self.pending = list(self.chunk_size)
        self.works = []
        self.active = True

    def finish(self):
        """After backward: every chunk must have gone out (a variable that received no gradient keeps its zero and its
        chunk is sent now); then make the current stream wait for the collectives."""
        self.active = False
        for k, left in enumerate(self.pending):
            if left > 0 and self.world > 1:
                self.works.append(torch.distributed.all_reduce(self.chunks[k], op=torch.distributed.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self.works:
            w.wait()
        self.works = []


class Optimizer:
    def __init__(self, name='Train', learning_rate=0.001, share=None, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 minibatch_multiplier=None, use_loss_scaling=False, process_group=None, **kwargs):
        if use_loss_scaling:
            raise NotImplementedError('Optimizer: dynamic loss scaling is an fp16 feature; this engine is fp32')
        if kwargs:
            raise TypeError('Optimizer: unsupported arguments %s' % sorted(kwargs))
        self.name = name
        self.learning_rate = learning_rate      # float or zero-arg callable (lrate_in placeholder)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.minibatch_multiplier = minibatch_multiplier
        self.process_group = process_group
        self._net = None
        self._num_registered = 0
        self._exchange = None       # GradientExchange of the bound network when the exchange rides inside differentiate()
        self._exchanged = False
        if share is not None:
            assert isinstance(share, Optimizer)
            assert (self.beta1, self.beta2, self.epsilon) == (share.beta1, share.beta2, share.epsilon)
            self._state = share._state          # shared Adam slots (optimizer.py:77-82)
        else:
            self._state = {}

    # ------------------------------------------------------------------
    def _bind(self, net):
        if self._net is None:
            self._net = net
        assert self._net is net, 'Optimizer %s is bound to network %s' % (self.name, self._net.name)
        st = self._state
        if 'm' not in st:
            st['m'] = torch.zeros_like(net.flat_params)
            st['v'] = torch.zeros_like(net.flat_params)
            st['pow'] = torch.ones((2,), device=net.flat_params.device, dtype=torch.float32)
            st['flag'] = torch.zeros((1,), device=net.flat_params.device, dtype=torch.int32)
            st['overflows'] = torch.zeros((1,), device=net.flat_params.device, dtype=torch.int64)

    @staticmethod
    def _backprop_into_bucket(loss, net, accumulate):
        """d loss / d trainables written (or added) into the flat gradient bucket.  torch.autograd.grad
        hands the gradients back instead of running one AccumulateGrad add per variable; a multi-tensor
        copy then places them in the bucket views (a few launches for the whole network)."""
        params = [p for p in net.trainables.values() if p.requires_grad]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        dst = [p.grad for p, g in zip(params, grads) if g is not None]
        src = [g for g in grads if g is not None]
        if not dst:
            return
        with torch.no_grad():
            if accumulate:
                torch._foreach_add_(dst, src)
            else:
                torch._foreach_copy_(dst, src)

    def register_gradients(self, loss, net):
        """Accumulate d loss / d trainables into the network's gradient bucket.
        `net` is the Network whose trainables are optimised (the reference passes `G_gpu.trainables`)."""
        self._bind(net)
        if self._num_registered == 0:
            net.flat_grads.zero_()
        self._backprop_into_bucket(loss, net, accumulate=True)
        self._num_registered += 1

    def differentiate(self, loss, net, overlap_exchange=None):
        """The device work of register_gradients() for ONE registration, without Python-side
        bookkeeping: zero the bucket, backpropagate.  Safe to capture into a hipGraph; after each replay
        call `mark_registered(net)` and then `apply_updates()`.

        With more than one rank (and `overlap_exchange` not False) the cross-device averaging happens in here too,
        chunk by chunk while backward is still running (GradientExchange); `apply_updates()` then finds the bucket
        already averaged.  Pass overlap_exchange=False to keep the exchange in apply_updates() (one blocking all-reduce
        after backward: the form used when the collective cannot be part of a captured graph)."""
        self._bind(net)
        net.flat_grads.zero_()
        world = _dist_world(self.process_group)
        if overlap_exchange is None:
            overlap_exchange = world > 1
        if not overlap_exchange:
            self._backprop_into_bucket(loss, net, accumulate=False)
            self._exchanged = False
            return
        ex = self._exchange_for(net)
        ex.begin(num_registered=1)
        params = [p for p in net.trainables.values() if p.requires_grad]
        torch.autograd.grad(loss, params, allow_unused=True)      # the hooks place, scale and send the gradients
        ex.finish()
        self._exchanged = True

    def _exchange_for(self, net):
        st = self._state
        if 'exchange' not in st:        # shared with the regularisation optimizer, like the Adam slots
            st['exchange'] = GradientExchange(net, self.process_group)
        return st['exchange']

    def mark_registered(self, net, count=1):
        self._bind(net)
        self._num_registered = count

    def apply_updates(self, allow_no_op=False):
        if self._num_registered == 0:
            if allow_no_op:
                return
            raise RuntimeError('Optimizer.apply_updates() without registered gradients')
        net = self._net
        st = self._state
        if self._exchanged:
            assert self._num_registered == 1
            g = net.flat_grads           # averaged inside differentiate()
            self._exchanged = False
        else:
            g = allreduce_mean_(net.flat_grads, self._num_registered, self.process_group)
        lr = self.learning_rate() if callable(self.learning_rate) else self.learning_rate
        with torch.no_grad():
            st['flag'].zero_()
            hip_ops.finite_check_raw(g, st['flag'])                                          # :237
            hip_ops.adam_step_raw(net.flat_params, g, st['m'], st['v'], lr, self.beta1, self.beta2, self.epsilon,
                                  st['pow'], st['flag'])                                     # :318-332
            st['overflows'] += st['flag'].to(torch.int64)                                    # overflow_frequency (:251)
        net.invalidate_derived()        # the variables changed: drop cached w * coef etc.
        self._num_registered = 0

    def reset_optimizer_state(self):
        st = self._state
        if 'm' in st:
            st['m'].zero_(); st['v'].zero_(); st['pow'].fill_(1.0)

    def overflow_count(self):
        return int(self._state['overflows'].item()) if 'overflows' in self._state else 0


class SimpleAdam:
    """The reference's stand-alone Adam (optimizer.py:290-336: "behaves identically" to tf.train.AdamOptimizer under
    tflib.Optimizer) for loose lists of tensors: `compute_gradients(loss, var_list)` -> [(grad, var)],
    `apply_gradients(grads_and_vars)` updates the variables in place.  One beta-power pair per apply_gradients call set
    (:311-317), one (m, v) pair per variable (:321-324), lr_new = lr * sqrt(1 - b2^t) / (1 - b1^t) (:317), all through the
    flat Adam kernel of csrc/optimizer.hip (no overflow gate here: the gate belongs to Optimizer.apply_updates, :237)."""

    def __init__(self, name='Adam', learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.name = name
        self.learning_rate = learning_rate
        self.beta1 = beta1
        self.beta2 = beta2
        self.epsilon = epsilon
        self.all_state_vars = []
        self._slots = {}

    def variables(self):
        return self.all_state_vars

    def compute_gradients(self, loss, var_list):
        return list(zip(torch.autograd.grad(loss, var_list, allow_unused=True), var_list))

    def apply_gradients(self, grads_and_vars):
        lr = self.learning_rate() if callable(self.learning_rate) else self.learning_rate
        key = tuple(id(v) for _, v in grads_and_vars)
        if key not in self._slots:
            dev = grads_and_vars[0][1].device
            slot = dict(pow=torch.ones((2,), device=dev), flag=torch.zeros((1,), device=dev, dtype=torch.int32),
                        mv=[(torch.zeros(v.numel(), device=dev), torch.zeros(v.numel(), device=dev)) for _, v in grads_and_vars])
            self._slots[key] = slot
            self.all_state_vars += [slot['pow']] + [t for mv in slot['mv'] for t in mv]
        slot = self._slots[key]
        with torch.no_grad():
            pow_before = slot['pow'].clone()
            for (g, v), (m, vv) in zip(grads_and_vars, slot['mv']):
                if g is None:
                    continue
                if not v.is_contiguous():
                    raise ValueError('SimpleAdam: variables must be contiguous')
                slot['pow'].copy_(pow_before)           # every variable of the set sees the same power pair (:317)
                hip_ops.adam_step_raw(v.detach().reshape(-1), g.contiguous().reshape(-1), m, vv, lr, self.beta1, self.beta2, self.epsilon,
                                      slot['pow'], slot['flag'])
