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
    [RCCL all-reduce of the bucket on the side stream] -> finite-check stream -> Adam stream,
three launches regardless of the number of variables.  The skip decision and the beta powers stay
on the device (no host sync; hipGraph-capturable).  Loss scaling / gradient accumulation
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

    def differentiate(self, loss, net):
        """The device work of register_gradients() for ONE registration, without Python-side
        bookkeeping: zero the bucket, backpropagate.  Safe to capture into a hipGraph; after each replay
        call `mark_registered(net)` and then `apply_updates()`."""
        self._bind(net)
        net.flat_grads.zero_()
        self._backprop_into_bucket(loss, net, accumulate=False)

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
    """Name kept for parity with optimizer.py:290; the arithmetic lives in csrc/optimizer.hip."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('use Optimizer; SimpleAdam arithmetic is fused into the flat-bucket HIP kernel')
