"""Main training loop on the HIP path.  `training_loop(**kwargs)` accepts the kwargs of the
reference's `training/training_loop.py:123-160` (as assembled by run_training.py:36-165) and runs
the same host loop (:332-482):

    every data_size*init_staleness images: IMLE refresh (generate candidates with G, assign every
        real its nearest candidate) (:353-406)
    assemble 2*minibatch (real, label, matched latent) triples in dataset order, optional attribute
        AND-mask, slerp-perturb the latents, split in halves, shuffle each (:409-464)
    G step; G reg every G_reg_interval; D step + Gs EMA; D reg every D_reg_interval (:466-479)
    cur_nimg += 2*minibatch (:481)

MI355X execution model (instead of one process building an in-graph tower per GPU, :257-291):
one process per GPU (`num_gpus` = torch.distributed world size), every rank runs the same host
logic from the same NumPy seed (run_training.py:61 `rnd.np_random_seed`) and takes its slice of
the global minibatch (tf.split, :231-239); gradients are averaged with one RCCL all-reduce per
step over the flat bucket (tflib/optimizer.py); the IMLE candidates are sharded over ranks and the
per-real minima (fp64 distance, index) are combined with two all-reduce(min).

With `run_dir` set, rank 0 writes the reference's snapshots at its cadence (:165-166,506-519): image grids of Gs on fixed
latents, the reconstruction pairs of the IMLE term, and (G, D, Gs) pickles in the reference's own layout (training/misc.py);
`resume_pkl` continues from such a pickle -- or from one the reference wrote; the tick lines are teed into run_dir/log.txt (the
file resume reads its kimg / time from, misc.py:147-162), `metric_arg_list` is evaluated on every network snapshot (:519) and a
final snapshot is written (:527-530).  tfevents stay out.  A `hooks` dict lets a driver (bench.py, tests) observe iterations and stop early.
"""
import functools
import os
import time

import numpy as np
import torch

from .. import dnnlib
from ..dnnlib import tflib
from ..dnnlib.tflib import tfutil
from ..dnnlib.tflib.autosummary import autosummary
from ..dnnlib.tflib import autosummary as autosummary_mod
from ..dnnlib.tflib import graphs
from . import dataset
from . import misc
from ..dci_code.dci import DCI, unpack_best
from .. import hip_ops
from . import imle

#----------------------------------------------------------------------------
# Function to determine the dimension of random projection (:28-35).

def func_proj_dim(init_proj_dim, data_size, num_samples_factor, G):
    if init_proj_dim is None:
        proj_dim = int(np.prod(G.output_shape[1:]))
    elif init_proj_dim == 0:
        # sklearn.random_projection.johnson_lindenstrauss_min_dim(n_samples, eps=0.1)
        n, eps = data_size * num_samples_factor, 0.1
        proj_dim = int(4 * np.log(n) / ((eps ** 2 / 2) - (eps ** 3 / 3)))
    else:
        proj_dim = init_proj_dim
    return proj_dim

#----------------------------------------------------------------------------
# Just-in-time processing of training images before feeding them to the networks (:40-60).

def process_reals(x, labels, lod, mirror_augment, drange_data, drange_net):
    x = x.to(torch.float32)
    x = misc.adjust_dynamic_range(x, drange_data, drange_net)
    if mirror_augment:
        flip = tfutil.random_uniform([x.shape[0]], x.device) >= 0.5
        x = torch.where(flip[:, None, None, None], x.flip(3), x)
    lod = float(lod)
    if lod != 0:
        # FadeLOD (:50-57): cross-fade towards the 2x2 box-filtered image by the fractional part; UpscaleLOD (:58-59): nearest-neighbour
        # upscale by 2^floor(lod).  Both are the identity at lod == 0, the only value config-e/f produce (:93-94), so this is torch glue.
        n, c, h, w = x.shape
        y = x.reshape(n, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5), keepdim=True).expand(n, c, h // 2, 2, w // 2, 2).reshape(n, c, h, w)
        x = tfutil.lerp(x, y, lod - np.floor(lod))
        factor = int(2 ** np.floor(lod))
        if factor != 1:
            x = x.reshape(n, c, h, 1, w, 1).expand(n, c, h, factor, w, factor).reshape(n, c, h * factor, w * factor)
    return x.contiguous(memory_format=torch.channels_last), labels

def lazy_regularization_args(opt_args, reg_interval, lazy_regularization):
    """:244-251: with lazy regularisation the main and the regularisation optimizer of a network both run with the learning rate
    times mb_ratio = interval / (interval + 1) and with beta1, beta2 raised to that power.  -> (learning-rate ratio, adjusted args)."""
    args = dict(opt_args)
    mb_ratio = reg_interval / (reg_interval + 1) if lazy_regularization else 1.0
    if lazy_regularization:
        if 'beta1' in args: args['beta1'] **= mb_ratio
        if 'beta2' in args: args['beta2'] **= mb_ratio
    return mb_ratio, args

def smoothing_beta(minibatch_size, G_smoothing_kimg):
    """:222: Gs_beta = 0.5 ** (minibatch_size / (G_smoothing_kimg * 1000)), 0 when the smoothing is switched off."""
    return 0.5 ** (minibatch_size / (G_smoothing_kimg * 1000.0)) if G_smoothing_kimg > 0.0 else 0.0

#----------------------------------------------------------------------------
# Evaluate time-varying training parameters (:65-118).

def training_schedule(
    cur_nimg,
    training_set,
    lod_initial_resolution  = None,
    lod_training_kimg       = 600,
    lod_transition_kimg     = 600,
    minibatch_size_base     = 64,
    minibatch_size_dict     = {},
    minibatch_gpu_base      = 32,
    minibatch_gpu_dict      = {},
    G_lrate_base            = 0.002,
    G_lrate_dict            = {},
    D_lrate_base            = 0.002,
    D_lrate_dict            = {},
    lrate_rampup_kimg       = 0,
    tick_kimg_base          = 1,
    tick_kimg_dict          = {}):

    s = dnnlib.EasyDict()
    s.kimg = cur_nimg / 1000.0
    if lod_initial_resolution is not None:
        raise NotImplementedError('progressive growing (configs a-d) is not on the hot path')
    s.lod = 0.0
    s.resolution = 2 ** training_set.resolution_log2
    s.minibatch_size = minibatch_size_dict.get(s.resolution, minibatch_size_base)
    s.minibatch_gpu = minibatch_gpu_dict.get(s.resolution, minibatch_gpu_base)
    s.G_lrate = G_lrate_dict.get(s.resolution, G_lrate_base)
    s.D_lrate = D_lrate_dict.get(s.resolution, D_lrate_base)
    if lrate_rampup_kimg > 0:
        rampup = min(s.kimg / lrate_rampup_kimg, 1.0)
        s.G_lrate *= rampup
        s.D_lrate *= rampup
    s.tick_kimg = tick_kimg_dict.get(s.resolution, tick_kimg_base)
    return s

#----------------------------------------------------------------------------

def _dist_info():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def combine_best_(best_d2, best_idx, world):
    """Merge the per-rank running minima (fp64 squared distance, int32 candidate index) into the lexicographic minimum
    over all ranks' candidate shards: all-reduce(min) of the distances, then all-reduce(min) of the indices of the ranks
    that hold that distance (the others propose INT32_MAX) -- ties go to the lowest index, as in a single search."""
    if world > 1:
        mine = best_d2.clone()
        torch.distributed.all_reduce(best_d2, op=torch.distributed.ReduceOp.MIN)
        proposal = torch.where(mine == best_d2, best_idx, torch.full_like(best_idx, 2 ** 31 - 1))
        torch.distributed.all_reduce(proposal, op=torch.distributed.ReduceOp.MIN)
        best_idx.copy_(proposal)
    return best_d2, best_idx


def exclusive_assignment(knn_idx, knn_dist):
    """training_loop.py:382-396: walk the reals in data-set order; each takes its nearest candidate that no earlier real has
    taken among its k nearest, or its very nearest when all k are taken.  knn_idx / knn_dist: [data_size, k] ascending."""
    taken = set()
    idx_out = np.empty(knn_idx.shape[0], dtype=np.int64)
    dist_out = np.empty(knn_idx.shape[0], dtype=np.float64)
    for i in range(knn_idx.shape[0]):
        pick = 0
        for j in range(knn_idx.shape[1]):
            if int(knn_idx[i, j]) not in taken:
                pick = j
                break
        idx_out[i] = knn_idx[i, pick]
        dist_out[i] = knn_dist[i, pick]
        taken.add(int(knn_idx[i, pick]))
    return idx_out, dist_out


def imle_refresh(G, training_set_rec, latent_candidates, label_candidates, data_size, minibatch_size, candidate_batch_size,
                 drange_net, device, rank=0, world=1, query_chunk=8192, infer_minibatch=None, projector=None, exclusive_k=0, cand_pack=4096):
    """IMLE assignment (:357-406, non-exclusive): every real image (dataset order, [-1,1] range, flattened CHW;
    times `projector` [C*H*W, proj_dim] when random projection is on, :377-380) gets the index of its nearest generated
    candidate and the Euclidean distance.  Candidates are generated batch by batch with G (training weights, validation
    mode, random noise -- `G.run(..., is_validation=True)`, :361) and folded, a pack of `cand_pack` at a time, into a running
    per-real minimum; rank r handles packs r, r+world, ... and the minima are combined across ranks.  The reals are pulled from
    `training_set_rec` exactly as the reference's query loop does (2 * minibatch_size per call, data_size in all, :374-403).
    Returns (nearest_indices int64 [data_size], dists float64 [data_size]) as NumPy."""
    num_cand = latent_candidates.shape[0]
    dim = int(np.prod(training_set_rec.shape))
    pdim = dim if projector is None else int(projector.shape[1])
    # Reals resident on the device in fp32 (30 000 x 49 152 x 4 B = 5.9 GB for CelebA-128).
    reals = torch.empty((data_size, pdim), device=device, dtype=torch.float32)
    mb2 = minibatch_size * 2
    assert data_size % mb2 == 0
    per_upload = max(1, (256 << 20) // (4 * dim * mb2))        # query batches per host->device copy
    i = 0
    while i < data_size:
        nb = min(per_upload, (data_size - i) // mb2)
        host = np.concatenate([training_set_rec.get_minibatch_np(mb2)[0] for _ in range(nb)], axis=0)
        r = torch.from_numpy(host).to(device).to(torch.float32)
        r = misc.adjust_dynamic_range(r, training_set_rec.dynamic_range, drange_net).reshape(r.shape[0], -1)
        reals[i:i + r.shape[0]] = r if projector is None else hip_ops.matmul(r, projector)
        i += r.shape[0]
    rnorm = hip_ops.row_sqnorm_raw(reals)
    best_d2, best_idx = hip_ops.nn1_state(data_size, device)
    resident = [] if exclusive_k > 1 else None      # exclusive assignment: this rank's candidates stay on the device (59 GB for CelebA-128)
    # Candidates are generated `candidate_batch_size` at a time like the reference (:359-366) but searched in PACKS of several
    # batches: the distance GEMM of one 256-candidate batch against a 4096-query chunk is 64 tiles -- a quarter of the
    # device -- and ran at 31 TFLOP/s; a [query_chunk x dim] x [dim x pack] product fills it (profiles/r03_refresh_*.txt).
    # The assignment is the exact minimum over all candidates, so it does not depend on how they are grouped.
    pack = max(candidate_batch_size, (int(cand_pack) // candidate_batch_size) * candidate_batch_size)
    npacks = (num_cand + pack - 1) // pack
    # Inference batch: the reference feeds sched.minibatch_size at a time (G.run(..., minibatch_size=), :361); the images do not
    # depend on how the candidates are batched, so use a batch that fills the MFMA tiles (bounded so that the largest
    # intermediate, [n, C, 2R+1, 2R+1], stays under 2 GiB: the kernels address with 32-bit byte offsets).
    per_img = 4 * max(training_set_rec.shape[0], 128) * (training_set_rec.shape[1] * 2 + 1) ** 2 // 4
    infer_batch = infer_minibatch or int(max(minibatch_size, min(candidate_batch_size, 64, (1 << 30) // max(per_img, 1))))
    pack_buf = None
    with torch.no_grad():
        for pk in range(rank, npacks, world):
            c0 = pk * pack
            n = min(pack, num_cand - c0)
            if pack_buf is None or resident is not None:
                pack_buf = torch.empty((min(pack, num_cand), pdim), device=device, dtype=torch.float32)
            cand = pack_buf[:n]
            z = torch.from_numpy(latent_candidates[c0:c0 + n]).to(device)
            lab = torch.from_numpy(np.ascontiguousarray(label_candidates[c0:c0 + n], dtype=np.float32)).to(device)
            for j in range(0, n, infer_batch):
                img = G.get_output_for(z[j:j + infer_batch], lab[j:j + infer_batch], is_validation=True)
                flat = img.contiguous().reshape(img.shape[0], -1)                     # logical NCHW flatten (:363)
                if projector is not None:
                    flat = hip_ops.matmul(flat, projector)                            # :365
                cand[j:j + flat.shape[0]].copy_(flat)
            if not bool(torch.isfinite(cand).all()):
                raise FloatingPointError('IMLE refresh: the generator produced non-finite candidate images (candidates %d..%d)' % (c0, c0 + n))
            if resident is not None:
                resident.append((c0, cand))
                continue
            cnorm = hip_ops.row_sqnorm_raw(cand)
            for q0 in range(0, data_size, query_chunk):
                hip_ops.nn1_update_raw(reals[q0:q0 + query_chunk], rnorm[q0:q0 + query_chunk], cand, cnorm,
                                       best_d2[q0:q0 + query_chunk], best_idx[q0:q0 + query_chunk], c0)
    if resident is not None:
        # k nearest candidates of every real over this rank's shard, merged over the ranks, then the reference's greedy
        # exclusive pick on the host (identical on every rank: same inputs)
        db = DCI(pdim, device=device)
        base = torch.cat([torch.arange(c0, c0 + c.shape[0], device=device) for c0, c in resident]) if resident else torch.zeros(0, dtype=torch.int64, device=device)
        k = int(exclusive_k)
        if resident:
            db._data = torch.cat([c for _, c in resident], dim=0)
            db._norms = hip_ops.row_sqnorm_raw(db._data)
            li, ld = db.query_device_k(reals, min(k, db.num_points))
            gi = base[li]
        else:
            gi = torch.zeros((data_size, 0), dtype=torch.int64, device=device); ld = torch.zeros((data_size, 0), dtype=torch.float64, device=device)
        if gi.shape[1] < k:     # a shard smaller than k: pad with "no candidate"
            pad = k - gi.shape[1]
            gi = torch.cat([gi, torch.full((data_size, pad), 2 ** 31 - 1, dtype=torch.int64, device=device)], dim=1)
            ld = torch.cat([ld, torch.full((data_size, pad), float('inf'), dtype=torch.float64, device=device)], dim=1)
        if world > 1:
            all_i = [torch.empty_like(gi) for _ in range(world)]; all_d = [torch.empty_like(ld) for _ in range(world)]
            torch.distributed.all_gather(all_i, gi); torch.distributed.all_gather(all_d, ld)
            gi, ld = torch.cat(all_i, dim=1), torch.cat(all_d, dim=1)
            o = torch.argsort(gi, dim=1, stable=True); gi, ld = torch.gather(gi, 1, o), torch.gather(ld, 1, o)
            o = torch.argsort(ld, dim=1, stable=True); gi, ld = torch.gather(gi, 1, o)[:, :k], torch.gather(ld, 1, o)[:, :k]
        return exclusive_assignment(gi.cpu().numpy(), ld.cpu().numpy())
    combine_best_(best_d2, best_idx, world)
    idx, dist = unpack_best(best_d2, best_idx)
    return idx.cpu().numpy(), dist.cpu().numpy().astype(np.float64)

#----------------------------------------------------------------------------
# Main training script.

def _with_random_source(fn):
    """hooks['random_source'] (a tflib.tfutil source, e.g. TapRandom) supplies every device-side random draw of the run --
    the parity tests read the draws of the captured graphs back through it."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        src = (kwargs.get('hooks') or {}).get('random_source')
        if src is None:
            return fn(*args, **kwargs)
        with tfutil.use_random(src):
            return fn(*args, **kwargs)
    return wrapper


_f16_window_seen = [0, 0]


def _f16_window_warning(bound=1e-4):
    """The two-piece fp16 form holds an operand to 2^-23 inside a window of 2^26 below the largest of its scale group.  In the row images of
    the forward / data-gradient kernel the group is a pixel's channel vector and elements below the window should be chance near-zeros (1e-6 on random-init
    networks); a fraction above `bound` since the last call says the activations of single pixels span more than 2^26 -- worth a line in the log.  (Column images
    are not judged: their group runs along the summed axis.)  None when there is nothing to say."""
    import ctypes
    from .. import _abi
    lib = _abi.get_plugin()
    if lib.igan_conv_piece_form() != 2:
        return None
    v = (ctypes.c_ulonglong * 4)()
    if lib.igan_debug_f16_window_by_kind(v, 0) != 0:        # the counters are left alone (bench.py reports their totals): the difference to the last call is judged
        return None
    below, imaged = v[0] - _f16_window_seen[0], v[1] - _f16_window_seen[1]
    _f16_window_seen[0], _f16_window_seen[1] = v[0], v[1]
    if imaged <= 0 or below < 0:
        return None
    frac = below / imaged
    if frac <= bound:
        return None
    return ('WARNING: fp16 convolution form: %.2e of the row-image elements since the last tick lay more than 2^26 below their own pixel\'s largest channel '
            '(bound %.0e); IGAN_CONV_PLANES=1 runs the exact three-piece bf16 form' % (frac, bound))


class SubmitThread:
    """A second host thread that owns the per-iteration device submissions (input copies, graph replays, optimizer updates), fed in
    order through a queue.  With the HIP runtime's graph packet capture off (inclusivegan_amd/__init__.py) `CUDAGraph.replay()` blocks its
    caller for about as long as the device executes the graph (the runtime hands over the kernel nodes as the queue drains), so a single
    host thread alternates between "inside replay()" and "preparing the next iteration" -- and the device idles during the second
    (profiles/r03_bench_steady_state_bf16_pieces_variant.txt: 94 % busy; host profile: 12.2 of 13.4 s inside replay()).  replay() releases
    the GIL, so the main thread assembles iteration i + 1 (NumPy sampler, pinned staging) while this thread submits iteration i.
    Everything device-side stays on ONE stream in program order: the arithmetic and every buffer hand-over are those of the single-thread
    loop (tests/test_gpu_loop_parity.py::test_async_submission_equals_synchronous_loop: bit-identical).
    MEASURED AND NOT THE DEFAULT (round 4, same-box A/B over 96 iterations): 237 / 269 img/s against 262 / 291 with
    CPython's 5 ms GIL switch interval, 273 / 279 against 279 / 287 with 0.1 ms -- the second thread's GIL hand-overs between replays cost
    more than the ~3 ms of host work they hide.  IGAN_ASYNC_SUBMIT=1 switches it on."""

    def __init__(self, device):
        import queue
        import threading
        self.q = queue.Queue(maxsize=64)
        self.error = None
        self.device = device
        self.cond = threading.Condition()
        self.submitted = 0
        self.completed = 0
        # the submission thread re-takes the GIL after every replay(): with CPython's default 5 ms switch interval the main thread's
        # NumPy work would hold it off for longer than the work it is meant to hide
        import sys
        self._switch = sys.getswitchinterval()
        sys.setswitchinterval(float(os.environ.get('IGAN_ASYNC_SWITCH_S', '1e-4')))
        self.thread = threading.Thread(target=self._run, name='igan-submit', daemon=True)
        self.thread.start()
        self.closed = False
        SubmitThread.live.add(self)

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            fn = self.q.get()
            try:
                if fn is None:
                    return
                if self.error is None:
                    fn()
            except BaseException as e:      # noqa: BLE001 -- handed to the main thread at its next submit() / drain()
                self.error = e
            finally:
                self.q.task_done()
                with self.cond:
                    self.completed += 1
                    self.cond.notify_all()

    def _check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def submit(self, fn):
        self._check()
        with self.cond:
            self.submitted += 1
        self.q.put(fn)

    def wait_backlog(self, n):
        """Returns when at most n submitted items have not been run yet (bounds how far the main thread runs ahead)."""
        with self.cond:
            self.cond.wait_for(lambda: self.submitted - self.completed <= n or self.error is not None)
        self._check()

    def drain(self):
        """Returns when everything submitted so far has been HANDED to the device (not: executed)."""
        self.q.join()
        self._check()

    live = set()        # submitters that have not been closed: training_loop closes them on EVERY way out (an exception used to leave the thread and the switch interval behind)

    def close(self, check=True, timeout=None):
        """timeout (seconds): give up waiting for the worker (a replay that never returns: a hung device) instead of blocking the caller's own
        exception for ever; the daemon thread is then left behind and a RuntimeWarning says so."""
        import sys
        if self.closed:
            return
        self.closed = True
        SubmitThread.live.discard(self)
        self.q.put(None)
        self.thread.join(timeout)
        sys.setswitchinterval(self._switch)
        if self.thread.is_alive():
            import warnings
            warnings.warn('inclusivegan_amd: the submission thread did not exit within %.0f s (a device submission is still blocked); left behind as a daemon thread' % timeout, RuntimeWarning)
            return
        if check:
            self._check()


def _closing_submitters(fn):
    """Whatever way training_loop() ends, the submission threads THIS call started and its altered GIL switch interval do not outlive it (submitters of
    another loop running in the process are not touched: ADVICE r05).  On the way out of an exception the wait for the worker is bounded, so that a hung
    replay cannot swallow the exception that reports it."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        before = set(SubmitThread.live)
        host_threads = torch.get_num_threads()      # the loop caps PyTorch's CPU pool while it runs (hostaffinity.limit_host_threads, IGAN_HOST_THREADS); a library caller gets its setting back
        failed = True
        try:
            out = fn(*args, **kwargs)
            failed = False
            return out
        finally:
            for sub in list(SubmitThread.live - before):
                sub.close(check=False, timeout=30.0 if failed else None)
            if torch.get_num_threads() != host_threads:
                torch.set_num_threads(host_threads)
    return wrapper


@_with_random_source
@_closing_submitters
def training_loop(
    G_args                  = {},
    D_args                  = {},
    G_opt_args              = {},
    D_opt_args              = {},
    G_loss_args             = {},
    D_loss_args             = {},
    dataset_args            = {},
    sched_args              = {},
    grid_args               = {},
    metric_arg_list         = [],
    tf_config               = {},
    data_dir                = None,
    G_smoothing_kimg        = 10.0,
    minibatch_repeats       = 4,
    lazy_regularization     = True,
    G_reg_interval          = 4,
    D_reg_interval          = 16,
    reset_opt_for_new_lod   = True,
    total_kimg              = 25000,
    mirror_augment          = False,
    drange_net              = [-1,1],
    save_tf_graph           = False,
    save_weight_histograms  = False,
    resume_pkl              = None,

    data_size               = 3000,
    num_epochs              = 10000,

    init_proj_dim           = None,
    init_staleness          = 10,
    num_samples_factor      = 25,
    knn_perturb_factor      = 0.1,
    candidate_batch_size    = 256,
    exclusive_retrieved_code = 0,
    dist_thres_percentile   = 100.0,
    attr_interesting        = None,

    # --- extensions (not in the reference) ---
    attr_names              = None,     # list of attribute names (reference reads celeba/Anno/list_attr_celeba.txt, :174-180)
    lpips_func_name         = 'inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual',
    hooks                   = None,     # {'on_iteration': f(state) -> bool stop, 'on_refresh': f(seconds), 'on_batch': f(host batch dict),
                                        #  'on_start': f(state) before the first iteration, 'on_op': f(step name, output, feed) after each training op,
                                        #  'on_assignment': f(nearest indices, distances) after every IMLE refresh,
                                        #  'random_source': tfutil source for all device draws}
    hip_graphs              = True,     # capture the four training ops into hipGraphs (env IGAN_HIP_GRAPHS=0 disables)
    run_dir                 = None,     # where snapshots go (arb-reals.png, arb-fakes-*.png, rec-*.png, network-snapshot-*.pkl, :171-172,506-519); None = none
    reference_src           = None,     # path of a reference checkout's training/networks_stylegan2.py: snapshots then open in the reference too
    submit_config           = None,
    ):

    hooks = hooks or {}
    from .. import hostaffinity
    hostaffinity.limit_host_threads()       # every entry into the loop, not only run_training / bench.py: PyTorch's CPU pool sized to the affinity mask stalls the submissions; for the duration of the call only (the wrapper restores the caller's pool size)

    # Initialize (tflib.init_tf: rnd.np_random_seed, tfutil.py:122-147).
    rank, world = _dist_info()
    num_gpus = world
    np_seed = tf_config.get('rnd.np_random_seed', 1000) if tf_config else 1000
    np.random.seed(np_seed)                 # identical host-side stream on every rank
    device = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(np_seed * 7919 + rank)   # per-rank device stream (latents, noise)

    # Load training set (:169-170).
    ds_args = dict(dataset_args)
    ds_args.setdefault('data_size', data_size)
    training_set = dataset.load_dataset(data_dir=data_dir, verbose=(rank == 0), device=device, rank=rank, world_size=world, **ds_args)
    training_set_rec = dataset.load_dataset(data_dir=data_dir, verbose=False, device=device, rank=0, world_size=1, **ds_args)

    grid_size, grid_reals, grid_labels = misc.setup_snapshot_image_grid(training_set, **grid_args)   # :171 (walks training_set's iterator)
    image_snapshot_ticks = data_size // 1000            # :165
    network_snapshot_ticks = data_size // 1000 * 5      # :166
    tick_reals_old = None
    module_src = ''
    if run_dir is not None and rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        misc.save_image_grid(grid_reals, os.path.join(run_dir, 'arb-reals.png'), drange=training_set.dynamic_range, grid_size=grid_size)   # :172
        if reference_src is not None:
            with open(reference_src) as f:
                module_src = f.read()

    if attr_interesting is not None and attr_names is None:
        # :174-180 reads the vocabulary from celeba/Anno/list_attr_celeba.txt (relative to the working directory)
        attr_file = 'celeba/Anno/list_attr_celeba.txt'
        if not os.path.isfile(attr_file):
            raise FileNotFoundError('attr_interesting=%r needs the attribute names: pass attr_names=[...] or provide %s' % (attr_interesting, attr_file))
        if rank == 0:
            print('Loading attributes from "%s"...' % attr_file)
        attr_names = imle.attribute_names(attr_file)

    # Construct networks (:188-197).  Same seed on every rank => bit-identical replicas.
    G_args = dict(G_args); D_args = dict(D_args)
    G_args['func_name'] = _retarget(G_args.get('func_name', 'training.networks_stylegan2.G_main'))
    D_args['func_name'] = _retarget(D_args.get('func_name', 'training.networks_stylegan2.D_stylegan2_feature'))
    resume_kimg, resume_time = 0.0, 0.0
    if resume_pkl is None:
        G = tflib.Network('G', num_channels=training_set.shape[0], resolution=training_set.shape[1], label_size=training_set.label_size, device=device, seed=np_seed + 1, **G_args)
        D = tflib.Network('D', num_channels=training_set.shape[0], resolution=training_set.shape[1], label_size=training_set.label_size, device=device, seed=np_seed + 2, **D_args)
        Gs = G.clone('Gs')
    else:       # :193-195 -- a snapshot of this engine or of the reference (same pickle layout, training/misc.py)
        if rank == 0:
            print('Loading networks from "%s"...' % resume_pkl)
        resume_kimg, resume_time = misc.resume_kimg_time(resume_pkl)
        G, D, Gs = misc.as_networks(misc.load_pkl(resume_pkl), device=device)
    lpips = tflib.Network('lpips', func_name=lpips_func_name, resolution=training_set.shape[1], device=device, seed=np_seed + 3)
    proj_dim = func_proj_dim(init_proj_dim, data_size, num_samples_factor, G)
    grid_latents = np.random.randn(int(np.prod(grid_size)), *G.input_shapes[0][1:])      # :203 (consumes the host stream)

    # Build random projector (:205-213; the reference caches it in an .npy next to the run -- not done here)
    projector = None
    if init_proj_dim is not None:
        out_dim = int(np.prod(G.output_shape[1:]))
        if rank == 0:
            print('Building random projector %d to %d...' % (out_dim, proj_dim))
        projector_np = np.random.normal(loc=0.0, scale=1.0 / float(proj_dim), size=(out_dim, proj_dim)).astype(np.float64)
        projector = torch.from_numpy(projector_np.astype(np.float32)).to(device)

    if rank == 0:
        G.print_layers(); D.print_layers()
    sched = training_schedule(cur_nimg=0, training_set=training_set, **sched_args)
    from ..metrics import metric_base
    metrics = metric_base.MetricGroup(metric_arg_list)       # :198

    # Setup optimizers (:242-255).
    cur_lrate = [sched.G_lrate]
    ratios = {}
    ratios['G'], G_opt_args = lazy_regularization_args(G_opt_args, G_reg_interval, lazy_regularization)
    ratios['D'], D_opt_args = lazy_regularization_args(D_opt_args, D_reg_interval, lazy_regularization)
    G_lr = lambda: cur_lrate[0] * ratios['G']
    D_lr = lambda: cur_lrate[0] * ratios['D']      # both optimizers are fed sched.G_lrate (:218,246,347)
    G_opt = tflib.Optimizer(name='TrainG', learning_rate=G_lr, **G_opt_args)
    D_opt = tflib.Optimizer(name='TrainD', learning_rate=D_lr, **D_opt_args)
    G_reg_opt = tflib.Optimizer(name='RegG', share=G_opt, learning_rate=G_lr, **G_opt_args)
    D_reg_opt = tflib.Optimizer(name='RegD', share=D_opt, learning_rate=D_lr, **D_opt_args)

    G_loss_args = dict(G_loss_args); D_loss_args = dict(D_loss_args)
    G_loss_fn = dnnlib.util.get_obj_by_name(_retarget(G_loss_args.pop('func_name')))
    D_loss_fn = dnnlib.util.get_obj_by_name(_retarget(D_loss_args.pop('func_name')))

    minibatch_size_holder = [sched.minibatch_size]
    Gs_beta = lambda: smoothing_beta(minibatch_size_holder[0], G_smoothing_kimg)      # :222
    Gs_update_op = Gs.setup_as_moving_average_of(G, beta=Gs_beta)

    # ---- the four training ops (:278-297) ------------------------------------------
    # Each op = [device work: losses + backward into the flat gradient bucket] + [optimizer update].
    # The device work reads its inputs from static buffers and is captured into a hipGraph after
    # two eager executions (tflib/graphs.py); the update (all-reduce + finite check + Adam: a handful
    # of launches) stays eager so that the RCCL call is not part of a captured graph.
    B = sched.minibatch_gpu
    C, R = training_set.shape[0], training_set.shape[1]
    LS = training_set.label_size
    # The six per-iteration inputs of the G ops live in ONE device buffer and each staging set in ONE pinned host buffer (views per tensor, 256-byte aligned):
    # an iteration uploads them with a single asynchronous copy.  (Round 6: as six copies the device sat idle between them -- 0.8 ms per iteration in
    # profiles/r05_bench_steady_state.txt, "copyBuffer -> copyBuffer" -- because every copy is a host submission of its own.)
    rec_shapes = dict(reals_rec_1=(B, C, R, R), labels_rec_1=(B, LS), latents_rec_1=tuple([B] + G.input_shapes[0][1:]),
                      reals_rec_2=(B, C, R, R), labels_rec_2=(B, LS), latents_rec_2=tuple([B] + G.input_shapes[0][1:]))
    rec_offsets, rec_total = {}, 0
    for k, shp in rec_shapes.items():
        rec_offsets[k] = rec_total
        rec_total += (int(np.prod(shp)) + 63) // 64 * 64
    rec_total = max(rec_total, 64)
    rec_views = lambda flat: {k: flat[rec_offsets[k]:rec_offsets[k] + int(np.prod(shp))].view(shp) for k, shp in rec_shapes.items()}
    feed_flat = torch.zeros((rec_total,), device=device)
    feed = dict(rec_views(feed_flat),
                reals=torch.zeros((2 * B, C, R, R), device=device, dtype=torch.uint8), labels=torch.zeros((2 * B, LS), device=device))
    staging = []
    for _ in range(3):
        flat = torch.zeros((rec_total,), dtype=torch.float32).pin_memory()
        staging.append(dict(rec_views(flat), flat=flat, event=torch.cuda.Event()))
    stage_np = [{k: v.numpy() for k, v in st.items() if k not in ('event', 'flat')} for st in staging]       # views of the pinned buffers (same memory)
    use_graphs = graphs.graphs_enabled(hip_graphs)
    # Gradient exchange: chunk by chunk DURING backward (tflib/optimizer.py GradientExchange), inside the captured graph
    # when the process group's collectives can be captured (RCCL); otherwise (gloo) one all-reduce after each replay.
    overlap_exchange = world > 1 and (not use_graphs or tflib.optimizer.collectives_capturable())

    def G_grad():
        G.invalidate_derived(); D.invalidate_derived()
        D.requires_grad_(False)
        reals_1, labels_1 = process_reals(feed['reals_rec_1'], feed['labels_rec_1'], 0, mirror_augment, training_set.dynamic_range, drange_net)
        reals_2, labels_2 = process_reals(feed['reals_rec_2'], feed['labels_rec_2'], 0, mirror_augment, training_set.dynamic_range, drange_net)
        loss, reg = G_loss_fn(G=G, D=D, lpips=lpips, training_set=training_set, minibatch_size=B,
                              reals_rec_1=reals_1, labels_rec_1=labels_1, latents_rec_1=feed['latents_rec_1'],
                              reals_rec_2=reals_2, labels_rec_2=labels_2, latents_rec_2=feed['latents_rec_2'],
                              phase='loss' if lazy_regularization else 'both', **G_loss_args)
        if not lazy_regularization and reg is not None:
            loss = loss + reg       # :284-285 (broadcasts [B] + [B // pl_minibatch_shrink] exactly as the reference's `+=` does, or fails like it)
        G_opt.differentiate(torch.mean(loss), G, overlap_exchange=overlap_exchange)                    # register_gradients(tf.reduce_mean(G_loss)) :290
        D.requires_grad_(True)
        return loss

    def G_reg_grad():
        G.invalidate_derived(); D.invalidate_derived()
        D.requires_grad_(False)
        _, reg = G_loss_fn(G=G, D=D, lpips=lpips, training_set=training_set, minibatch_size=B,
                           reals_rec_1=None, labels_rec_1=None, latents_rec_1=feed['latents_rec_1'],
                           reals_rec_2=None, labels_rec_2=None, latents_rec_2=feed['latents_rec_2'], phase='reg', **G_loss_args)
        G_reg_opt.differentiate(torch.mean(reg * G_reg_interval), G, overlap_exchange=overlap_exchange)   # :288
        D.requires_grad_(True)
        return reg

    def D_grad():
        G.invalidate_derived(); D.invalidate_derived()
        reals, labels = process_reals(feed['reals'], feed['labels'], 0, mirror_augment, training_set.dynamic_range, drange_net)
        G.requires_grad_(False)
        loss, reg = D_loss_fn(G=G, D=D, training_set=training_set, minibatch_size=B, reals=reals, labels=labels,
                              phase='loss' if lazy_regularization else 'both', **D_loss_args)
        if not lazy_regularization and reg is not None:
            loss = loss + reg       # :286
        G.requires_grad_(True)
        D_opt.differentiate(torch.mean(loss), D, overlap_exchange=overlap_exchange)                    # :291
        return loss

    def D_reg_grad():
        G.invalidate_derived(); D.invalidate_derived()
        reals, labels = process_reals(feed['reals'], feed['labels'], 0, mirror_augment, training_set.dynamic_range, drange_net)
        G.requires_grad_(False)
        _, reg = D_loss_fn(G=G, D=D, training_set=training_set, minibatch_size=B, reals=reals, labels=labels, phase='reg', **D_loss_args)
        G.requires_grad_(True)
        D_reg_opt.differentiate(torch.mean(reg * D_reg_interval), D, overlap_exchange=overlap_exchange)   # :289
        return reg

    G_grad_step = graphs.GraphedStep(G_grad, use_graphs, eager_calls=1, name='G')
    G_reg_step = graphs.GraphedStep(G_reg_grad, use_graphs, eager_calls=1, name='G_reg')
    D_grad_step = graphs.GraphedStep(D_grad, use_graphs, eager_calls=1, name='D')
    D_reg_step = graphs.GraphedStep(D_reg_grad, use_graphs, eager_calls=1, name='D_reg')

    graph_checks = []

    def validate_graphs(reason):
        """check_replay() of every captured op (the others replayed in front): after capture, after any re-capture (GraphedStep.generation)
        and on every network-snapshot tick of a long run -- the fault this guards against depended on what else had run
        (profiles/r03_graph_packet_capture.txt).  An op whose graph disagrees runs eagerly from then on.  State, gradients and the
        generator are left as found; the summary accumulators the extra executions touched are the caller's to flush."""
        if not use_graphs or not graphs.validation_enabled():
            return True
        ok = True
        state = [G.vars['dlatent_avg']] + ([G.pl_mean_var] if hasattr(G, 'pl_mean_var') else [])
        all_steps = (G_grad_step, G_reg_step, D_grad_step, D_reg_step)
        for step, net in zip(all_steps, (G, G, D, D)):
            if step.graph is None or not step.enabled:
                continue
            bad = step.check_replay(state, lambda out, net=net: [out, net.flat_grads], context=[s for s in all_steps if s is not step] if os.environ.get('IGAN_GRAPH_VALIDATE_CONTEXT', '1') != '0' else [])
            flag = torch.tensor([1.0 if bad else 0.0], device=device)
            if world > 1:       # every rank takes the same decision (the ops contain collectives)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if bool(flag.item()):
                ok = False
                print('WARNING: hipGraph of training op %r does not reproduce its eager execution (%s; tensor index, max |diff|: %s); '
                      'running this op eagerly' % (step.name, reason, bad), flush=True)
                for i, pos, xa, xb in getattr(step, 'last_diff', []):          # DIAGNOSTIC (IGAN_GRAPH_CHECK_VERBOSE=1): which variables' gradients differ
                    if i == 1:
                        names = {}
                        for vn, (o, c) in net._offsets.items():
                            sel = (pos >= o) & (pos < o + c)
                            if bool(sel.any()):
                                rel = ((xa[sel].double() - xb[sel].double()).abs().max() / xb[sel].double().abs().max().clamp_min(1e-300)).item()
                                names[vn] = (int(sel.sum()), int(c), float('%.2e' % rel))
                        print('REPLAY-DIFF rank %d op %s gradient bucket: %d of %d elements differ; by variable (differing, size, max diff / max |value|): %s' % (
                            rank, step.name, pos.numel(), net.flat_grads.numel(), names), flush=True)
                    else:
                        print('REPLAY-DIFF rank %d op %s tensor %d: %d elements differ: replay %s eager %s' % (rank, step.name, i, pos.numel(), xa[:6].tolist(), xb[:6].tolist()), flush=True)
                if os.environ.get('IGAN_GRAPH_STRESS') != '1':     # stress runs (bench.py --revalidate-every) keep checking the op instead of retiring its graph
                    step.enabled = False
        graph_checks.append(dict(when=reason, faithful=ok))
        return ok

    if use_graphs:
        # Build all four graphs before the first iteration: one eager dry run of each op's device work
        # (creates every lazily allocated piece of state), then capture.  No optimizer update is applied;
        # the state the dry runs touch (dlatent_avg, pl_mean, summary accumulators) is restored afterwards.
        saved = {n: v.clone() for n, v in G.vars.items() if n in ('dlatent_avg',)}
        for k in ('latents_rec_1', 'latents_rec_2'):
            feed[k].normal_()
        for step in ((G_grad_step, G_reg_step, D_grad_step, D_reg_step) if lazy_regularization else (G_grad_step, D_grad_step)):
            step()      # eager
            step()      # capture + first replay
        # Every captured op must reproduce its own eager execution bit for bit (same inputs, same state, same generator
        # state) before the run is allowed to depend on it; a graph that does not is not replayed -- its op runs eagerly.
        graphs_ok = validate_graphs('after capture')
        if 'on_graphs' in hooks:
            hooks['on_graphs'](dict(captured=True, validated=graphs.validation_enabled(), faithful=graphs_ok, checks=graph_checks, runtime=graphs.runtime_info()))
        with torch.no_grad():
            for n, v in saved.items():
                G.vars[n].copy_(v)
            if hasattr(G, 'pl_mean_var'):
                G.pl_mean_var.zero_()
            G.flat_grads.zero_(); D.flat_grads.zero_()
        autosummary_mod.flush()
        torch.cuda.synchronize()

    def next_reals():
        # `training_set.get_minibatch_tf()` is an iterator op: every session.run that consumes it
        # (D_train_op and, separately, D_reg_op -- :231,477,479) pulls the next minibatch.
        reals, labels = training_set.get_minibatch_tf()
        feed['reals'].copy_(reals)
        feed['labels'].copy_(labels)

    def G_train_op():
        out = G_grad_step()
        G_opt.mark_registered(G)
        G_opt.apply_updates()
        return 'G', out

    def G_reg_op():
        out = G_reg_step()
        G_reg_opt.mark_registered(G)
        G_reg_opt.apply_updates(allow_no_op=True)
        return 'G_reg', out

    def D_train_op():
        next_reals()
        out = D_grad_step()
        D_opt.mark_registered(D)
        D_opt.apply_updates()
        return 'D', out

    def D_reg_op():
        next_reals()
        out = D_reg_step()
        D_reg_opt.mark_registered(D)
        D_reg_opt.apply_updates(allow_no_op=True)
        return 'D_reg', out

    if rank == 0:
        print('Training for %d kimg...\n' % total_kimg)
    cur_nimg = int(resume_kimg * 1000)
    cur_tick = -1
    tick_start_nimg = cur_nimg
    tick_start_time = time.time()
    start_time = tick_start_time
    maintenance_time = 0.0
    final_grids = None
    running_mb_counter = 0
    latent_candidates = np.random.randn(data_size * num_samples_factor, *G.input_shapes[0][1:]).astype(np.float32)  # :325

    def search(latents, label_candidates, minibatch_size):
        t0 = time.time()
        # packs of candidates are dealt to the ranks: keep them small enough that every rank gets some (and at most 4096)
        per_rank = -(-latents.shape[0] // world)
        pack = max(candidate_batch_size, min(4096, -(-per_rank // candidate_batch_size) * candidate_batch_size))
        out = imle_refresh(G, training_set_rec, latents, label_candidates, data_size, minibatch_size, candidate_batch_size,
                           drange_net, device, rank=rank, world=world, projector=projector,
                           exclusive_k=num_samples_factor if exclusive_retrieved_code else 0, cand_pack=pack)      # :382-386
        torch.cuda.synchronize()
        if 'on_refresh' in hooks:
            hooks['on_refresh'](time.time() - t0)
        if 'on_assignment' in hooks:
            hooks['on_assignment'](out[0], out[1])
        return out

    # Host side of the IMLE term (:325-464): refresh cadence, selection, carry-over, perturbation, shuffles -- training/imle.py
    sampler = imle.ImleSampler(training_set_rec, latent_candidates, data_size, num_samples_factor, init_staleness, knn_perturb_factor,
                               dist_thres_percentile=dist_thres_percentile, attr_interesting=attr_interesting, attr_names=attr_names,
                               search=search)
    if 'on_start' in hooks:
        hooks['on_start'](dict(G=G, D=D, Gs=Gs, lpips=lpips, feed=feed, training_set=training_set, G_opt=G_opt, D_opt=D_opt))
    stop = False
    # Submission thread (SubmitThread; off by default: measured slower, see its docstring): never when a hook wants to look at device state after every op / time the ops
    # (those hooks are synchronous by contract: the parity tests).  A hook set may carry 'async_ok': True to say that its on_iteration
    # does not read device state without calling info['drain']() first (bench.py); otherwise the queue is drained before on_iteration.
    use_async = (os.environ.get('IGAN_ASYNC_SUBMIT', '0') == '1' and use_graphs and not any(k in hooks for k in ('on_op', 'op_times', 'on_batch')))
    submitter = SubmitThread(device) if use_async else None
    submit = submitter.submit if use_async else (lambda fn: fn())
    drain = submitter.drain if use_async else (lambda: None)
    backlog = submitter.wait_backlog if use_async else (lambda n: None)
    while cur_nimg < total_kimg * 1000 and not stop:
        # Choose training parameters (:336-340).
        sched = training_schedule(cur_nimg=cur_nimg, training_set=training_set, **sched_args)
        assert sched.minibatch_size % (sched.minibatch_gpu * num_gpus) == 0
        assert sched.minibatch_size // (sched.minibatch_gpu * num_gpus) == 1   # "fast path without gradient accumulation" (:467)
        assert data_size % (sched.minibatch_size * 2) == 0
        training_set.configure(sched.minibatch_size * 2, sched.lod)
        training_set_rec.configure(sched.minibatch_size * 2, sched.lod)
        if sched.G_lrate != cur_lrate[0] or sched.minibatch_size != minibatch_size_holder[0]:
            drain()         # queued iterations read these lazily (learning rate, Gs beta): they must have run before the schedule's next values are set
        cur_lrate[0] = sched.G_lrate
        minibatch_size_holder[0] = sched.minibatch_size
        mb = sched.minibatch_size

        for _repeat in range(minibatch_repeats):
            run_G_reg = (lazy_regularization and running_mb_counter % G_reg_interval == 0)
            run_D_reg = (lazy_regularization and running_mb_counter % D_reg_interval == 0)

            # IMLE refresh (:354-406) and this iteration's (real, label, latent) triples (:409-464).
            if sampler.refresh_due(cur_nimg, mb):
                drain()                 # the refresh runs the generator from this thread: every earlier update must have been handed over
                sampler.refresh(mb)
            batch = sampler.next_batch(mb)
            if 'on_batch' in hooks:
                hooks['on_batch'](batch)
            halves = [(batch['reals_rec_%d' % h], batch['labels_rec_%d' % h], batch['latents_rec_%d' % h]) for h in (1, 2)]

            # This rank's slice of the global minibatch (tf.split, :231-239) -> pinned staging -> static device buffers.
            # The copies are asynchronous (a pageable source would make every copy a stream synchronisation, i.e. an idle
            # device while the host prepares the next iteration -- and an idle-prone device also runs at a lower clock).
            assert sched.minibatch_gpu == B, 'minibatch_gpu must stay constant (static buffers / captured graphs)'
            rs = slice(rank * B, (rank + 1) * B)
            backlog(1)                              # at most one iteration waits behind the one being submitted: the upload that last read the
            stage = staging[running_mb_counter % len(staging)]      # staging set below (three sets: three iterations ago) has been issued ...
            stage['event'].synchronize()            # ... and has run
            for h, (r_, l_, z_) in enumerate(halves):
                for key, arr in (('reals_rec_%d', r_), ('labels_rec_%d', l_), ('latents_rec_%d', z_)):
                    # filled through the pinned tensor's NumPy view: a torch CPU copy_ would open a parallel region over PyTorch's whole
                    # intra-op pool (128 threads on this host) for 600 KB -- inclusivegan_amd/hostaffinity.py has the measurement
                    np.copyto(stage_np[running_mb_counter % len(staging)][key % (h + 1)], arr[rs], casting='same_kind')

            def upload(stage=stage):
                feed_flat.copy_(stage['flat'], non_blocking=True)      # all six inputs: one copy
                stage['event'].record()

            # Run training ops (:474-479) -- handed to the submission thread in program order (or executed here when it is off).
            timed = hooks.get('op_times')          # optional: dict name -> list of (start, end) HIP events
            on_op = hooks.get('on_op')
            host_timed = hooks.get('op_host_times')  # optional: dict name -> list of seconds the HOST spent inside the op's call (submission time)
            def run(name, op):
                if timed is None:
                    res = op()
                else:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    t0 = time.perf_counter()
                    e0.record(); res = op(); e1.record()
                    timed.setdefault(name, []).append((e0, e1))
                    if host_timed is not None:
                        host_timed.setdefault(name, []).append(time.perf_counter() - t0)
                if on_op is not None and res is not None:
                    on_op(res[0], res[1], feed)

            def iteration(run_G_reg=run_G_reg, run_D_reg=run_D_reg, upload=upload):
                upload()
                run('G_train', G_train_op)
                if run_G_reg:
                    run('G_reg', G_reg_op)
                run('D_train', D_train_op)
                run('Gs_update', Gs_update_op)
                if run_D_reg:
                    run('D_reg', D_reg_op)
            submit(iteration)

            cur_nimg += mb * 2
            running_mb_counter += 1
            if 'on_iteration' in hooks:
                if not hooks.get('async_ok'):
                    drain()
                def revalidate(reason):
                    drain()
                    return validate_graphs(reason)
                if hooks['on_iteration'](dict(cur_nimg=cur_nimg, iteration=running_mb_counter, G=G, D=D, Gs=Gs, revalidate_graphs=revalidate, drain=drain)):
                    stop = True
                    break

        # Per-tick maintenance (:485-525): progress line (teed into run_dir/log.txt), snapshots, metrics on network snapshots.
        done = (cur_nimg >= total_kimg * 1000) or stop
        if cur_tick < 0 or cur_nimg >= tick_start_nimg + sched.tick_kimg * 1000 or done:
            cur_tick += 1
            drain()
            torch.cuda.synchronize()
            now = time.time()
            tick_kimg = max((cur_nimg - tick_start_nimg) / 1000.0, 1e-9)
            tick_time = now - tick_start_time
            total_time = now - start_time + resume_time
            sums = autosummary_mod.flush()
            window_warning = _f16_window_warning()     # fp16 form: row-image elements outside their pixel's exact window, since the last tick (watched at run time)
            if rank == 0 and window_warning:
                print(window_warning, flush=True)
            if rank == 0:
                lines = ['tick %-5d kimg %-8.1f lod %-5.2f minibatch %-4d time %-12s sec/tick %-7.1f sec/kimg %-7.2f maintenance %-6.1f gpumem %.1f' % (
                    cur_tick, cur_nimg / 1000.0, sched.lod, sched.minibatch_size, dnnlib.util.format_time(total_time),
                    tick_time, tick_time / tick_kimg, maintenance_time, torch.cuda.max_memory_allocated() / 2**30)]       # the reference's line, field for field (:495-504)
                lines += ['    %-32s %g' % (k, v) for k, v in sums.items()]
                print('\n'.join(lines))
                if run_dir is not None:
                    with open(os.path.join(run_dir, 'log.txt'), 'a') as f:
                        f.write('\n'.join(lines) + '\n')
            tick_start_nimg = cur_nimg

            # Save snapshots (:506-519); rank 0 only, and only when a run directory was asked for.
            if run_dir is not None and rank == 0:
                snap = lambda name: os.path.join(run_dir, name)
                if image_snapshot_ticks is not None and (cur_tick % max(image_snapshot_ticks, 1) == 0 or done):
                    grid_fakes = Gs.run(grid_latents, grid_labels, is_validation=True, minibatch_size=sched.minibatch_gpu)
                    misc.save_image_grid(grid_fakes, snap('arb-fakes-%06d.png' % (cur_nimg // 1000)), drange=drange_net, grid_size=grid_size)
                    if sampler.tick is not None:
                        t_reals, t_labels, t_latents = sampler.tick
                        rec_grid = (8, (sched.minibatch_size * 2) // 8) if (sched.minibatch_size * 2) % 8 == 0 else None
                        if tick_reals_old is None or np.sum(t_reals != tick_reals_old) > 0:
                            misc.save_image_grid(t_reals, snap('rec-reals.png'), drange=training_set.dynamic_range, grid_size=rec_grid)
                            tick_reals_old = np.array(t_reals)
                        fakes_nn = Gs.run(t_latents, t_labels, is_validation=True, minibatch_size=sched.minibatch_gpu)
                        misc.save_image_grid(fakes_nn, snap('rec-fakes-%06d.png' % (cur_nimg // 1000)), drange=drange_net, grid_size=rec_grid)
                        final_grids = (grid_fakes, fakes_nn, rec_grid)
                if network_snapshot_ticks is not None and (cur_tick % max(network_snapshot_ticks, 1) == 0 or done):
                    pkl = snap('network-snapshot-%06d.pkl' % (cur_nimg // 1000))
                    misc.save_pkl((G, D, Gs), pkl, reference_layout=True, build_module_src=module_src)
                    rng_state = torch.cuda.get_rng_state(device)    # the reference's metric ops carry their own seeds: evaluating a metric must not
                    metrics.run(pkl, run_dir=run_dir, data_dir=data_dir, dataset_args=ds_args, mirror_augment=mirror_augment,
                                num_gpus=min([2, num_gpus]), tf_config=tf_config, device=device)                   # :519
                    torch.cuda.set_rng_state(rng_state, device)     # shift this rank's training draws (a run with metrics == a run without)
            if network_snapshot_ticks is not None and cur_tick > 0 and cur_tick % max(network_snapshot_ticks, 1) == 0 and use_graphs:
                validate_graphs('tick %d' % cur_tick)      # long runs: the replays are re-checked at the network-snapshot cadence
                autosummary_mod.flush()                    # the check's extra executions do not reach the next tick's statistics
            metrics.update_autosummaries()                                                                        # :522
            torch.cuda.synchronize()
            tick_start_time = time.time()
            maintenance_time = tick_start_time - now

    if submitter is not None:
        submitter.close()
    # Save final snapshot (:527-530).
    if run_dir is not None and rank == 0:
        if final_grids is not None:
            misc.save_image_grid(final_grids[0], os.path.join(run_dir, 'arb-fakes-final.png'), drange=drange_net, grid_size=grid_size)
            misc.save_image_grid(final_grids[1], os.path.join(run_dir, 'rec-fakes-final.png'), drange=drange_net, grid_size=final_grids[2])
        misc.save_pkl((G, D, Gs), os.path.join(run_dir, 'network-final.pkl'), reference_layout=True, build_module_src=module_src)

    training_set.close()
    training_set_rec.close()
    return dict(G=G, D=D, Gs=Gs, cur_nimg=cur_nimg)


def _retarget(func_name):
    """A reference config names 'training.loss.X' / 'training.networks_stylegan2.X'
    (run_training.py:52-57); resolve those dotted names inside this package."""
    if func_name.startswith('training.') or func_name.startswith('metrics.'):
        return 'inclusivegan_amd.' + func_name
    return func_name

#----------------------------------------------------------------------------
