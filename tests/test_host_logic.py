"""CPU: host-side logic of the path -- network object model (meta device), dataset feeding / rank slicing,
schedule + refresh cadence, optimizer state sharing -- and the N > 1 data-parallel exchange under gloo
(world_size 2)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_network_inventory_matches_survey():
    """Trainable tensor / parameter counts of config-e-Gskip-Dresnet."""
    from inclusivegan_amd.dnnlib import tflib
    want = {32: (68, 21523475, 23, 21507073), 128: (96, 24525213, 33, 23882369)}
    for res, (gt, gp, dt, dp) in want.items():
        kw = dict(num_channels=3, resolution=res, label_size=40, fmap_base=8192, device='meta')
        G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', **kw)
        D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', **kw)
        assert (len(G.trainables), G.num_params(), len(D.trainables), D.num_params()) == (gt, gp, dt, dp)
        assert G.input_shapes == [[None, 512], [None, 40]] and G.output_shape == [None, 3, res, res]
        assert G.components.synthesis.input_shape == [None, int(np.log2(res)) * 2 - 2, 512]
    names = list(G.trainables)
    assert names[0] == 'G_synthesis/4x4/Const/const' and 'G_synthesis/128x128/Conv1/mod_weight' in names
    assert 'G_mapping/Dense7/bias' in names and list(G.vars)[-2:] == ['lod', 'dlatent_avg']
    assert tuple(G.vars['G_synthesis/64x64/Conv0_up/weight'].shape) == (3, 3, 512, 256)      # HWIO
    assert tuple(D.vars['4x4/Dense0/weight'].shape) == (8192, 512)


def test_flat_bucket_views_and_clone():
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=16, label_size=0, fmap_base=256, device='cpu')
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=5, **kw)
    G2 = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=5, **kw)
    assert torch.equal(G.flat_params, G2.flat_params)          # same seed => bit-identical replicas
    for v in G.trainables.values():                             # every trainable is a 16 B-aligned view of the bucket
        assert v.data_ptr() % 16 == 0 and v.grad.data_ptr() % 16 == 0
        assert G.flat_params.data_ptr() <= v.data_ptr() < G.flat_params.data_ptr() + 4 * G.flat_params.numel()
    w = G.vars['G_synthesis/8x8/Conv1/weight']
    std = float(w.detach().std())
    assert 0.9 < std < 1.1                                      # N(0,1) * init_mul, runtime coef applied in get_weight
    assert float(G.vars['G_synthesis/8x8/Conv1/bias'].abs().max()) == 0 and float(G.vars['G_synthesis/8x8/Conv1/noise_strength'].detach()) == 0
    Gs = G.clone('Gs')
    assert torch.equal(Gs.flat_params, G.flat_params) and Gs.flat_params.data_ptr() != G.flat_params.data_ptr()
    with torch.no_grad():
        G.flat_grads.fill_(1.0)
    G.zero_grad()
    assert float(G.flat_grads.abs().max()) == 0


def test_dataset_rank_slices_tile_the_global_minibatch():
    from inclusivegan_amd.training.dataset import SyntheticDataset
    full = SyntheticDataset(resolution=8, label_size=5, data_size=48, seed=3)
    full.configure(24)
    ref_imgs, ref_labels = full.get_minibatch_np(24)
    parts = []
    for r in range(4):
        ds = SyntheticDataset(resolution=8, label_size=5, data_size=48, seed=3, rank=r, world_size=4)
        ds.configure(24)
        x, l = ds.get_minibatch_tf()
        assert x.dtype == torch.uint8 and tuple(x.shape) == (6, 3, 8, 8)
        parts.append((x.numpy(), l.numpy()))
    assert np.array_equal(np.concatenate([p[0] for p in parts]), ref_imgs)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ref_labels)
    a, _ = full.get_minibatch_np(24)      # next 24, then wrap-around (shuffle_mb=0: dataset order)
    b, _ = full.get_minibatch_np(24)
    assert np.array_equal(b, ref_imgs) and not np.array_equal(a, ref_imgs)
    oh = SyntheticDataset(resolution=8, label_size=10, data_size=16, label_kind='onehot')
    assert np.array_equal(oh._labels.sum(1), np.ones(16))


def test_schedule_and_refresh_cadence():
    from inclusivegan_amd.training import training_loop as TL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    ts = SyntheticDataset(resolution=32, label_size=0, data_size=24)
    s = TL.training_schedule(0, ts, minibatch_size_base=12, minibatch_gpu_base=6, G_lrate_base=0.002, D_lrate_base=0.002)
    assert (s.lod, s.resolution, s.minibatch_size, s.minibatch_gpu, s.G_lrate, s.tick_kimg) == (0.0, 32, 12, 6, 0.002, 1)
    # refresh rule of training_loop.py:354-356 for data_size 30000, staleness 10, minibatch 6: first iteration, then
    # every time cur_nimg crosses a multiple of data_size*staleness, the staleness doubling after each one
    data_size, stale, mb, cur, selected, refreshes = 30000, 10, 6, 0, False, []
    for it in range(200000):
        if not selected or cur // (data_size * stale) != (cur - mb * 2) // (data_size * stale):
            if selected:
                stale *= 2
            refreshes.append(cur)
            selected = True
        cur += mb * 2
    assert refreshes == [0, 300000, 600000, 1200000] and stale == 80
    class FakeG:
        output_shape = [None, 3, 128, 128]
    assert TL.func_proj_dim(None, 30000, 10, FakeG) == 49152
    assert TL._retarget('training.loss.D_logistic_r1') == 'inclusivegan_amd.training.loss.D_logistic_r1'


def test_lazy_reg_optimizer_settings_follow_the_reference():
    """lr / beta scaling of training_loop.py:244-251 (c = k/(k+1)) as assembled by the loop."""
    c_g, c_d = 4 / 5, 16 / 17
    assert abs(0.99 ** c_g - 0.9919919) < 1e-6 and 0.0 ** c_g == 0.0
    from inclusivegan_amd.dnnlib import tflib
    opt = tflib.Optimizer(name='TrainG', learning_rate=lambda: 0.002 * c_g, beta1=0.0 ** c_g, beta2=0.99 ** c_g, epsilon=1e-8)
    reg = tflib.Optimizer(name='RegG', share=opt, learning_rate=lambda: 0.002 * c_g, beta1=0.0 ** c_g, beta2=0.99 ** c_g, epsilon=1e-8)
    assert reg._state is opt._state          # shared Adam slots (optimizer.py:77-82)
    with pytest.raises(AssertionError):
        tflib.Optimizer(name='bad', share=opt, beta1=0.5, beta2=0.99, epsilon=1e-8)
    with pytest.raises(NotImplementedError):
        tflib.Optimizer(use_loss_scaling=True)
    del c_d


def test_random_tape_is_strict():
    from inclusivegan_amd.dnnlib.tflib import tfutil
    tape = tfutil.RandomTape([('normal', np.zeros((2, 3))), ('uniform', np.zeros(()))])
    with tfutil.use_random(tape):
        assert tuple(tfutil.random_normal([2, 3], 'cpu').shape) == (2, 3)
        with pytest.raises(RuntimeError):
            tfutil.random_normal([1], 'cpu')       # kind mismatch (tape has uniform)
    with tfutil.use_random(tfutil.RandomTape([])):
        with pytest.raises(RuntimeError):
            tfutil.random_uniform([1], 'cpu')      # exhausted


def test_network_variable_api_matches_reference_surface():
    """find_var / get_var / set_var / reset_* / copy_* (dnnlib/tflib/network.py:223-255,316-329) on a CPU-built network
    (no kernel runs: only the variable store) and the naive resamplers (networks_stylegan2.py:73-84)."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.training import networks_stylegan2 as N
    kw = dict(num_channels=3, resolution=8, label_size=0, fmap_base=64, device='cpu')
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=4, **kw)
    D2 = D.clone('D2')
    name = next(iter(D.trainables))
    assert D.get_var_local_name('D/' + name) == name and D.find_var(name) is D.vars[name]
    v0 = D.get_var(name).copy()
    D.set_var(name, v0 * 0 + 3.0)
    assert float(D.get_var(name).mean()) == 3.0 and float(D2.get_var(name).reshape(-1)[0]) == float(v0.reshape(-1)[0])
    D2.copy_trainables_from(D)
    assert float(D2.get_var(name).mean()) == 3.0
    D.reset_trainables()
    np.testing.assert_array_equal(D.get_var(name), v0)          # same seed -> the original initial values
    D2.copy_own_vars_from(D)
    np.testing.assert_array_equal(D2.get_var(name), v0)
    x = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    up = N.naive_upsample_2d(x)
    assert up.shape == (2, 3, 8, 8) and torch.equal(up[:, :, ::2, ::2], x) and torch.equal(up[:, :, 1::2, 1::2], x)
    assert torch.equal(N.naive_downsample_2d(up), x)


# ----------------------------------------------------------------------------- world_size 2 under gloo
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    from inclusivegan_amd.dnnlib.tflib.optimizer import allreduce_mean_
    from inclusivegan_amd.training.training_loop import combine_best_, _dist_info
    from inclusivegan_amd.dci_code.dci import unpack_best
    assert _dist_info() == (rank, world)
    # gradient exchange: every rank holds its own bucket; after the exchange all hold the mean
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(g, num_registered=1)
    # IMLE shards: rank r saw candidates r, r+world, ...; running minimum = (squared distance fp64, index int32)
    d = torch.tensor([[4.0, 1.0, 9.0], [2.0, 3.0, 9.0]], dtype=torch.float64)[rank]
    idx = torch.tensor([[10, 11, 12], [20, 21, 5]], dtype=torch.int32)[rank]
    combine_best_(d, idx, world)
    bi, bd = unpack_best(d, idx)
    # a rank without any candidate batch still holds the initial state (+inf, INT32_MAX): it must lose the exchange
    ld = torch.full((3,), float('inf'), dtype=torch.float64) if rank == 1 else d.clone()
    li = torch.full((3,), 2 ** 31 - 1, dtype=torch.int32) if rank == 1 else idx.clone()
    combine_best_(ld, li, world)
    assert torch.equal(ld, d) and torch.equal(li, idx)
    # identical host-side streams on every rank (np seed) -> identical shuffles / candidate latents
    np.random.seed(1000)
    order = np.arange(12); np.random.shuffle(order)
    # hook-driven chunked exchange during backward (tflib/optimizer.py GradientExchange) on a CPU-resident network:
    # rank r differentiates (r + 1) * sum(p^2) / 2, so the averaged bucket must be 1.5 * p for every trainable
    from inclusivegan_amd.dnnlib import tflib
    net = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet',
                        num_channels=3, resolution=16, label_size=0, fmap_base=128, device='cpu', seed=3)
    opt = tflib.Optimizer(name='T', learning_rate=0.01, beta1=0.0, beta2=0.99)
    loss = sum((p * p).sum() for p in net.trainables.values()) * (0.5 * (rank + 1))
    opt.differentiate(loss, net)            # world 2 -> overlapped exchange by default
    ex = opt._state['exchange']
    want = 1.5 * net.flat_params
    exch = (len(ex.chunks), sum(c.numel() for c in ex.chunks) == net.flat_grads.numel(), bool(torch.allclose(net.flat_grads, want, rtol=1e-6, atol=1e-7)),
            opt._exchanged)
    # the blocking form gives the same bucket bit for bit
    loss = sum((p * p).sum() for p in net.trainables.values()) * (0.5 * (rank + 1))
    bucket = net.flat_grads.clone()
    opt.differentiate(loss, net, overlap_exchange=False)
    allreduce_mean_(net.flat_grads, 1)
    exch = exch + (bool(torch.equal(bucket, net.flat_grads)),)
    out[rank] = (g.tolist(), bi.tolist(), bd.tolist(), order.tolist(), exch)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_exchange_gloo_world2():
    ctx = mp.get_context('spawn')
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        r0, r1 = out[0], out[1]
    mean = [(i * 1 + i * 2) / 2 for i in range(10)]
    assert r0[0] == mean and r1[0] == mean                       # optimizer.py:186,199: sum of grads / num_devices
    assert r0[1] == [20, 11, 5] and r1[1] == [20, 11, 5]         # per-real winner over both shards; tie (9.0) -> lower index
    assert r0[2] == [pytest.approx(2.0 ** 0.5), 1.0, 3.0]        # Euclidean (sqrt) distances
    assert r0[3] == r1[3]
    for r in (r0, r1):
        nchunks, covers, averaged, flagged, same_as_blocking = r[4]
        assert nchunks >= 2 and covers and averaged and flagged and same_as_blocking


def _probe_worker(rank, world, port, out):
    import os
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    from inclusivegan_amd.dnnlib.tflib.optimizer import collectives_capturable
    res = [collectives_capturable(probe=lambda g: True),                  # every rank's probe passes
           collectives_capturable(probe=lambda g: rank != 1),             # rank 1's probe fails: NOBODY may capture the collectives
           collectives_capturable()]                                      # gloo itself: never (host staging)
    out[rank] = res
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_capture_decision_is_the_same_on_every_rank():
    """the ranks must agree on whether the gradient exchange rides inside the captured graphs -- a rank that
    captured it next to one that did not would deadlock.  The per-rank probe results are combined with all-reduce(MIN)."""
    ctx = mp.get_context('spawn')
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_probe_worker, args=(r, 3, port, out)) for r in range(3)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        res = [out[r] for r in range(3)]
    assert res[0] == res[1] == res[2] == [True, False, False]


def test_mbstd_preserving_interleave_is_exact():
    """One D pass over interleaved (fakes, reals) gives every sample the minibatch-stddev statistic of its own
    separate pass (loss.D_logistic_r1 relies on this)."""
    from inclusivegan_amd.training.loss import _mbstd_preserving_interleave
    from oracle.networks_stylegan2 import minibatch_stddev_layer
    for n, g in [(12, 6), (6, 6), (24, 6), (8, 4)]:
        a = torch.randn(n, 4, 4, 4, dtype=torch.float64)
        b = torch.randn(n, 4, 4, 4, dtype=torch.float64)
        perm, inv = _mbstd_preserving_interleave(n, g, 'cpu')
        assert sorted(perm.tolist()) == list(range(2 * n)) and torch.equal(perm[inv], torch.arange(2 * n))
        y = minibatch_stddev_layer(torch.cat([a, b]).index_select(0, perm), g).index_select(0, inv)
        assert torch.equal(y[:n], minibatch_stddev_layer(a, g)) and torch.equal(y[n:], minibatch_stddev_layer(b, g))


def test_bench_evidence_plumbing():
    """bench.py's roofline bookkeeping (host side only): kernel names group into families by template name, the piece families are priced against the
    16-bit dense peak over their products and everything else against the fp32 matrix peak, the sustained rate of the piece products' instruction is read
    from the committed tools/mfma_rate run, and a counter file is either believed (taken on the kernel sources as they are now) or refused with the command to re-run."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.family_of('conv_fwd_planes_w4_kernel') == 'conv_fwd_planes_w4_kernel'
    assert b.family_of('conv_wgrad_planes_kernel<2> (+ reduce)') == 'conv_wgrad_planes_kernel'
    assert b.family_of('conv_fwd_dma_kernel<false, true, 0>') == 'conv_fwd_dma_kernel'
    assert b.family_peak('conv_fwd_dma_kernel') == b.F32_MATRIX_PEAK_TFLOPS
    sus = b.sustained_instruction_rate('f16')
    assert sus is not None and sus['instruction'] == 'v_mfma_f32_32x32x16_f16' and 1000.0 < sus['tflops'] < b.BF16_DENSE_PEAK_TFLOPS and 1000 < sus['clock_mhz'] < 2500, sus
    assert b.sustained_instruction_rate('bf16')['tflops'] > sus['tflops']       # the bf16 instruction holds a higher clock (profiles/r05_mfma_rate.txt)
    # a counter file is believed only on the kernel sources it was taken on: either a positive figure or a refusal that says what to re-run
    for suffix in ('pmc_dominant.json', 'pmc_conv_headline.json'):
        traffic, source = b.pmc_traffic(suffix)
        assert (traffic is not None and traffic > 0 and source.endswith(suffix)) or (traffic is None and 'collect_pmc' in source), (suffix, traffic, source)
    with open(b._latest_profile('pmc_dominant.json')) as f:
        assert b.family_of(json.load(f)['kernel']).startswith('conv_fwd_planes')
