"""Worker process of the world-size-2 GPU tests (tests/test_gpu_dist.py): both ranks share GPU 0 and talk over gloo
(RCCL refuses two ranks on one device; gloo moves device tensors through the host, which exercises exactly the same
Optimizer / GradientExchange / training-loop code paths).
usage: python tests/dist_worker.py <mode> <rank> <world> <port> <outdir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

RES, FMAP, B = int(os.environ.get('IGAN_TEST_RES', '32')), int(os.environ.get('IGAN_TEST_FMAP', '512')), 3


def build(dev):
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=RES, label_size=0, fmap_base=FMAP, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=41, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=42, **kw)
    lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=RES, device=dev, seed=43)
    c = 4 / 5
    G_opt = tflib.Optimizer(name='TrainG', learning_rate=0.002 * c, beta1=0.0, beta2=0.99 ** c, epsilon=1e-8)
    D_opt = tflib.Optimizer(name='TrainD', learning_rate=0.002 * c, beta1=0.0, beta2=0.99 ** c, epsilon=1e-8)
    return G, D, lp, G_opt, D_opt


def rank_inputs(rank, dev):
    g = torch.Generator().manual_seed(1000 + rank)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    return dict(r1=cl(torch.rand(B, 3, RES, RES, generator=g) * 2 - 1), r2=cl(torch.rand(B, 3, RES, RES, generator=g) * 2 - 1),
                z1=torch.randn(B, 512, generator=g).to(dev), z2=torch.randn(B, 512, generator=g).to(dev),
                reals=cl(torch.rand(2 * B, 3, RES, RES, generator=g) * 2 - 1))


def g_backward(G, D, lp, G_opt, inp, rank, ts, overlap):
    from inclusivegan_amd.training import loss as PL
    torch.manual_seed(500 + rank)
    lab = torch.zeros(B, 0, device=inp['z1'].device)
    D.requires_grad_(False)
    loss, _ = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, lp, ts, B, inp['r1'], lab, inp['z1'], inp['r2'], lab, inp['z2'],
                                                      NN_rec_lpips_weight=2.5, phase='loss')
    G_opt.differentiate(torch.mean(loss), G, overlap_exchange=overlap)
    D.requires_grad_(True)


def d_backward(G, D, D_opt, inp, rank, ts, overlap):
    from inclusivegan_amd.training import loss as PL
    torch.manual_seed(700 + rank)
    lab = torch.zeros(2 * B, 0, device=inp['z1'].device)
    G.requires_grad_(False)
    loss, _ = PL.D_logistic_r1(G, D, ts, B, inp['reals'], lab, gamma=100, phase='loss')
    G.requires_grad_(True)
    D_opt.differentiate(torch.mean(loss), D, overlap_exchange=overlap)


def mode_exchange(rank, world, outdir):
    """One G step + one D step through Optimizer.differentiate (hook-driven chunked exchange) / apply_updates."""
    from inclusivegan_amd.training.dataset import SyntheticDataset
    dev = torch.device('cuda', 0)
    G, D, lp, G_opt, D_opt = build(dev)
    ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
    inp = rank_inputs(rank, dev)
    g_backward(G, D, lp, G_opt, inp, rank, ts, overlap=True)
    g_avg = G.flat_grads.clone()
    G_opt.mark_registered(G); G_opt.apply_updates()
    d_backward(G, D, D_opt, inp, rank, ts, overlap=True)
    d_avg = D.flat_grads.clone()
    D_opt.mark_registered(D); D_opt.apply_updates()
    ex = G_opt._state['exchange']
    torch.save(dict(G=G.flat_params.cpu(), D=D.flat_params.cpu(), g_avg=g_avg.cpu(), d_avg=d_avg.cpu(),
                    chunks=[int(c.numel()) for c in ex.chunks]), os.path.join(outdir, 'rank%d.pt' % rank))


def mode_nonfinite(rank, world, outdir):
    """The finite gate AFTER the exchange (dnnlib/tflib/optimizer.py:237 of the reference: `tf.reduce_all(tf.is_finite(g))` on the summed gradients):
    rank 1's D loss is multiplied by inf, rank 0's is finite -- the averaged bucket is non-finite on BOTH ranks, NOBODY updates (weights, Adam
    moments and beta powers untouched, one overflow counted on each rank); the next, finite step updates both ranks identically."""
    from inclusivegan_amd.training.dataset import SyntheticDataset
    from inclusivegan_amd.training import loss as PL
    dev = torch.device('cuda', 0)
    G, D, lp, G_opt, D_opt = build(dev)
    ts = SyntheticDataset(resolution=RES, label_size=0, data_size=24, device=dev)
    inp = rank_inputs(rank, dev)
    lab = torch.zeros(2 * B, 0, device=dev)
    rec = {}
    D_opt._bind(D)        # the Adam slots exist from here on
    for step, poison in (('poisoned', rank == 1), ('clean', False)):
        before = D.flat_params.clone()
        st0 = {k: D_opt._state[k].clone() for k in ('m', 'v', 'pow')}
        torch.manual_seed(700 + rank)
        G.requires_grad_(False)
        loss, _ = PL.D_logistic_r1(G, D, ts, B, inp['reals'], lab, gamma=100, phase='loss')
        G.requires_grad_(True)
        l = torch.mean(loss) * (float('inf') if poison else 1.0)
        D_opt.differentiate(l, D, overlap_exchange=True)
        rec[step + '_bucket_finite'] = bool(torch.isfinite(D.flat_grads).all())
        D_opt.mark_registered(D); D_opt.apply_updates()
        rec[step + '_moved'] = not torch.equal(before, D.flat_params)
        rec[step + '_state_moved'] = any(not torch.equal(st0[k], D_opt._state[k]) for k in st0)
        rec[step + '_overflows'] = D_opt.overflow_count()
        rec[step + '_params'] = D.flat_params.cpu()
    torch.save(rec, os.path.join(outdir, 'rank%d.pt' % rank))


def mode_loop(rank, world, outdir):
    """Two iterations of the real training loop (IMLE refresh sharded over the ranks, rank slices of the global minibatch,
    graphs with the exchange after each replay -- gloo collectives cannot be captured)."""
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    fed = []
    state = dict(n=0)

    def on_iteration(info):
        state['n'] += 1
        return state['n'] >= 2

    res = TL.training_loop(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=FMAP, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=FMAP, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=RES, num_channels=3, label_size=4, label_kind='attributes'),
        sched_args=EasyDict(minibatch_gpu_base=B, minibatch_size_base=B * world),
        tf_config={'rnd.np_random_seed': 1000}, total_kimg=1, data_size=24, init_staleness=10, num_samples_factor=3,
        knn_perturb_factor=0.05, candidate_batch_size=16,
        hooks=dict(on_iteration=on_iteration, on_batch=lambda b: fed.append(b['order_1'].tolist())))
    torch.save(dict(G=res['G'].flat_params.cpu(), D=res['D'].flat_params.cpu(), Gs=res['Gs'].flat_params.cpu(), fed=fed),
               os.path.join(outdir, 'rank%d.pt' % rank))


def mode_record(rank, world, outdir):
    """Five iterations of the real loop with every op recorded (draws, fed slices, loss outputs, final state):
    tests/test_gpu_loop_parity.py replays both ranks' logs into the oracle's two towers."""
    from tests.test_gpu_loop_parity import record_loop, loop_kwargs
    log = record_loop(5, loop_kwargs(512, 3, world=world, data_size=48), keep_state=(rank == 0))
    torch.save(log, os.path.join(outdir, 'rank%d.pt' % rank))


def mode_record5(rank, world, outdir):
    """Two iterations of the real loop at BASELINE config 5's own size (128x128, fmap_base 8192, minibatch_gpu 3, attribute mask) with every
    op recorded, for tests/test_gpu_loop_parity.py::test_config5_two_ranks_at_its_own_size_match_oracle_towers."""
    from tests.test_gpu_loop_parity import record_loop, config5_kwargs
    log = record_loop(2, config5_kwargs(world), keep_state=(rank == 0))
    torch.save(log, os.path.join(outdir, 'rank%d.pt' % rank))


def mode_config5(rank, world, outdir):
    """BASELINE config 5's shape at any world size: minibatch_gpu 3, CelebA-style 40 attribute labels, attribute AND-mask
    (one attribute here: the synthetic labels are Bernoulli(0.2), and an epoch must hold 2 * minibatch matches), a data_size
    divisible by 2 * minibatch_gpu * world (480 = 10 * 48); two iterations of the real loop."""
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL
    from inclusivegan_amd.training import imle
    rec = dict(fed=[], slices=[], assign=[])
    state = dict(n=0)

    def on_iteration(info):
        state['n'] += 1
        return state['n'] >= 2

    def on_op(name, out, feed):
        if name == 'G':
            rec['slices'].append(dict(lat=feed['latents_rec_1'].cpu().numpy().copy(), lab=feed['labels_rec_1'].cpu().numpy().copy()))

    mb_gpu = int(os.environ.get('IGAN_TEST_MB_GPU', '3'))
    res5, fmap5 = int(os.environ.get('IGAN_TEST_RES', RES)), int(os.environ.get('IGAN_TEST_FMAP', '256'))     # 128 / 8192 = config 5's own size
    res = TL.training_loop(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=fmap5, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=fmap5, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=res5, num_channels=3, label_size=40, label_kind='attributes'),
        sched_args=EasyDict(minibatch_gpu_base=mb_gpu, minibatch_size_base=mb_gpu * world),
        tf_config={'rnd.np_random_seed': 1000}, total_kimg=1, data_size=480, init_staleness=10, num_samples_factor=2,
        knn_perturb_factor=0.05, candidate_batch_size=16, attr_interesting='Smiling', attr_names=list(imle.CELEBA_ATTRIBUTES),
        hooks=dict(on_iteration=on_iteration, on_op=on_op, on_batch=lambda b: rec['fed'].append({k: np.array(v) for k, v in b.items() if isinstance(v, np.ndarray)}),
                   on_assignment=lambda i, d: rec['assign'].append((np.array(i), np.array(d)))))
    rec.update(G=res['G'].flat_params.cpu(), D=res['D'].flat_params.cpu(), Gs=res['Gs'].flat_params.cpu())
    torch.save(rec, os.path.join(outdir, 'rank%d.pt' % rank))


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    torch.cuda.set_device(0)
    if world > 1:
        torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    {'exchange': mode_exchange, 'nonfinite': mode_nonfinite, 'loop': mode_loop, 'record': mode_record, 'record5': mode_record5, 'config5': mode_config5}[mode](rank, world, outdir)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
