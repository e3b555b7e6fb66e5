"""GPU: the callers and data formats either side of the hot path THROUGH the HIP loop:
  f3  a TFRecord directory written with the exporter (the reference's dataset_tool.py layout) feeds training_loop();
  f2  snapshots in the reference's pickle layout; `resume_pkl` restores G / D / Gs bit for bit, reads kimg + elapsed time back
      from the run directory's log.txt (misc.py:147-162) and continues counting images from there (training_loop.py:300);
  f1  `metric_arg_list` is evaluated on every network snapshot and lands in metric-<name>.txt and the tick summaries (:519,522);
  f4  the image grids of the snapshot cadence and the final snapshot (:506-530);
and process_reals (training_loop.py:40-60) against its oracle, including the mirror branch with the coin injected."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
RES = 32


def _write_tfrecords(path, n, rng):
    from inclusivegan_amd.training import tfrecord
    images = rng.randint(0, 256, size=(n, 3, RES, RES)).astype(np.uint8)
    labels = np.zeros((n, 10), np.float32)
    labels[np.arange(n), rng.randint(0, 10, size=n)] = 1
    with tfrecord.TFRecordExporter(path, n, print_progress=False) as tfr:
        order = tfr.choose_shuffled_order()
        for i in order:
            tfr.add_image(images[i])
        tfr.add_labels(labels[order])
    return images[order], labels[order]


def _kwargs(data_dir, run_dir, total_kimg, **extra):
    from inclusivegan_amd.dnnlib import EasyDict
    kw = dict(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', fmap_base=512, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=512, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=2.5),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(tfrecord_dir='smnist_tiny', max_label_size='full'), data_dir=data_dir,
        sched_args=EasyDict(minibatch_gpu_base=6, minibatch_size_base=6), tf_config={'rnd.np_random_seed': 1000},
        metric_arg_list=[EasyDict(func_name='metrics.metric_base.DummyMetric', name='dummy')],
        total_kimg=total_kimg, data_size=96, num_samples_factor=3, init_staleness=10, knn_perturb_factor=0.05, candidate_batch_size=64,
        run_dir=run_dir)
    kw.update(extra)
    return kw


def test_tfrecord_fed_run_with_snapshots_metrics_and_resume(cuda_device, tmp_path):
    from inclusivegan_amd.training import training_loop as TL, misc
    rng = np.random.RandomState(3)
    data_dir = str(tmp_path / 'datasets')
    images, labels = _write_tfrecords(os.path.join(data_dir, 'smnist_tiny'), 96, rng)
    run_dir = str(tmp_path / 'run')
    fed, seen = [], []
    out = TL.training_loop(hooks=dict(on_batch=lambda b: fed.append(b['reals_rec_1'].copy()), on_iteration=lambda i: seen.append(i['cur_nimg']) or False),
                           **_kwargs(data_dir, run_dir, 1))
    # f3: the loop trained on the records (the fed reals are rows of the written images) for 84 iterations = 1.008 kimg
    assert out['cur_nimg'] == 1008 and seen[0] == 12 and len(seen) == 84
    assert fed[0].shape == (6, 3, RES, RES) and all(any(np.array_equal(r.astype(np.uint8), im) for im in images) for r in fed[0])
    files = sorted(os.listdir(run_dir))
    # f4 / f2 / f1: the snapshot cadence (:165-166: every tick at this data size), the final snapshot, the metric file, the log
    for want in ('arb-reals.png', 'arb-fakes-000000.png', 'arb-fakes-000001.png', 'rec-reals.png', 'rec-fakes-000001.png', 'arb-fakes-final.png', 'rec-fakes-final.png',
                 'network-snapshot-000000.pkl', 'network-snapshot-000001.pkl', 'network-final.pkl', 'metric-dummy.txt', 'log.txt'):
        assert want in files, (want, files)
    log = open(os.path.join(run_dir, 'log.txt')).read()
    assert sum(l.startswith('tick ') for l in log.splitlines()) == 2 and 'maintenance' in log and 'Metrics/dummy' in log and 'Loss/loss_G_arb' in log
    metric_lines = open(os.path.join(run_dir, 'metric-dummy.txt')).read().strip().splitlines()
    assert len(metric_lines) == 2 and metric_lines[1].startswith('network-snapshot-000001') and 'dummy 0.0000' in metric_lines[1]
    import PIL.Image
    assert PIL.Image.open(os.path.join(run_dir, 'arb-fakes-final.png')).size == PIL.Image.open(os.path.join(run_dir, 'arb-reals.png')).size

    # f2: resume -- networks bit-identical, kimg / time read back from log.txt, the image counter continues
    pkl = os.path.join(run_dir, 'network-snapshot-000001.pkl')
    kimg, secs = misc.resume_kimg_time(pkl)
    assert kimg == 1.0 and secs >= 0
    G2, D2, Gs2 = misc.as_networks(misc.load_pkl(pkl), device=cuda_device)
    for a, b in ((out['G'], G2), (out['D'], D2), (out['Gs'], Gs2)):
        assert torch.equal(a.flat_params, b.flat_params) and all(torch.equal(a.vars[n], b.vars[n]) for n in a.vars)
    z = np.random.RandomState(1).randn(4, 512).astype(np.float32)
    lab = labels[:4]
    assert np.array_equal(out['Gs'].run(z, lab, is_validation=True, randomize_noise=False), Gs2.run(z, lab, is_validation=True, randomize_noise=False))
    seen2 = []
    run2 = str(tmp_path / 'run2')
    out2 = TL.training_loop(hooks=dict(on_iteration=lambda i: seen2.append(i['cur_nimg']) or len(seen2) >= 3), resume_pkl=pkl, **_kwargs(data_dir, run2, 2))
    assert seen2 == [1012, 1024, 1036]                                    # cur_nimg = int(resume_kimg * 1000) (:300), then + 2 * minibatch per iteration
    assert not torch.equal(out2['G'].flat_params, out['G'].flat_params)   # it trained on
    assert 'kimg 1.0' in open(os.path.join(run2, 'log.txt')).read()


@pytest.mark.parametrize('mirror', [False, True])
def test_process_reals_matches_oracle(cuda_device, mirror):
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import training_loop as TL
    from oracle import training_loop as OT
    rng = np.random.RandomState(4)
    x = rng.randint(0, 256, size=(12, 3, 16, 16)).astype(np.uint8)
    lab = rng.rand(12, 5).astype(np.float32)
    coin = np.array([0.1, 0.9, 0.5, 0.49999, 0.0, 0.75, 0.3, 0.5000001, 0.2, 0.99, 0.6, 0.4], np.float32)
    with tfutil.use_random(tfutil.RandomTape([('uniform', coin)] if mirror else [])):
        y, l = TL.process_reals(torch.from_numpy(x).to(cuda_device), torch.from_numpy(lab).to(cuda_device), 0, mirror, [0, 255], [-1, 1])
    yo, lo = OT.process_reals(x, lab, 0, mirror, [0, 255], [-1, 1], coin=coin)
    assert y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(y.cpu().numpy(), yo) and np.array_equal(l.cpu().numpy(), lo)       # same fp32 scale / bias arithmetic: exact
    if mirror:
        flipped = [i for i in range(12) if not np.array_equal(yo[i], OT.process_reals(x, lab, 0, False, [0, 255], [-1, 1])[0][i])]
        assert flipped == [i for i in range(12) if coin[i] >= 0.5]                            # tf.where(coin < 0.5, x, reverse(x, [3]))
