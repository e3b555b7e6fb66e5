"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's own runnable
Python (and its DCI C library built into oracle/_ref/) in the build container.  The reference code
never travels to the GPU box; only the arrays / JSON written here do.

Run from the repo root:  python tests/golden/make_golden.py   (needs /root/reference)

Fixtures:
  misc_golden.npz          training/misc.py: slerp, normalize, adjust_dynamic_range; dnnlib.util.format_time
  run_training_golden.json run_training.run(...) kwargs for the five BASELINE.json configs
                           (dnnlib.submit_run monkey-patched to capture instead of launching)
  dci_golden.npz           seeded (data, queries) + the reference DCI library's (idx, dist) with the
                           training-time parameters (training_loop.py:197,368,398)
  grid_golden.npz          training/misc.py: create_image_grid, convert_to_pil_image, setup_snapshot_image_grid (every size /
                           layout) on a small stand-in data set, apply_mirror_augment, time_to_seconds (python make_golden.py grids)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def make_misc():
    sys.path.insert(0, REF)
    from training import misc            # reference module (NumPy only on these paths)
    import dnnlib
    rng = np.random.RandomState(123)
    a = rng.randn(7, 512).astype(np.float32)
    b = rng.randn(7, 512).astype(np.float32)
    t_scalar = 0.05
    t_vec = rng.rand(7, 1).astype(np.float32)
    img = rng.randint(0, 256, size=(3, 3, 8, 8)).astype(np.float32)
    secs = [0, 1, 59, 60, 61, 3599, 3600, 86399, 86400, 90061, 123456.7]
    np.savez(os.path.join(HERE, 'misc_golden.npz'),
             a=a, b=b, t_vec=t_vec, img=img,
             slerp_scalar=misc.slerp(a, b, t_scalar), slerp_vec=misc.slerp(a, b, t_vec),
             normalize=misc.normalize(a),
             adr_255_to_pm1=misc.adjust_dynamic_range(img, [0, 255], [-1, 1]),
             adr_pm1_to_255=misc.adjust_dynamic_range(img / 127.5 - 1, [-1, 1], [0, 255]),
             secs=np.array(secs), format_time=np.array([dnnlib.util.format_time(s) for s in secs]))


def make_run_training():
    sys.path.insert(0, REF)
    import dnnlib
    import run_training
    captured = {}

    def fake_submit_run(submit_config, run_func_name, **kwargs):
        out = dict(run_func_name=run_func_name, num_gpus=submit_config.num_gpus, run_desc=submit_config.run_desc)
        for k, v in kwargs.items():
            out[k] = json.loads(json.dumps(v, default=lambda o: dict(o)))
        captured['cfg'] = out

    dnnlib.submit_run = fake_submit_run
    run_training.dnnlib.submit_run = fake_submit_run
    common = dict(data_dir='.', result_dir='results', gamma=None, mirror_augment=False, metrics=[], resume_pkl=None,
                  num_epochs=10000, init_proj_dim=None, init_staleness=10, num_samples_factor=10, knn_perturb_factor=0.05,
                  candidate_batch_size=256, exclusive_retrieved_code=0, NN_rec_lpips_weight=2.5, dist_thres_percentile=100.0,
                  init_mul=1.0)
    cases = {
        'cfg0_smnist_cpu':   dict(dataset='stacked_mnist_240k', config_id='config-e-Gskip-Dresnet', num_gpus=1, minibatch_gpu=6, data_size=240000, attr_interesting=None),
        'cfg1_smnist_1gpu':  dict(dataset='stacked_mnist_240k', config_id='config-e-Gskip-Dresnet', num_gpus=1, minibatch_gpu=6, data_size=240000, attr_interesting=None),
        'cfg2_celeba_adv':   dict(dataset='celeba_align_png_cropped_30k', config_id='config-e-Gskip-Dresnet', num_gpus=1, minibatch_gpu=6, data_size=30000, attr_interesting=None, NN_rec_lpips_weight=0.0),
        'cfg3_celeba_2gpu':  dict(dataset='celeba_align_png_cropped_30k', config_id='config-e-Gskip-Dresnet', num_gpus=2, minibatch_gpu=6, data_size=30000, attr_interesting=None),
        'cfg4_celeba_8gpu':  dict(dataset='celeba_align_png_cropped_30k', config_id='config-e-Gskip-Dresnet', num_gpus=8, minibatch_gpu=3, data_size=30000, attr_interesting='Bald,Eyeglasses'),
    }
    out = {}
    for name, over in cases.items():
        kw = dict(common)
        kw.update(over)
        run_training.run(**kw)
        out[name] = dict(args=kw, kwargs=captured['cfg'])
    with open(os.path.join(HERE, 'run_training_golden.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


def make_dci():
    sys.path.insert(0, REPO)
    from oracle import dci_ref
    assert dci_ref.available(), 'build oracle/_ref first: make -C oracle'
    rng = np.random.RandomState(7)
    # low intrinsic dimension like dci_code/src/util.c:gen_data
    lat = rng.uniform(-1, 1, size=(2000, 6))
    tr = rng.uniform(-1, 1, size=(6, 48))
    data = lat @ tr
    queries = rng.uniform(-1, 1, size=(40, 6)) @ tr
    d = dci_ref.DCIRef(48, 3, 15)
    d.add(data, num_levels=3, field_of_view=10, prop_to_retrieve=0.002)
    idx, dist = d.query(queries, num_neighbours=1, field_of_view=200, prop_to_retrieve=1.0)
    d.close()
    np.savez(os.path.join(HERE, 'dci_golden.npz'), data=data, queries=queries, idx=idx[:, 0], dist=dist[:, 0])


class GridSet:
    """Stand-in for the training set object setup_snapshot_image_grid walks (shape, dtype, label_size, label_dtype,
    get_minibatch_np): image i carries i in every pixel (mod 251) and the label one_hot(LABEL_SEQ[i])."""
    LABEL_SEQ = [0, 2, 1, 1, 0, 2, 2, 0, 1, 0, 1, 2, 2, 1, 0, 0, 0, 1, 2, 2, 1, 1, 0, 2]

    def __init__(self, shape, label_size=3):
        self.shape, self.dtype, self.label_size, self.label_dtype, self.cur = list(shape), 'uint8', label_size, 'float32', 0

    def get_minibatch_np(self, n):
        idx = (self.cur + np.arange(n)) % 240
        self.cur += n
        imgs = np.broadcast_to((idx % 251).astype(np.uint8)[:, None, None, None], [n] + self.shape).copy()
        labels = np.zeros((n, self.label_size), np.float32)
        labels[np.arange(n), [self.LABEL_SEQ[i % len(self.LABEL_SEQ)] % self.label_size for i in idx]] = 1
        return imgs, labels


def make_grids():
    sys.path.insert(0, REF)
    from training import misc
    rng = np.random.RandomState(321)
    out = {}
    for j, (shape, gs) in enumerate([((7, 3, 5, 4), None), ((7, 3, 5, 4), (4, 2)), ((5, 6, 6), None), ((1, 1, 4, 4), None), ((12, 3, 2, 3), (3, 5))]):
        imgs = rng.rand(*shape).astype(np.float32)
        out['grid_%d_in' % j] = imgs
        out['grid_%d_size' % j] = np.array(gs if gs is not None else (-1, -1))
        out['grid_%d_out' % j] = misc.create_image_grid(imgs, gs)
    for j, (img, drange) in enumerate([(rng.rand(3, 6, 5) * 2 - 1, [-1, 1]), (rng.rand(1, 6, 5) * 255, [0, 255]), (rng.rand(6, 5), [0, 1])]):
        out['pil_%d_in' % j], out['pil_%d_drange' % j] = img, np.array(drange)
        out['pil_%d_out' % j] = np.array(misc.convert_to_pil_image(img, drange))
    k = 0
    for shape in ([1, 540, 640], [3, 128, 128], [3, 32, 32]):
        for size in ('1080p', '4k', '8k', 'other'):
            for layout in ('random', 'row_per_class', 'col_per_class', 'class4x4'):
                if shape[1] < 128 and (size != '1080p' or layout not in ('random', 'row_per_class')):
                    continue        # 32x32 grids are 32 x 32 cells: two cases are enough
                ts = GridSet(shape)
                (gw, gh), reals, labels = misc.setup_snapshot_image_grid(ts, size=size, layout=layout)
                out['snap_%d_cfg' % k] = np.array([shape[0], shape[1], shape[2], ['1080p', '4k', '8k', 'other'].index(size),
                                                  ['random', 'row_per_class', 'col_per_class', 'class4x4'].index(layout)])
                out['snap_%d_grid' % k] = np.array([gw, gh, ts.cur])
                out['snap_%d_ids' % k] = reals[:, 0, 0, 0].copy()
                out['snap_%d_labels' % k] = labels
                k += 1
    out['snap_cases'] = np.array(k)
    np.random.seed(5)
    mb = rng.randint(0, 256, size=(9, 3, 4, 6)).astype(np.uint8)
    out['mirror_in'], out['mirror_out'] = mb, misc.apply_mirror_augment(mb)
    strings = ['0s', '59s', '1m 00s', '12m 34s', '1h 00m 00s', '9h 59m 59s', '1d 00h 00m', '12d 23h 59m']
    def tts(x):        # the reference's parser raises on single-digit seconds ('5s'): recorded as NaN (error behaviour is part of the surface)
        try:
            return misc.time_to_seconds(x)
        except ValueError:
            return np.nan
    out['tts_in'], out['tts_out'] = np.array(strings + ['5s']), np.array([tts(x) for x in strings + ['5s']])
    np.savez_compressed(os.path.join(HERE, 'grid_golden.npz'), **out)


if __name__ == '__main__':
    os.chdir(REF)
    which = sys.argv[1:] or ['misc', 'run_training', 'dci', 'grids']
    for name, fn in (('misc', make_misc), ('run_training', make_run_training), ('dci', make_dci), ('grids', make_grids)):
        if name in which:
            fn()
    print('golden fixtures written to', HERE)
