"""CPU: the snapshot helpers of training/misc.py against golden vectors from the reference's own module (imported in the build
container by tests/golden/make_golden.py -> grid_golden.npz): create_image_grid, convert_to_pil_image, setup_snapshot_image_grid for
every size / layout, apply_mirror_augment, time_to_seconds (including the strings the reference's parser rejects); and the
metrics' label draws (labels must be data-set rows, drawn without touching the loop's NumPy stream)."""
import os

import numpy as np
import pytest
import torch

from inclusivegan_amd.training import misc

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'grid_golden.npz'))


def test_create_image_grid_and_pil_conversion():
    for j in range(5):
        gs = tuple(int(v) for v in G['grid_%d_size' % j])
        out = misc.create_image_grid(G['grid_%d_in' % j], None if gs[0] < 0 else gs)
        assert out.dtype == G['grid_%d_out' % j].dtype and np.array_equal(out, G['grid_%d_out' % j]), j
    for j in range(3):
        img = np.array(misc.convert_to_pil_image(G['pil_%d_in' % j], [float(v) for v in G['pil_%d_drange' % j]]))
        assert img.dtype == np.uint8 and np.array_equal(img, G['pil_%d_out' % j]), j


def test_setup_snapshot_image_grid_every_size_and_layout():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from make_golden import GridSet        # the stand-in data set (inputs only; shared with the generator)
    sizes, layouts = ['1080p', '4k', '8k', 'other'], ['random', 'row_per_class', 'col_per_class', 'class4x4']
    assert int(G['snap_cases']) == 34
    for k in range(int(G['snap_cases'])):
        c, h, w, si, li = [int(v) for v in G['snap_%d_cfg' % k]]
        ts = GridSet([c, h, w])
        (gw, gh), reals, labels = misc.setup_snapshot_image_grid(ts, size=sizes[si], layout=layouts[li])
        assert [int(gw), int(gh), ts.cur] == G['snap_%d_grid' % k].tolist(), k        # grid size AND how far the iterator moved
        assert np.array_equal(reals[:, 0, 0, 0], G['snap_%d_ids' % k]) and np.array_equal(labels, G['snap_%d_labels' % k]), k


def test_mirror_augment_and_time_parser():
    np.random.seed(5)
    assert np.array_equal(misc.apply_mirror_augment(G['mirror_in']), G['mirror_out'])
    for s, want in zip(G['tts_in'], G['tts_out']):
        if np.isnan(want):       # the reference's parser raises on these (single-digit leading field): same behaviour here
            with pytest.raises(ValueError):
                misc.time_to_seconds(str(s))
        else:
            assert misc.time_to_seconds(str(s)) == want, s


def test_resume_kimg_time_reads_the_loops_own_log_line(tmp_path):
    """The tick line training_loop() tees into log.txt is the reference's (training_loop.py:495-504); misc.resume_kimg_time
    (misc.py:147-162) finds the snapshot's kimg and the elapsed time in it.  Without a log.txt: kimg from the name, time 0."""
    from inclusivegan_amd import dnnlib
    line = 'tick %-5d kimg %-8.1f lod %-5.2f minibatch %-4d time %-12s sec/tick %-7.1f sec/kimg %-7.2f maintenance %-6.1f gpumem %.1f' % (
        3, 12.0, 0.0, 6, dnnlib.util.format_time(754), 10.0, 2.5, 0.3, 4.2)
    (tmp_path / 'log.txt').write_text('something else\n' + line + '\n')
    assert misc.resume_kimg_time(str(tmp_path / 'network-snapshot-000012.pkl')) == (12.0, 754.0)
    other = tmp_path / 'elsewhere'
    other.mkdir()
    assert misc.resume_kimg_time(str(other / 'network-snapshot-000007.pkl')) == (7.0, 0.0)


def test_metric_fakes_are_conditioned_on_dataset_label_rows():
    from inclusivegan_amd.metrics.metric_base import MetricBase
    seen = []

    class FakeGs:
        device = torch.device('cpu')
        input_shapes = [[None, 8], [None, 10]]

        def get_output_for(self, latents, labels, **kw):
            seen.append(labels.clone())
            return torch.zeros(latents.shape[0], 3, 4, 4)

    m = MetricBase('probe')
    m._configure('x.pkl', None, dict(resolution=32, num_channels=3, label_size=10, label_kind='onehot', data_size=64))
    np.random.seed(123)
    before = np.random.get_state()[1].copy()
    m._generate(FakeGs(), 16, {})
    m._generate(FakeGs(), 16, {})
    assert np.array_equal(np.random.get_state()[1], before)                        # the loop's host stream was not consumed
    rows = m._get_dataset_obj()._labels
    for lab in seen:
        assert lab.shape == (16, 10) and bool((lab.sum(dim=1) == 1).all())        # one-hot data-set rows, not zeros
        assert all(any(np.array_equal(l.numpy(), r) for r in rows) for l in lab)
    assert not torch.equal(seen[0], seen[1])
    FakeGs.input_shapes = [[None, 8], [None, 0]]
    seen.clear()
    m._generate(FakeGs(), 4, {})
    assert seen[0].shape == (4, 0)
