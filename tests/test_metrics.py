"""Quality-metric statistics: FID formula, Stacked-MNIST mode count, KL to uniform.

CPU: product and oracle against tests/golden/metrics_golden.npz -- values produced by the reference's OWN statements
(metrics/frechet_inception_distance.py:44-45,60-71, mode_counts.py:49, KL.py:49-52 executed by
tests/golden/make_metrics_golden.py).  GPU: the metric classes end to end on the HIP generator with injected feature /
classifier networks."""
import os

import numpy as np
import pytest
import torch

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics_golden.npz'))


@pytest.mark.parametrize('case', ['small', 'wide', 'same'])
def test_fid_formula_against_reference_statements(case):
    from inclusivegan_amd.metrics import frechet_inception_distance as F
    from oracle import metrics as OM
    real, fake, want = GOLD['fid/%s/real' % case], GOLD['fid/%s/fake' % case], float(GOLD['fid/%s/value' % case])
    got = float(F.fid_from_activations(real, fake))
    assert got == want                                        # same NumPy / SciPy statements on the same inputs: bit-identical
    assert abs(OM.fid(real, fake) - want) <= 1e-6 * max(1.0, abs(want))      # independent eigenvalue form agrees


@pytest.mark.parametrize('case', ['spread', 'collapsed', 'all'])
def test_mode_count_and_kl_against_reference_statements(case):
    from inclusivegan_amd.metrics import mode_counts as MC, KL as K
    from oracle import metrics as OM
    labels = GOLD['cls/%s/labels' % case]
    assert MC.count_modes(labels) == int(GOLD['cls/%s/modes' % case]) == OM.mode_count(labels)
    want = float(GOLD['cls/%s/kl' % case])
    assert float(K.kl_to_uniform(labels, 1000)) == want
    assert abs(OM.kl_to_uniform(labels, 1000) - want) < 1e-6


def test_known_answers():
    from inclusivegan_amd.metrics import frechet_inception_distance as F, KL as K, mode_counts as MC
    rng = np.random.RandomState(1)
    a = rng.randn(2000, 8)
    shift = np.array([0.5, 0, 0, -0.25, 0, 0, 0, 0])
    assert abs(F.fid_from_activations(a, a + shift) - (shift ** 2).sum()) < 1e-6       # equal covariances: FID = |mean shift|^2
    assert abs(F.fid_from_activations(a, a)) < 1e-6
    assert abs(K.kl_to_uniform(np.zeros(50), 10) - np.log(10)) < 1e-6                 # one mode of ten
    assert abs(K.kl_to_uniform(np.arange(10).repeat(5), 10)) < 1e-7                   # uniform
    assert MC.count_modes(np.array([3, 3, 7, 999.0])) == 3


def test_uint8_conversion_matches_reference_rounding():
    from inclusivegan_amd.metrics.metric_base import convert_images_to_uint8
    x = torch.tensor([-1.5, -1.0, -0.996, 0.0, 0.5, 0.999, 1.0, 2.0]).reshape(1, 1, 1, 8)
    want = np.clip(x.numpy() * 127.5 + 128.0, 0, 255).astype(np.uint8)                # tfutil.py:265-267: scale, +0.5 - drange[0]*scale, saturate_cast
    assert np.array_equal(convert_images_to_uint8(x).numpy(), want)


def test_result_line_format():
    from inclusivegan_amd.metrics.metric_base import DummyMetric, MetricGroup
    m = DummyMetric(name='dummy')
    m._network_pkl = 'results/00001-run/network-snapshot-012345.pkl'
    m._eval_time = 75
    m._report_result(1.5)
    assert m.get_result_str() == '%-30s' % 'network-snapshot-012345' + ' time %-12s' % '1m 15s' + ' dummy ' + '%-10.4f' % 1.5
    g = MetricGroup([dict(func_name='metrics.metric_base.DummyMetric', name='d1')])
    assert len(g.metrics) == 1 and g.metrics[0].name == 'd1'


@pytest.mark.gpu
def test_metric_classes_on_the_hip_generator(cuda_device, tmp_path):
    """FID / mode count / KL end to end: fakes from Gs on the HIP path (validation mode), injected feature / classifier
    networks (fixed random projections), results equal to the oracle statistics of the very same network outputs; a
    snapshot pickle written by this engine is accepted as `network_pkl`."""
    from inclusivegan_amd.dnnlib import tflib
    from inclusivegan_amd.metrics import frechet_inception_distance as F, mode_counts as MC, KL as K
    from inclusivegan_amd.training import misc
    from inclusivegan_amd.training.dataset import SyntheticDataset
    from oracle import metrics as OM
    dev = cuda_device
    Gs = tflib.Network('Gs', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', num_channels=3,
                       resolution=32, label_size=0, fmap_base=512, device=dev, seed=5)
    proj = torch.randn(3 * 32 * 32, 24, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / 55.0
    seen = dict(real=[], fake=[], logits=[])

    def feature_fn(images):
        assert images.dtype == torch.uint8 and tuple(images.shape[1:]) == (3, 32, 32)
        f = images.float().reshape(images.shape[0], -1) @ proj
        seen['fake' if seen['real_done'] else 'real'].append(f.cpu().numpy())
        return f

    def classify_fn(images):
        assert images.dtype == torch.float32
        l = images.reshape(images.shape[0], -1) @ proj[:, :10]
        seen['logits'].append(l.cpu().numpy())
        return l

    seen['real_done'] = False
    fid = F.FID(num_images=96, minibatch_per_gpu=32, feature_fn=feature_fn, name='fid96')
    orig = fid._generate
    fid._generate = lambda *a, **k: (seen.__setitem__('real_done', True), orig(*a, **k))[1]
    fid.run(Gs, dataset_args=dict(resolution=32, num_channels=3, label_size=0, data_size=128), mirror_augment=False, log_results=False)
    want = OM.fid(np.concatenate(seen['real'])[:96], np.concatenate(seen['fake'])[:96])
    assert abs(fid._results[0].value - want) <= 1e-3 * max(1.0, abs(want))
    mc = MC.mode_counts(num_images=64, minibatch_per_gpu=32, classify_fn=classify_fn, name='modes')
    mc.run(Gs, log_results=False)
    labels = np.argmax(np.concatenate(seen['logits']), axis=1)[:64]
    assert mc._results[0].value == OM.mode_count(labels)
    seen['logits'] = []
    kl = K.KL(num_images=64, minibatch_per_gpu=32, classify_fn=classify_fn, name='kl')
    f = str(tmp_path / 'network-snapshot-000001.pkl')
    misc.save_pkl((Gs, Gs, Gs), f, reference_layout=True)
    kl.run(f, log_results=False, device=dev)
    labels = np.argmax(np.concatenate(seen['logits']), axis=1)[:64]
    assert abs(kl._results[0].value - OM.kl_to_uniform(labels, 10)) < 1e-6
    assert 'network-snapshot-000001' in kl.get_result_str()
