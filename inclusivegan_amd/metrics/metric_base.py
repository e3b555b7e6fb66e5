"""Common definitions for the quality metrics (reference: metrics/metric_base.py:22-165).

Kept: `MetricBase` (name, `run(network_pkl | Gs, ...)`, `_report_result`, `get_result_str` in the reference's line format,
`_iterate_reals` / `_iterate_fakes`), `MetricGroup`, `DummyMetric`.  What changes: the reference opens three more pickles
(Inception-v3 features, the Stacked-MNIST classifier: .MISSING_LARGE_BLOBS) that are not available here, so every metric
takes the network that turns uint8 images into features / logits as an INJECTED callable (`feature_fn`, `classify_fn`);
the statistics on top of them (the FID formula, mode count, KL to uniform) are the reference's, value for value
(tests/test_metrics.py pins them to the reference's own statements).  Fakes come from Gs on this engine's HIP path.
"""
import os
import time

import numpy as np
import torch

from .. import dnnlib
from ..dnnlib.tflib import tfutil


def convert_images_to_uint8(images, drange=[-1, 1], nchw_to_nhwc=False, shrink=1):
    """float images -> uint8 with the reference's rounding (dnnlib/tflib/tfutil.py:255-267: scale, + 0.5, saturating cast)."""
    images = images.to(torch.float32)
    if shrink > 1:
        images = torch.nn.functional.avg_pool2d(images, shrink, shrink)
    if nchw_to_nhwc:
        images = images.permute(0, 2, 3, 1)
    scale = 255 / (drange[1] - drange[0])
    images = images * scale + (0.5 - drange[0] * scale)
    return images.clamp(0, 255).to(torch.uint8)       # saturate_cast truncates after clamping


class MetricBase:
    def __init__(self, name):
        self.name = name
        self._dataset_obj = None
        self._reset()

    def close(self):
        self._reset()

    def _reset(self, network_pkl=None, run_dir=None, data_dir=None, dataset_args=None, mirror_augment=None):
        if self._dataset_obj is not None:
            self._dataset_obj.close()
        self._network_pkl = network_pkl
        self._data_dir = data_dir
        self._dataset_args = dataset_args
        self._dataset_obj = None
        self._mirror_augment = bool(mirror_augment)
        self._eval_time = 0
        self._results = []

    def run(self, network_pkl, run_dir=None, data_dir=None, dataset_args=None, mirror_augment=None, num_gpus=1, tf_config=None,
            log_results=True, Gs_kwargs=dict(is_validation=True), device=None):
        """`network_pkl`: a snapshot file (the last object of the pickled tuple is Gs, metric_base.py:66) or a live Gs Network."""
        from ..training import misc
        self._reset(network_pkl=network_pkl if isinstance(network_pkl, str) else 'live-network', run_dir=run_dir, data_dir=data_dir,
                    dataset_args=dataset_args, mirror_augment=mirror_augment)
        time_begin = time.time()
        Gs = misc.as_networks(misc.load_pkl(network_pkl), device=device)[-1] if isinstance(network_pkl, str) else network_pkl
        with torch.no_grad():
            self._evaluate(Gs, Gs_kwargs=Gs_kwargs, num_gpus=num_gpus)
        self._eval_time = time.time() - time_begin
        if log_results:
            line = self.get_result_str().strip()
            if run_dir is not None:
                with open(os.path.join(run_dir, 'metric-%s.txt' % self.name), 'a') as f:
                    f.write(line + '\n')
            print(line)

    def get_result_str(self):
        network_name = os.path.splitext(os.path.basename(self._network_pkl))[0]
        if len(network_name) > 29:
            network_name = '...' + network_name[-26:]
        result_str = '%-30s' % network_name
        result_str += ' time %-12s' % dnnlib.util.format_time(self._eval_time)
        for res in self._results:
            result_str += ' ' + self.name + res.suffix + ' '
            result_str += res.fmt % res.value
        return result_str

    def update_autosummaries(self):
        from ..dnnlib.tflib.autosummary import autosummary
        for res in self._results:
            autosummary('Metrics/' + self.name + res.suffix, res.value)

    def _evaluate(self, Gs, Gs_kwargs, num_gpus):
        raise NotImplementedError   # to be overridden by subclasses

    def _report_result(self, value, suffix='', fmt='%-10.4f'):
        self._results += [dnnlib.EasyDict(value=value, suffix=suffix, fmt=fmt)]

    def _get_dataset_obj(self):
        from ..training import dataset
        if self._dataset_obj is None:
            self._dataset_obj = dataset.load_dataset(data_dir=self._data_dir, **self._dataset_args)
        return self._dataset_obj

    def _iterate_reals(self, minibatch_size):
        from ..training import misc
        dataset_obj = self._get_dataset_obj()
        while True:
            images, _labels = dataset_obj.get_minibatch_np(minibatch_size)
            if self._mirror_augment:
                images = misc.apply_mirror_augment(images)
            yield images

    def _generate(self, Gs, minibatch_size, Gs_kwargs, as_uint8=True):
        """One minibatch of fakes from Gs: latents ~ N(0, I) on the device, random labels (metric_base.py:139-146 / the
        per-GPU graphs of the metric files), optionally converted like tflib.convert_images_to_uint8."""
        latents = tfutil.random_normal([minibatch_size] + Gs.input_shapes[0][1:], Gs.device)
        labels = torch.zeros([minibatch_size] + Gs.input_shapes[1][1:], device=Gs.device)
        images = Gs.get_output_for(latents, labels, **Gs_kwargs)
        return convert_images_to_uint8(images) if as_uint8 else images


class MetricGroup:
    def __init__(self, metric_kwarg_list):
        self.metrics = [dnnlib.util.call_func_by_name(**_retarget(kwargs)) for kwargs in metric_kwarg_list]

    def run(self, *args, **kwargs):
        for metric in self.metrics:
            metric.run(*args, **kwargs)

    def get_result_str(self):
        return ' '.join(metric.get_result_str() for metric in self.metrics)

    def update_autosummaries(self):
        for metric in self.metrics:
            metric.update_autosummaries()


def _retarget(kwargs):
    kwargs = dict(kwargs)
    fn = kwargs.get('func_name', '')
    if fn.startswith('metrics.'):
        kwargs['func_name'] = 'inclusivegan_amd.' + fn
    return kwargs


class DummyMetric(MetricBase):
    def _evaluate(self, Gs, Gs_kwargs, num_gpus):
        _ = Gs, Gs_kwargs, num_gpus
        self._report_result(0.0)
