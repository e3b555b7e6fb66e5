"""KL divergence of the generated Stacked-MNIST class histogram from the uniform distribution (reference: metrics/KL.py:20-52):
    density_fake = histogram(labels, bins = 0..K, density=True);  KL = sum_{p > 0} p * log(p / (1 / K)).
Classifier injected as in mode_counts."""
import numpy as np

from . import metric_base
from .mode_counts import predicted_labels


def kl_to_uniform(labels_all, num_classes):
    density_fake = np.histogram(labels_all, bins=np.arange(num_classes + 1), density=True)[0]
    density_real = np.ones(num_classes, dtype=np.float32) / float(num_classes)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.sum(np.where(density_fake != 0, density_fake * np.log(density_fake / density_real), 0))


class KL(metric_base.MetricBase):
    def __init__(self, num_images, minibatch_per_gpu, classify_fn=None, **kwargs):
        super().__init__(**kwargs)
        self.num_images = num_images
        self.minibatch_per_gpu = minibatch_per_gpu
        self.classify_fn = classify_fn

    def _evaluate(self, Gs, Gs_kwargs, num_gpus):
        labels_all, num_classes = predicted_labels(self, Gs, Gs_kwargs, num_gpus)
        self._report_result(kl_to_uniform(labels_all, num_classes))
