"""Number of Stacked-MNIST modes covered by the generator (reference: metrics/mode_counts.py:20-49): classify `num_images`
fakes into the 1000 digit triples and count the distinct classes hit.  The classifier (metrics/stacked_mnist_classifier.pkl
in the reference) is injected: `classify_fn(float images [n, C, H, W] in [-1, 1]) -> logits [n, K]`."""
import numpy as np

from . import metric_base


def predicted_labels(metric, Gs, Gs_kwargs, num_gpus):
    import torch
    if metric.classify_fn is None:
        raise RuntimeError('%s needs classify_fn: the reference\'s metrics/stacked_mnist_classifier.pkl is not available in this tree' % metric.name)
    minibatch_size = num_gpus * metric.minibatch_per_gpu
    labels_all = np.empty([metric.num_images], dtype=np.float32)
    num_classes = None
    for begin in range(0, metric.num_images, minibatch_size):
        end = min(begin + minibatch_size, metric.num_images)
        logits = metric.classify_fn(metric._generate(Gs, minibatch_size, Gs_kwargs, as_uint8=False))     # the classifier sees G's float output (:39-40)
        logits = logits.detach().cpu().numpy() if torch.is_tensor(logits) else np.asarray(logits)
        num_classes = logits.shape[1]
        labels_all[begin:end] = np.argmax(logits, axis=1)[:end - begin]
    return labels_all, num_classes


def count_modes(labels_all):
    return len(np.unique(labels_all))


class mode_counts(metric_base.MetricBase):
    def __init__(self, num_images, minibatch_per_gpu, classify_fn=None, **kwargs):
        super().__init__(**kwargs)
        self.num_images = num_images
        self.minibatch_per_gpu = minibatch_per_gpu
        self.classify_fn = classify_fn

    def _evaluate(self, Gs, Gs_kwargs, num_gpus):
        labels_all, _ = predicted_labels(self, Gs, Gs_kwargs, num_gpus)
        self._report_result(count_modes(labels_all))
