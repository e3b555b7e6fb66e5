"""Frechet Inception Distance (reference: metrics/frechet_inception_distance.py:19-71).

    mu, sigma = mean / covariance (np.cov, rowvar=False) of the feature activations of `num_images` reals and fakes
    FID = |mu_f - mu_r|^2 + trace(sigma_f + sigma_r - 2 sqrtm(sigma_f sigma_r)),   real part            (:64-71)

The Inception-v3 feature network of the reference (metrics/inception_v3_features.pkl) is not available; the metric takes
`feature_fn(uint8 images [n, C, H, W] on the device) -> [n, F] array / tensor` instead.  Real statistics are cached per
object (the reference caches them in .stylegan2-cache)."""
import numpy as np
import scipy.linalg

from . import metric_base


def fid_from_statistics(mu_real, sigma_real, mu_fake, sigma_fake):
    m = np.square(mu_fake - mu_real).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_fake, sigma_real), disp=False)
    dist = m + np.trace(sigma_fake + sigma_real - 2 * s)
    return np.real(dist)


def statistics(activations):
    return np.mean(activations, axis=0), np.cov(activations, rowvar=False)


def fid_from_activations(act_real, act_fake):
    return fid_from_statistics(*statistics(act_real), *statistics(act_fake))


class FID(metric_base.MetricBase):
    def __init__(self, num_images, minibatch_per_gpu, feature_fn=None, **kwargs):
        super().__init__(**kwargs)
        self.num_images = num_images
        self.minibatch_per_gpu = minibatch_per_gpu
        self.feature_fn = feature_fn
        self._real_stats = None

    def _features(self, images):
        import torch
        f = self.feature_fn(images)
        return f.detach().cpu().numpy() if torch.is_tensor(f) else np.asarray(f)

    def _evaluate(self, Gs, Gs_kwargs, num_gpus):
        import torch
        if self.feature_fn is None:
            raise RuntimeError('FID needs feature_fn: the reference\'s metrics/inception_v3_features.pkl is not available in this tree')
        minibatch_size = num_gpus * self.minibatch_per_gpu
        activations = None
        if self._real_stats is None:
            for idx, images in enumerate(self._iterate_reals(minibatch_size=minibatch_size)):
                begin = idx * minibatch_size
                end = min(begin + minibatch_size, self.num_images)
                f = self._features(torch.from_numpy(images[:end - begin]).to(Gs.device))
                if activations is None:
                    activations = np.empty([self.num_images, f.shape[1]], dtype=np.float32)
                activations[begin:end] = f
                if end == self.num_images:
                    break
            self._real_stats = statistics(activations)
        mu_real, sigma_real = self._real_stats
        activations = np.empty([self.num_images, mu_real.shape[0]], dtype=np.float32)
        for begin in range(0, self.num_images, minibatch_size):
            end = min(begin + minibatch_size, self.num_images)
            activations[begin:end] = self._features(self._generate(Gs, minibatch_size, Gs_kwargs))[:end - begin]
        mu_fake, sigma_fake = statistics(activations)
        self._report_result(fid_from_statistics(mu_real, sigma_real, mu_fake, sigma_fake))
