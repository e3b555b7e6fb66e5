"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent fp64 forms of the quality-metric statistics (metrics/frechet_inception_distance.py:64-71,
metrics/mode_counts.py:49, metrics/KL.py:49-52):
  * FID through the eigenvalues of sigma_f sigma_r (trace sqrtm(A) = sum sqrt(eig(A)); the product of two covariance
    matrices has real non-negative eigenvalues) instead of scipy.linalg.sqrtm;
  * mode count and KL from explicit per-class counts instead of np.unique / np.histogram(density=True).
PINNED: tests/golden/metrics_golden.npz holds the values the reference's own statements produce on seeded inputs
(tests/golden/make_metrics_golden.py executes those statements); tests/test_metrics.py checks both this file and the product."""
import numpy as np


def fid(act_real, act_fake):
    a = np.asarray(act_real, dtype=np.float64); b = np.asarray(act_fake, dtype=np.float64)
    mu_r, mu_f = a.mean(0), b.mean(0)
    cr = (a - mu_r).T @ (a - mu_r) / (a.shape[0] - 1)
    cf = (b - mu_f).T @ (b - mu_f) / (b.shape[0] - 1)
    ev = np.linalg.eigvals(cf @ cr)
    tr_sqrt = np.sqrt(np.clip(ev.real, 0, None)).sum()
    return float(((mu_f - mu_r) ** 2).sum() + np.trace(cf) + np.trace(cr) - 2 * tr_sqrt)


def mode_count(labels):
    seen = set(int(l) for l in labels)
    return len(seen)


def kl_to_uniform(labels, num_classes):
    counts = np.zeros(num_classes, dtype=np.float64)
    for l in labels:
        counts[int(l)] += 1
    p = counts / counts.sum()
    q = 1.0 / num_classes
    return float(sum(pi * np.log(pi / q) for pi in p if pi > 0))
