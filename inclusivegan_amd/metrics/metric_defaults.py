"""Default metric definitions (reference: metrics/metric_defaults.py:13-29) for the metrics built here; the remaining
entries of the reference table (is50k, ppl_*, ls, pr50k3) need networks that are not available and are not offered."""
from ..dnnlib import EasyDict

metric_defaults = EasyDict([(args.name, args) for args in [
    EasyDict(name='mode_counts_24k', func_name='metrics.mode_counts.mode_counts', num_images=24000, minibatch_per_gpu=32),
    EasyDict(name='KL24k', func_name='metrics.KL.KL', num_images=24000, minibatch_per_gpu=32),
    EasyDict(name='fid30k', func_name='metrics.frechet_inception_distance.FID', num_images=30000, minibatch_per_gpu=8),
]])
