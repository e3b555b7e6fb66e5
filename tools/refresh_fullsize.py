#!/usr/bin/env python3
"""ONE IMLE refresh at the full size of the CelebA configuration (training_loop.py:353-406 with run_training.py's defaults):
data_size 30 000 reals, num_samples_factor 10 -> 300 000 candidates of 49 152 dims, candidate_batch_size 256, generated
by a random-init config-e generator and assigned by the exact on-device 1-NN.  Prints a JSON line with the wall time and its
split (the reference materialises a 118 GB fp64 candidate array on the host and indexes it with DCI for this step).
usage: python tools/refresh_fullsize.py [data_size=30000] [num_samples_factor=10]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from inclusivegan_amd.dnnlib import tflib  # noqa: E402
from inclusivegan_amd.training import dataset, training_loop as TL  # noqa: E402


def main():
    data_size = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    factor = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device('cuda', 0)
    np.random.seed(1000)
    ts = dataset.SyntheticDataset(resolution=128, num_channels=3, label_size=40, data_size=data_size, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', num_channels=3, resolution=128,
                      label_size=40, fmap_base=8192, device=dev, seed=1001)
    latents = np.random.randn(data_size * factor, 512).astype(np.float32)
    labels = ts.get_random_labels_np(data_size * factor)
    # warm up kernels / clocks on a small refresh
    small = dataset.SyntheticDataset(resolution=128, num_channels=3, label_size=40, data_size=48, device=dev)
    TL.imle_refresh(G, small, latents[:512], labels[:512], 48, 6, 256, [-1, 1], dev)
    torch.cuda.synchronize()
    t0 = time.time()
    idx, dist = TL.imle_refresh(G, ts, latents, labels, data_size, 6, 256, [-1, 1], dev)
    torch.cuda.synchronize()
    dt = time.time() - t0
    gen_flops = data_size * factor * 2 * 1.126e10          # G forward MACs / image
    nn_flops = 2.0 * data_size * data_size * factor * 49152
    print(json.dumps(dict(imle_refresh_s_fullsize=round(dt, 2), data_size=data_size, candidates=data_size * factor, dim=49152,
                          generator_flops=gen_flops, nn_gemm_flops=nn_flops, tflops_overall=round((gen_flops + nn_flops) / dt / 1e12, 1),
                          unique_winners=int(len(np.unique(idx))), mean_dist=float(dist.mean()),
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))))


if __name__ == '__main__':
    main()
