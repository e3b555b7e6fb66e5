#!/usr/bin/env python3
"""Dense layer shapes of the step (mapping 512->512, D's 8192->512 and 512->1) forward / dgrad / wgrad times.
usage: python tools/dense_bench.py [batch]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from inclusivegan_amd import hip_ops  # noqa: E402
from tools.conv_bench import time_ms  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device('cuda', 0)
    g = hip_ops.ConvGeom(1, 1, 1, 1, 0, 0)
    for (K, N) in [(512, 512), (8192, 512), (512, 1), (512, 128)]:
        x = torch.randn(B, K, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(1, 1, K, N, device=dev) / K ** 0.5
        dy = torch.randn(B, N, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
        tf = time_ms(lambda: hip_ops.conv2d_raw(x, w, g, (1, 1), N), 20)
        td = time_ms(lambda: hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(g), (1, 1), K, w_transposed=True), 20)
        tw = time_ms(lambda: hip_ops.conv2d_wgrad_raw(x, dy, g), 20)
        print('dense %5d -> %4d  batch %2d:  fwd %6.1f us  dgrad %6.1f us  wgrad %6.1f us' % (K, N, B, tf * 1e3, td * 1e3, tw * 1e3))


if __name__ == '__main__':
    main()
