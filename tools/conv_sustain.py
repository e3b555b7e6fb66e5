#!/usr/bin/env python3
"""Burst vs sustained rate of the headline modulated conv (128x128, Cin = Cout = 128, 3x3) at several batches
(tiles per CU: batch 4 -> 2, 6 -> 3, 8 -> 4, 12 -> 6): each shape runs back to back for `seconds`; the first and
the last window of launches are reported (the chip lowers its clock under sustained matrix load).
usage: python tools/conv_sustain.py [seconds] [batches...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

_FORM = {'0': 0, '1': 1}.get(os.environ.get('IGAN_CONV_PLANES', '2'), 2)      # as csrc/conv2d_mfma.hip planes_mode(): unset or 2 = two fp16 pieces (the default), 1 = three bf16 pieces, 0 = none
PIECE_FORM = _FORM != 0
# fp32-equivalent TFLOP/s each form is priced against (the same figures bench.py uses): the fp16 / bf16 dense peak over the form's piece products
# per fp32 product -- 3 for the two-piece fp16 form, 6 for the three-piece bf16 form -- or the f32 matrix peak
PEAK = {0: 157.3, 1: 2500.0 / 6, 2: 2500.0 / 3}[_FORM]
PIECE_PEAK = PEAK

from inclusivegan_amd import hip_ops  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    batches = [int(a) for a in sys.argv[2:]] or [4, 6, 8, 12]
    dev = torch.device('cuda', 0)
    g = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    for B in batches:
        x = torch.randn(B, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
        s = torch.rand(B, 128, device=dev) + 0.5
        d = torch.rand(B, 128, device=dev) + 0.5
        fl = 2.0 * B * 128 * 128 * 128 * 128 * 9
        fn = lambda: hip_ops.conv2d_raw(x, w, g, (128, 128), 128, in_scale=s, out_scale=d)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        import time
        time.sleep(1.0)      # let the clock recover: the first window is an idle-device burst
        wins = []
        t_all = 0.0
        while t_all < seconds:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            wins.append(ms)
            t_all += ms * 10e-3
        tf = lambda ms: fl / ms / 1e9
        tail = sorted(wins[-10:])[len(wins[-10:]) // 2]
        print('modconv 128x128 B=%-2d (%d tiles/CU): first window %.1f us %.1f TFLOP/s | sustained (median of last 10 windows) %.1f us %.1f TFLOP/s = %.1f %% of %.1f'
              % (B, B * 128 * 128 // 128 // 256, wins[0] * 1e3, tf(wins[0]), tail * 1e3, tf(tail), tf(tail) / PEAK * 100, PEAK), flush=True)


if __name__ == '__main__':
    main()
