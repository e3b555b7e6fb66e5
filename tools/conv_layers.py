#!/usr/bin/env python3
"""Sustained per-layer rates of the MFMA conv family on the launch shapes of one training iteration (config-e @128: G with its
four batched calls = 24 samples, D with 12 / 24, VGG with 18): each shape runs back to back for `seconds` (the clock needs a
few hundred ms of load to leave idle, so short bursts read low) and reports forward / data-gradient / weight-gradient
TFLOP/s.  Made for same-box A/B runs of an environment switch:
    IGAN_CONV_DMA=0 python tools/conv_layers.py > a.txt; python tools/conv_layers.py > b.txt; paste a.txt b.txt
usage: python tools/conv_layers.py [seconds per measurement = 0.2] [filter substring]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402


def sustained_us(fn, seconds):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    wins = []
    while time.time() - t0 < seconds or len(wins) < 3:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fn()
        e1.record(); torch.cuda.synchronize()
        wins.append(e0.elapsed_time(e1) / 8 * 1e3)
    tail = sorted(wins[len(wins) // 2:])
    return tail[len(tail) // 2]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    dev = torch.device('cuda', 0)
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128}
    layers = []   # (name, N, Cin, H, Cout, K, stride, up, pad, out, scales)
    layers.append(('G 4x4 Conv', 24, 512, 4, 512, 3, 1, 1, 1, 4, True))
    for r in (8, 16, 32, 64, 128):
        layers.append(('G %d Conv0_up' % r, 24, ch[r // 2], r // 2, ch[r], 3, 1, 2, 2, r + 1, True))
        layers.append(('G %d Conv1' % r, 24, ch[r], r, ch[r], 3, 1, 1, 1, r, True))
    for r in (128, 64, 32, 16, 8):
        layers.append(('D %d Conv0' % r, 24, ch[r], r, ch[r], 3, 1, 1, 1, r, False))
        layers.append(('D %d Conv1_down' % r, 24, ch[r], r + 1, ch[r // 2], 3, 2, 1, 0, r // 2, False))
        layers.append(('D %d Skip' % r, 24, ch[r], r - 1, ch[r // 2], 1, 2, 1, 0, r // 2, False))
    layers.append(('D 128 Conv0 (N12)', 12, 128, 128, 128, 3, 1, 1, 1, 128, False))
    for name, c_in, c_out, r in (('VGG conv1_2', 64, 64, 128), ('VGG conv2_2', 128, 128, 64), ('VGG conv3_2', 256, 256, 32), ('VGG conv4_2', 512, 512, 16), ('VGG conv5_2', 512, 512, 8)):
        layers.append((name, 18, c_in, r, c_out, 3, 1, 1, 1, r, False))
    print('%-20s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GFLOP', 'fwd us', 'TF/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
    tot = [0.0, 0.0, 0.0, 0.0]
    for (name, N, Cin, H, Cout, K, stride, up, pad, out, scales) in layers:
        if filt and filt not in name:
            continue
        geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad)
        x = torch.randn(N, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(K, K, Cin, Cout, device=dev) / (K * K * Cin) ** 0.5
        dy = torch.randn(N, Cout, out, out, device=dev).contiguous(memory_format=torch.channels_last)
        s = torch.rand(N, Cin, device=dev) + 0.5 if scales else None
        d = torch.rand(N, Cout, device=dev) + 0.5 if scales else None
        flops = hip_ops.conv_flops(N, H, H, Cin, out, out, Cout, geom)
        t_f = sustained_us(lambda: hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d), seconds)
        t_d = sustained_us(lambda: hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True, in_scale=d), seconds)
        t_w = sustained_us(lambda: hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d), seconds)
        tf = lambda t: flops / t / 1e6
        print('%-20s %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' % (name, flops / 1e9, t_f, tf(t_f), t_d, tf(t_d), t_w, tf(t_w)), flush=True)
        tot[0] += flops; tot[1] += t_f; tot[2] += t_d; tot[3] += t_w
    print('%-20s %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' % ('TOTAL', tot[0] / 1e9, tot[1], tot[0] / tot[1] / 1e6, tot[2], tot[0] / tot[2] / 1e6, tot[3], tot[0] / tot[3] / 1e6))


if __name__ == '__main__':
    main()
