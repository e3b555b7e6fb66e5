#!/usr/bin/env python3
"""Per-layer microbenchmark of the MFMA conv family on the shapes of config-e-Gskip-Dresnet @128x128
(SURVEY.md section 8a): forward / data-gradient / weight-gradient time and algorithmic TFLOP/s.
usage: python tools/conv_bench.py [batch] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402

PEAK = 157.3


def time_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device('cuda', 0)
    layers = []   # (name, N, Cin, H, Cout, K, stride, up, pad, out, scales)
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128}
    layers.append(('G 4x4 Conv', B, 512, 4, 512, 3, 1, 1, 1, 4, True))
    for r in (8, 16, 32, 64, 128):
        layers.append(('G %dx%d Conv0_up' % (r, r), B, ch[r // 2], r // 2, ch[r], 3, 1, 2, 2, r + 1, True))
        layers.append(('G %dx%d Conv1' % (r, r), B, ch[r], r, ch[r], 3, 1, 1, 1, r, True))
    layers.append(('G 128 ToRGB', B, 128, 128, 3, 1, 1, 1, 0, 128, True))
    for r in (128, 64, 32, 16, 8):
        layers.append(('D %dx%d Conv0' % (r, r), 2 * B, ch[r], r, ch[r], 3, 1, 1, 1, r, False))
        layers.append(('D %dx%d Conv1_down' % (r, r), 2 * B, ch[r], r + 1, ch[r // 2], 3, 2, 1, 0, r // 2, False))
        layers.append(('D %dx%d Skip' % (r, r), 2 * B, ch[r], r - 1, ch[r // 2], 1, 2, 1, 0, r // 2, False))
    layers.append(('D 128 FromRGB', 2 * B, 3, 128, 128, 1, 1, 1, 0, 128, False))
    layers.append(('mapping dense', B, 512, 1, 512, 1, 1, 1, 0, 1, False))
    layers.append(('VGG conv1_2 @128', B, 64, 128, 64, 3, 1, 1, 1, 128, False))
    layers.append(('VGG conv3_2 @32', B, 256, 32, 256, 3, 1, 1, 1, 32, False))
    print('%-22s %10s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GFLOP', 'fwd us', 'TF/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
    tot = [0.0, 0.0, 0.0, 0.0]
    for (name, N, Cin, H, Cout, K, stride, up, pad, out, scales) in layers:
        geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad)
        x = torch.randn(N, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(K, K, Cin, Cout, device=dev) / (K * K * Cin) ** 0.5
        dy = torch.randn(N, Cout, out, out, device=dev).contiguous(memory_format=torch.channels_last)
        s = torch.rand(N, Cin, device=dev) + 0.5 if scales else None
        d = torch.rand(N, Cout, device=dev) + 0.5 if (scales and Cout > 4) else None    # ToRGB: demodulate=False
        # algorithmic MACs: every (output pixel, tap) pair that hits a real input sample (or zero padding)
        taps = K * K if up == 1 else ((K + 1) // 2) ** 2 + 2 * ((K + 1) // 2) * (K // 2) + (K // 2) ** 2   # summed over the 4 parity classes
        pix = out * out if up == 1 else ((out + 1) // 2) ** 2  # per class (approx for odd out)
        flops = 2.0 * N * pix * taps * Cin * Cout
        t_f = time_ms(lambda: hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d), reps)
        t_d = time_ms(lambda: hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True, in_scale=d), reps)
        t_w = time_ms(lambda: hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d), reps)
        tf = lambda t: flops / (t * 1e-3) / 1e12
        print('%-22s %10.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' % (name, flops / 1e9, t_f * 1e3, tf(t_f), t_d * 1e3, tf(t_d), t_w * 1e3, tf(t_w)))
        tot[0] += flops; tot[1] += t_f; tot[2] += t_d; tot[3] += t_w
    print('%-22s %10.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f   (peak %.1f)' % (
        'TOTAL', tot[0] / 1e9, tot[1] * 1e3, tot[0] / tot[1] / 1e9, tot[2] * 1e3, tot[0] / tot[2] / 1e9, tot[3] * 1e3, tot[0] / tot[3] / 1e9, PEAK))


if __name__ == '__main__':
    main()
