#!/usr/bin/env python3
"""Which kernels of the training step pull the clock down?  Each phase runs ONE kernel shape back to back for ~2 s and
reports its sustained time / TFLOP/s together with the median rocm-smi sclk and socket power of that phase.
usage: python tools/clock_probe2.py"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402

samples = []
stop = False


def smi_loop():
    while not stop:
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5)
            c = json.loads(r.stdout).get('card0', {})
            sclk = [v for k, v in c.items() if 'sclk clock speed' in k]
            pw = [v for k, v in c.items() if 'Power' in k]
            samples.append((time.time(), float(sclk[0].strip('()Mhz')) if sclk else 0.0, float(pw[0]) if pw else 0.0))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.2)


def phase(name, fn, flops, seconds=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    wins = []
    while time.time() - t0 < seconds:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        wins.append(e0.elapsed_time(e1) / 10 * 1e3)
    t1 = time.time()
    tail = sorted(wins[len(wins) // 2:]); med = tail[len(tail) // 2]
    ss = [s for s in samples if t0 + 0.8 < s[0] < t1]
    sclk = np.median([s[1] for s in ss]) if ss else 0
    pw = np.median([s[2] for s in ss]) if ss else 0
    print('%-46s %8.1f us %7.1f TFLOP/s (%4.1f %%)   sclk %4.0f MHz  power %4.0f W' % (name, med, flops / med / 1e6, flops / med / 1e6 / 1.573, sclk, pw), flush=True)


def main():
    global stop
    dev = torch.device('cuda', 0)
    th = threading.Thread(target=smi_loop, daemon=True); th.start()
    g3 = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)

    def mk(B, C, R, Co=None):
        Co = Co or C
        x = torch.randn(B, C, R, R, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(3, 3, C, Co, device=dev) / (9 * C) ** 0.5
        dy = torch.randn(B, Co, R, R, device=dev).contiguous(memory_format=torch.channels_last)
        s = torch.rand(B, C, device=dev) + 0.5
        d = torch.rand(B, Co, device=dev) + 0.5
        return x, w, dy, s, d, 2.0 * B * R * R * C * Co * 9

    x, w, dy, s, d, fl = mk(6, 128, 128)
    phase('modconv fwd 128x128 C128 B6', lambda: hip_ops.conv2d_raw(x, w, g3, (128, 128), 128, in_scale=s, out_scale=d), fl)
    phase('plain conv fwd 128x128 C128 B6', lambda: hip_ops.conv2d_raw(x, w, g3, (128, 128), 128), fl)
    phase('modconv dgrad 128x128 C128 B6', lambda: hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(g3), (128, 128), 128, w_transposed=True, in_scale=d), fl)
    phase('modconv wgrad 128x128 C128 B6', lambda: hip_ops.conv2d_wgrad_raw(x, dy, g3, in_scale=s, out_scale=d), fl)
    x2, w2, dy2, s2, d2, fl2 = mk(12, 128, 128)
    phase('plain conv fwd 128x128 C128 B12 (D)', lambda: hip_ops.conv2d_raw(x2, w2, g3, (128, 128), 128), fl2)
    phase('plain wgrad 128x128 C128 B12 (D)', lambda: hip_ops.conv2d_wgrad_raw(x2, dy2, g3), fl2)
    x3, w3, dy3, s3, d3, fl3 = mk(6, 256, 64)
    phase('modconv fwd 64x64 C256 B6', lambda: hip_ops.conv2d_raw(x3, w3, g3, (64, 64), 256, in_scale=s3, out_scale=d3), fl3)
    phase('modconv wgrad 64x64 C256 B6', lambda: hip_ops.conv2d_wgrad_raw(x3, dy3, g3, in_scale=s3, out_scale=d3), fl3)
    x4, w4, dy4, s4, d4, fl4 = mk(6, 512, 32)
    phase('modconv fwd 32x32 C512 B6', lambda: hip_ops.conv2d_raw(x4, w4, g3, (32, 32), 512, in_scale=s4, out_scale=d4), fl4)
    phase('modconv wgrad 32x32 C512 B6', lambda: hip_ops.conv2d_wgrad_raw(x4, dy4, g3, in_scale=s4, out_scale=d4), fl4)
    x5, w5, dy5, s5, d5, fl5 = mk(18, 64, 128)
    phase('VGG conv1_2 fwd 128x128 C64 B18', lambda: hip_ops.conv2d_raw(x5, w5, g3, (128, 128), 64), fl5)
    # the mix of one G step in miniature: fwd + dgrad + wgrad of the three resolutions
    def mix():
        hip_ops.conv2d_raw(x, w, g3, (128, 128), 128, in_scale=s, out_scale=d)
        hip_ops.conv2d_raw(x3, w3, g3, (64, 64), 256, in_scale=s3, out_scale=d3)
        hip_ops.conv2d_raw(x4, w4, g3, (32, 32), 512, in_scale=s4, out_scale=d4)
        hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(g3), (128, 128), 128, w_transposed=True, in_scale=d)
        hip_ops.conv2d_wgrad_raw(x, dy, g3, in_scale=s, out_scale=d)
        hip_ops.conv2d_wgrad_raw(x3, dy3, g3, in_scale=s3, out_scale=d3)
        hip_ops.conv2d_wgrad_raw(x4, dy4, g3, in_scale=s4, out_scale=d4)
    phase('mix (3 fwd + dgrad + 3 wgrad)', mix, 2 * fl + 2 * fl3 + 2 * fl4 + fl)
    stop = True


if __name__ == '__main__':
    main()
