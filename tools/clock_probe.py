#!/usr/bin/env python3
"""What clock does the headline conv get in different company?  Times the modulated 128x128 conv (B = 6) over ~2.5 s of
(a) back-to-back launches, (b) launches interleaved with an HBM-bound kernel (upfirdn2d, ~15 % / ~40 % of the time),
(c) launches separated by host-side idle gaps, (d) inside a hipGraph replay loop; rocm-smi clocks / power are sampled
alongside.  usage: python tools/clock_probe.py"""
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
            samples.append((time.time(), r.stdout.strip()))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), 'ERR %s' % e))
        time.sleep(0.25)


def main():
    global stop
    dev = torch.device('cuda', 0)
    B = 6
    g = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    x = torch.randn(B, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
    s = torch.rand(B, 128, device=dev) + 0.5
    d = torch.rand(B, 128, device=dev) + 0.5
    conv = lambda: hip_ops.conv2d_raw(x, w, g, (128, 128), 128, in_scale=s, out_scale=d)
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64
    xu = torch.randn(2 * B, 128, 128, 128, device=dev)
    fir = lambda: hip_ops.upfirdn2d_raw(xu, k, 1, 1, 1, 1, 2, 2, 2, 2)
    fl = 2.0 * B * 128 * 128 * 128 * 128 * 9
    th = threading.Thread(target=smi_loop, daemon=True)
    th.start()

    def timed_conv(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            conv()
        e1.record()
        return e0, e1, n

    def phase(name, body, seconds=2.5):
        conv(); fir(); torch.cuda.synchronize()
        t0 = time.time()
        evs = []
        while time.time() - t0 < seconds:
            evs += body()
            torch.cuda.synchronize()
        us = [a.elapsed_time(b) / n * 1e3 for a, b, n in evs]
        tail = sorted(us[len(us) // 2:])
        med = tail[len(tail) // 2]
        print('%-58s conv %.1f us  %.1f TFLOP/s (median of the second half, %d windows)' % (name, med, fl / med / 1e6, len(us)), flush=True)
        samples.append((time.time(), 'PHASE_END ' + name))

    phase('(a) conv back to back', lambda: [timed_conv(10)])

    def mix(nfir):
        def body():
            out = []
            for _ in range(4):
                out.append(timed_conv(5))
                for _ in range(nfir):
                    fir()
            return out
        return body
    phase('(b1) 5 convs + 2 FIR passes (HBM-bound, ~15 % of time)', mix(2))
    phase('(b2) 5 convs + 8 FIR passes (~40 % of time)', mix(8))

    def gaps():
        out = [timed_conv(5)]
        torch.cuda.synchronize()
        time.sleep(0.0015)
        return out
    phase('(c) 5 convs, then 1.5 ms of idle device', gaps)

    # (d) graph replay of [5 convs + 2 FIR]
    gph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            conv(); fir()
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=st):
            for _ in range(5):
                conv()
            for _ in range(2):
                fir()
    torch.cuda.synchronize()
    t_fir = None
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fir()
    e1.record(); torch.cuda.synchronize()
    t_fir = e0.elapsed_time(e1) / 20 * 1e3
    t0 = time.time()
    tot = []
    while time.time() - t0 < 2.5:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gph.replay()
        e1.record(); torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e1) / 10 * 1e3)
    tail = sorted(tot[len(tot) // 2:]); med = tail[len(tail) // 2]
    print('(d) hipGraph [5 convs + 2 FIR] replay: %.1f us per replay; FIR alone %.1f us -> conv ~%.1f us' % (med, t_fir, (med - 2 * t_fir) / 5), flush=True)
    samples.append((time.time(), 'PHASE_END (d)'))
    stop = True
    th.join(timeout=3)
    print('--- rocm-smi samples ---')
    import json
    for t, sline in samples:
        if sline.startswith('PHASE_END') or sline.startswith('ERR'):
            print('%.2f %s' % (t, sline))
            continue
        try:
            j = json.loads(sline)
            c = j.get('card0', {})
            keep = {k2: v for k2, v in c.items() if 'sclk' in k2.lower() or 'mclk' in k2.lower() or 'power' in k2.lower() or 'fclk' in k2.lower()}
            print('%.2f %s' % (t, keep))
        except Exception:  # noqa: BLE001
            print('%.2f RAW %s' % (t, sline[:200]))


if __name__ == '__main__':
    main()
