#!/usr/bin/env python3
"""Where does a forward-conv launch spend its time?  Uses the library's diagnostic stamps (igan_debug_set_conv_diag): per
workgroup the ticks at kernel entry, main-loop start, main-loop end and exit.  Reports, for the headline modulated conv at the
given batches: launch span, spread of the workgroup start times, prologue / main loop / epilogue durations (median, max), the
main loop's efficiency against the matrix-pipe time of its chunks, and how the workgroups of one launch overlap in time.
usage: python tools/conv_phases.py [batch ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from inclusivegan_amd import _abi, hip_ops  # noqa: E402


def main():
    loop_stamps = '--loop' in sys.argv or '--prologue' in sys.argv         # library built with -DIGAN_LOOP_STAMPS / -DIGAN_PROLOGUE_STAMPS: per-wave stamps after the per-workgroup ones
    batches = [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [2, 4, 6, 12]
    dev = torch.device('cuda', 0)
    lib = _abi.get_plugin()
    g = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    for B in batches:
        x = torch.randn(B, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
        s = torch.rand(B, 128, device=dev) + 0.5
        d = torch.rand(B, 128, device=dev) + 0.5
        fn = lambda: hip_ops.conv2d_raw(x, w, g, (128, 128), 128, in_scale=s, out_scale=d)
        import time
        t0 = time.time()
        while time.time() - t0 < 0.6:          # bring the clock up
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        tiles = B * 128 * 128 // 128
        diag = torch.zeros(tiles * 4 + (tiles * 32 if loop_stamps else 0), device=dev, dtype=torch.int64)
        lib.igan_debug_set_conv_diag(ctypes.c_void_p(diag.data_ptr()))
        fn()
        torch.cuda.synchronize()
        lib.igan_debug_set_conv_diag(ctypes.c_void_p(0))
        raw = diag.cpu().numpy()
        t = raw[:tiles * 4].reshape(tiles, 4).astype(np.float64) * 0.01      # us
        t0 = t[:, 0].min()
        span = t[:, 3].max() - t0
        pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        ideal_alone = 36 * 32 * 2 * 64 / 2400.0          # us: one tile's MFMAs on its 4 SIMDs (2 waves each) at 2.4 GHz
        print('B=%d: %d workgroups, launch span %.1f us' % (B, tiles, span))
        print('   start times: first %.1f, median %.1f, last %.1f us after the first' % (0.0, np.median(t[:, 0] - t0), (t[:, 0] - t0).max()))
        print('   prologue  median %.1f max %.1f us | main loop median %.1f min %.1f max %.1f us | epilogue median %.1f max %.1f us' % (
            np.median(pro), pro.max(), np.median(loop), loop.min(), loop.max(), np.median(epi), epi.max()))
        print('   main loop: a tile alone needs %.1f us of matrix-pipe time, a pair sharing a CU 2x that' % ideal_alone)
        # concurrency profile: how many workgroups are inside their main loop over time
        edges = np.linspace(t0, t0 + span, 21)
        active = [int(((t[:, 1] <= e) & (t[:, 2] > e)).sum()) for e in edges]
        print('   workgroups inside the main loop at 5 % steps of the span:', active)
        if loop_stamps:
            ph = raw[tiles * 4:].reshape(tiles, 8, 4).astype(np.float64) / 36.0      # cycles per chunk
            med = np.median(ph.reshape(-1, 4), axis=0)
            print('   per wave and chunk (shader cycles, median over all waves): first MFMA half %.0f | stage -> LDS (incl. wait for the global loads) %.0f | second MFMA half %.0f | barrier %.0f | total %.0f (ideal: 2 waves x 32 MFMAs x 64 = 4096)' % (med[0], med[1], med[2], med[3], med.sum()))
            w = np.median(ph, axis=0)
            for wv in range(8):
                print('      wave %d: %6.0f %6.0f %6.0f %6.0f' % (wv, w[wv, 0], w[wv, 1], w[wv, 2], w[wv, 3]))
        if '--prologue' in sys.argv:      # library built with -DIGAN_PROLOGUE_STAMPS: per-wave 100 MHz stamps inside the prologue
            pw = raw[tiles * 4:].reshape(tiles, 8, 4).astype(np.float64) * 0.01
            start = t[:, 0][:, None]
            print('   per wave, us after the workgroup\'s first stamp (median over workgroups): tables start %s | loads issued %s | own loads landed %s | barrier passed %s' % tuple(
                np.array2string(np.median(pw[:, :, k] - start, axis=0), precision=1, separator=' ') for k in range(4)))
        first_round = np.sort(t[:, 0] - t0)[:512]
        print('   start of the first %d workgroups spans %.1f us' % (len(first_round), first_round.max()))


if __name__ == '__main__':
    main()
