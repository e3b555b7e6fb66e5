#!/usr/bin/env python3
"""Single-kernel microbenchmarks for roofline evidence (run bare, or under rocprofv3 --pmc ...).
  python tools/kernel_bench.py conv     [batch] [reps]   modulated conv 128x128 (north-star GEMM shape) -> TFLOP/s vs the form's peak (833.3 fp32-equivalent for the default two-piece fp16 form, 416.7 with IGAN_CONV_PLANES=1, 157.3 with =0)
  python tools/kernel_bench.py upfirdn  [batch] [reps]   the three upfirdn2d call sites at 128x128     -> GB/s vs 8 TB/s
  python tools/kernel_bench.py epilogue [batch] [reps]   fused noise+bias+lrelu forward / backward      -> GB/s
  python tools/kernel_bench.py thin     [batch] [reps]   the 3-channel layers (ToRGB, FromRGB, VGG conv1_1 and their data gradients) at the bench's shapes -> GB/s
Algorithmic bytes = (numel_in + numel_out) * 4."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

_FORM = {'0': 0, '1': 1}.get(os.environ.get('IGAN_CONV_PLANES', '2'), 2)      # as csrc/conv2d_mfma.hip planes_mode(): unset or 2 = two fp16 pieces (the default), 1 = three bf16 pieces, 0 = none
PIECE_FORM = _FORM != 0
# fp32-equivalent TFLOP/s each form is priced against (the same figures bench.py uses): the fp16 / bf16 dense peak over the form's piece products
# per fp32 product -- 3 for the two-piece fp16 form, 6 for the three-piece bf16 form -- or the f32 matrix peak
PEAK = {0: 157.3, 1: 2500.0 / 6, 2: 2500.0 / 3}[_FORM]
PIECE_PEAK = PEAK

from inclusivegan_amd import hip_ops  # noqa: E402

HBM_PEAK = 8000.0   # GB/s (spec); ~6300 achievable (MI355X_MICROARCH.md)


def time_ms(fn, reps):
    for _ in range(3):
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
    mode = sys.argv[1] if len(sys.argv) > 1 else 'conv'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device('cuda', 0)
    if mode == 'conv':
        x = torch.randn(B, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
        s = torch.rand(B, 128, device=dev) + 0.5
        d = torch.rand(B, 128, device=dev) + 0.5
        g = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
        ms = time_ms(lambda: hip_ops.conv2d_raw(x, w, g, (128, 128), 128, in_scale=s, out_scale=d), reps)
        fl = 2.0 * B * 128 * 128 * 128 * 128 * 9
        print('modconv 128x128 B=%d: %.1f us  %.1f TFLOP/s  = %.1f %% of %.1f' % (B, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / PEAK * 100, PEAK))
    elif mode == 'upfirdn':
        k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64
        sites = [('G Conv0_up post-filter  [B,129,129,128] pad 1/1 x4', (B, 129, 129, 128), k * 4, 1, 1),
                 ('D Conv1_down pre-filter [2B,128,128,128] pad 2/2', (2 * B, 128, 128, 128), k, 2, 2),
                 ('D Skip pre-filter       [2B,128,128,128] pad 1/1', (2 * B, 128, 128, 128), k, 1, 1)]
        for name, shape, kk, p0, p1 in sites:
            x = torch.randn(*shape, device=dev)
            y = hip_ops.upfirdn2d_raw(x, kk, 1, 1, 1, 1, p0, p1, p0, p1)
            ms = time_ms(lambda: hip_ops.upfirdn2d_raw(x, kk, 1, 1, 1, 1, p0, p1, p0, p1), reps)
            by = (x.numel() + y.numel()) * 4.0
            print('%-52s %7.1f us  %7.1f GB/s = %.1f %% of 8 TB/s' % (name, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / HBM_PEAK * 100))
    elif mode == 'epilogue':
        x = torch.randn(B, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(128, device=dev); noise = torch.randn(B, 1, 128, 128, device=dev); st = torch.tensor(0.3, device=dev)
        y = hip_ops.bias_act_noise_fwd_raw(x, noise, st, b, 3, 0.2, 2 ** 0.5)
        ms = time_ms(lambda: hip_ops.bias_act_noise_fwd_raw(x, noise, st, b, 3, 0.2, 2 ** 0.5), reps)
        print('epilogue fwd  [B,128,128,128] %7.1f us  %7.1f GB/s' % (ms * 1e3, 2 * x.numel() * 4 / ms / 1e6))
        dy = torch.randn_like(x)
        ms = time_ms(lambda: hip_ops.bias_act_noise_bwd_raw(dy, y, noise, 3, 0.2, 2 ** 0.5, True), reps)
        print('epilogue bwd  [B,128,128,128] %7.1f us  %7.1f GB/s' % (ms * 1e3, 3 * x.numel() * 4 / ms / 1e6))
        sc = torch.rand(B, 128, device=dev) + 0.5
        dxs = torch.randn_like(x)
        ms = time_ms(lambda: hip_ops.scale_dot_raw(x, dxs, sc, want_scaled=True), reps)
        print('scale_dot     [B,128,128,128] %7.1f us  %7.1f GB/s  (2 reads + 1 write)' % (ms * 1e3, 3 * x.numel() * 4 / ms / 1e6))
        ms = time_ms(lambda: hip_ops.scale_dot_raw(x, dxs), reps)
        print('channel dot   [B,128,128,128] %7.1f us  %7.1f GB/s  (2 reads)' % (ms * 1e3, 2 * x.numel() * 4 / ms / 1e6))
    elif mode == 'thin':
        # thin-channel layers (csrc/thin_conv.hip) at the bench configuration's shapes: batch = 4 x minibatch_gpu generator samples; bytes = the wide tensor + the thin one
        g1 = hip_ops.ConvGeom(1, 1, 1, 1, 0, 0)
        g3 = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
        N = 4 * B
        cases = [('ToRGB 128x128 C128 -> 3 (modulated)', N, 128, 128, 3, g1, False, True),
                 ('ToRGB 64x64 C256 -> 3 (modulated)', N, 64, 256, 3, g1, False, True),
                 ('ToRGB 32x32 C512 -> 3 (modulated)', N, 32, 512, 3, g1, False, True),
                 ('FromRGB dgrad 128x128 C128 -> 3', N, 128, 128, 3, g1, True, False),
                 ('VGG conv1_1 dgrad 3x3 C64 -> 3', 5 * B, 128, 64, 3, g3, True, False),
                 ('FromRGB 128x128 3 -> C128', N, 128, 3, 128, g1, False, False),
                 ('ToRGB dgrad 128x128 3 -> C128 (modulated)', N, 128, 3, 128, g1, True, True),
                 ('VGG conv1_1 3x3 3 -> C64', 5 * B, 128, 3, 64, g3, False, False)]
        for name, n, r, cin, cout, g, wt, mod in cases:
            x = torch.randn(n, cin, r, r, device=dev).contiguous(memory_format=torch.channels_last)
            w = torch.randn(*((g.kh, g.kw, cout, cin) if wt else (g.kh, g.kw, cin, cout)), device=dev)
            sc = (torch.rand(n, cin, device=dev) + 0.5) if (mod and cin > 4) else None
            osc = (torch.rand(n, cout, device=dev) + 0.5) if (mod and cin <= 4) else None
            fn = lambda: hip_ops.conv2d_raw(x, w, g, (r, r), cout, w_transposed=wt, in_scale=sc, out_scale=osc)
            y = fn()
            ms = time_ms(fn, reps)
            by = (x.numel() + y.numel()) * 4.0
            print('%-44s N%-3d %7.1f us  %7.1f GB/s = %.1f %% of 8 TB/s' % (name, n, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / HBM_PEAK * 100))
    else:
        raise SystemExit(__doc__)


if __name__ == '__main__':
    main()
