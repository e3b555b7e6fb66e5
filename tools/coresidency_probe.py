#!/usr/bin/env python3
"""Does a kernel of ANOTHER PROCESS disturb dense_small_kernel?  (Round 5: with eight ranks on one GPU two eager executions of a training op differed, and
the first differing call was a dense layer of the mapping network / of D's head given IDENTICAL inputs -- tools/archive_r05/r5_trace8b.sh; never with one process.)
`victims` processes repeat dense-layer calls (forward M = 48, data gradient M = 24 / 12 / 3, the shapes that differed) and compare every result with their
first; `aggressors` processes run one kind of kernel in a loop meanwhile:
    fwd2 / fwd3 / fwd0   the 3x3 convolution of G 128 Conv1 at N = 6 in the fp16 form / bf16 form / on the fp32 instruction (IGAN_CONV_PLANES per child)
    wgrad2               its weight gradient in the fp16 form
    dense                the victims' own calls (a process of the same kind)
    none                 nothing (victims beside victims only)
usage: [PROBE_VICTIMS=dense,thin,lpips,nn1,scale_dot,smallconv,stream] python tools/coresidency_probe.py <aggressor kind> [victims = 4] [aggressors = 4] [seconds = 20]
(round 6: the victim side covers every small / streaming kernel family of the library, not only dense_small_kernel: see victim())"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def victim(seconds):
    """PROBE_VICTIMS (comma list; default dense): which kernel families this victim process repeats --
       dense      dense_small_kernel (the round-5 victim): forward M = 48, data gradient M = 24 / 12 / 3
       thin       thin_out / thin_in / thin_wgrad (ToRGB, FromRGB and their gradients)
       lpips      lpips_kernel forward / backward through LpipsLayerFn
       nn1        row_sqnorm + the distance GEMM + nn1_fold_kernel (one refresh update)
       scale_dot  scale_dot_kernel (+ final) with and without the scaled output
       smallconv  the fp32-instruction families of the small layers: conv_fwd_dma_kernel, conv_fwd_kernel (ragged Cin), conv_wgrad_kernel with scales
                  (its 8-wave SCM instantiation spills ten registers to scratch: tools/asm_scan.py), conv_fixup / plain_reduce
       stream     upfirdn2d_fir4_kernel, ban_fwd / ban_bwd (+ final), maxpool2x2"""
    import torch
    from inclusivegan_amd import hip_ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(11 + os.getpid() % 7)
    fams = os.environ.get('PROBE_VICTIMS', 'dense').split(',')
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    runs = []
    geom = hip_ops.ConvGeom(1, 1, 1, 1, 0, 0)
    g3 = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    if 'dense' in fams:
        for name, M, K, N, wt in [('fwd 48x512->512', 48, 512, 512, False), ('dgrad 24x512->8192', 24, 512, 8192, True), ('dgrad 12x512->8192', 12, 512, 8192, True),
                                  ('dgrad 3x512->512', 3, 512, 512, True), ('fwd 24x512->512', 24, 512, 512, False)]:
            x = rn(M, K, 1, 1).to(dev)
            w = (rn(1, 1, N, K) if wt else rn(1, 1, K, N)).to(dev) / K ** 0.5
            runs.append(('dense ' + name, lambda x=x, w=w, N=N, wt=wt: hip_ops.conv2d_raw(x, w, hip_ops.dgrad_geom(geom) if wt else geom, (1, 1), N, w_transposed=wt)))
    if 'thin' in fams:
        x = cl(rn(6, 256, 64, 64)); w = rn(1, 1, 256, 3).to(dev); s = (torch.rand(6, 256, generator=g) + 0.5).to(dev)
        runs.append(('thin_out 64x64 C256->3 +s', lambda: hip_ops.conv2d_raw(x, w, geom, (64, 64), 3, in_scale=s)))
        x3 = cl(rn(6, 3, 64, 64)); w3 = rn(1, 1, 3, 128).to(dev)
        runs.append(('thin_in 64x64 3->C128', lambda: hip_ops.conv2d_raw(x3, w3, geom, (64, 64), 128)))
        dy3 = cl(rn(6, 3, 64, 64))
        runs.append(('thin_wgrad 64x64 C256 x 3', lambda: hip_ops.conv2d_wgrad_raw(x, dy3, geom, in_scale=s)))
        xv = cl(rn(6, 64, 64, 64)); wv = rn(3, 3, 3, 64).to(dev)
        runs.append(('thin_out 3x3 C64->3 (dgrad)', lambda: hip_ops.conv2d_raw(xv, wv, hip_ops.dgrad_geom(g3), (64, 64), 3, w_transposed=True)))
    if 'lpips' in fams:
        fa = cl(rn(6, 256, 32, 32)).requires_grad_(True); fb = cl(rn(6, 256, 32, 32)); lin = (torch.rand(256, generator=g) / 256 / 1024).to(dev)
        def lp():
            d = hip_ops.LpipsLayerFn.apply(fa, fb, lin)
            gr, = torch.autograd.grad(d.sum(), [fa])
            return torch.cat([d.reshape(-1), gr.reshape(-1)])
        runs.append(('lpips layer fwd+bwd C256 32x32', lp))
    if 'nn1' in fams:
        q = rn(64, 3072).to(dev); c = rn(256, 3072).to(dev)
        def nn():
            best_d2, best_idx = hip_ops.nn1_state(64, dev)
            hip_ops.nn1_update_raw(q, hip_ops.row_sqnorm_raw(q), c, hip_ops.row_sqnorm_raw(c), best_d2, best_idx, 0)
            return torch.cat([best_d2.double().reshape(-1), best_idx.double().reshape(-1)])
        runs.append(('nn1 64 queries x 256 candidates x 3072', nn))
    if 'scale_dot' in fams:
        a = cl(rn(6, 256, 32, 32)); b = cl(rn(6, 256, 32, 32)); s = (torch.rand(6, 256, generator=g) + 0.5).to(dev)
        def sd():
            dot, scaled = hip_ops.scale_dot_raw(a, b.clone(), s, want_scaled=True)        # the scaled output is written over its input: a fresh copy per call
            return torch.cat([dot.reshape(-1), scaled.reshape(-1)])
        runs.append(('scale_dot +scaled C256 32x32', sd))
        runs.append(('channel dot C256 32x32', lambda: hip_ops.scale_dot_raw(a, b)[0]))
    if 'smallconv' in fams:
        x = cl(rn(6, 512, 8, 8)); w = (rn(3, 3, 512, 512) / 68).to(dev); dy = cl(rn(6, 512, 8, 8))
        s = (torch.rand(6, 512, generator=g) + 0.5).to(dev); d = (torch.rand(6, 512, generator=g) + 0.5).to(dev)
        runs.append(('conv fwd 8x8 C512 +s+d (fp32 DMA tile, sliced)', lambda: hip_ops.conv2d_raw(x, w, g3, (8, 8), 512, in_scale=s, out_scale=d)))
        runs.append(('conv dgrad 8x8 C512', lambda: hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(g3), (8, 8), 512, w_transposed=True, in_scale=d)))
        runs.append(('conv wgrad 8x8 C512 +s+d (SCM, spills)', lambda: hip_ops.conv2d_wgrad_raw(x, dy, g3, in_scale=s, out_scale=d)))
        runs.append(('conv wgrad 8x8 C512 plain', lambda: hip_ops.conv2d_wgrad_raw(x, dy, g3)))
        xr = cl(rn(6, 513, 4, 4)); wr = (rn(3, 3, 513, 512) / 68).to(dev)
        runs.append(('conv fwd 4x4 C513 (ragged)', lambda: hip_ops.conv2d_raw(xr, wr, g3, (4, 4), 512)))
    if 'stream' in fams:
        import numpy as np
        k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64
        xf = rn(6, 65, 65, 128).to(dev)
        runs.append(('upfirdn2d fir4 [6,65,65,128]', lambda: hip_ops.upfirdn2d_raw(xf, k * 4, 1, 1, 1, 1, 1, 1, 1, 1)))
        xb = cl(rn(6, 128, 64, 64)); bb = rn(128).to(dev); nz = rn(6, 1, 64, 64).to(dev); st = torch.tensor(0.3, device=dev)
        yb = hip_ops.bias_act_noise_fwd_raw(xb, nz, st, bb, 3, 0.2, 2 ** 0.5)
        runs.append(('ban_fwd C128 64x64', lambda: hip_ops.bias_act_noise_fwd_raw(xb, nz, st, bb, 3, 0.2, 2 ** 0.5)))
        runs.append(('ban_bwd C128 64x64', lambda: torch.cat([t.reshape(-1) for t in hip_ops.bias_act_noise_bwd_raw(xb, yb, nz, 3, 0.2, 2 ** 0.5, True) if t is not None])))
        runs.append(('pool+tap C128 64x64', lambda: torch.cat([t.reshape(-1) for t in hip_ops.PoolTapFn.apply(xb)])))
    cases = [(name, run, run().clone(), torch.zeros((), device=dev, dtype=torch.int64), torch.zeros((), device=dev, dtype=torch.float64)) for name, run in runs]
    torch.cuda.synchronize()
    t0 = time.time()
    rounds = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            for name, run, first, nbad, worst in cases:
                y = run()
                ne = (y != first)
                nbad += ne.any().to(torch.int64)
                worst.copy_(torch.maximum(worst, ((y.double() - first.double()).abs().max() / first.double().abs().max().clamp_min(1e-300))))
            rounds += 1
        torch.cuda.synchronize()
    print('VICTIM rounds %d: ' % rounds + '; '.join('%s: %d calls differ (max rel %.1e)' % (name, int(nbad), float(worst)) for name, run, first, nbad, worst in cases), flush=True)


def aggressor(kind, seconds):
    import torch
    from inclusivegan_amd import hip_ops
    dev = torch.device('cuda', 0)
    if kind == 'dense':
        return victim(seconds)
    g = torch.Generator().manual_seed(3)
    N, C, H = 6, 128, 128
    x = torch.randn(N, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(3, 3, C, C, generator=g) / (9 * C) ** 0.5).to(dev)
    dy = torch.randn(N, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds + 3:
        for _ in range(10):
            if kind.startswith('wgrad'):
                hip_ops.conv2d_wgrad_raw(x, dy, geom)
            else:
                hip_ops.conv2d_raw(x, w, geom, (H, H), C)
            n += 1
        torch.cuda.synchronize()
    print('AGGRESSOR %s: %d calls' % (kind, n), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == '--victim':
        victim(float(sys.argv[2])); sys.exit(0)
    if sys.argv[1] == '--aggressor':
        aggressor(sys.argv[2], float(sys.argv[3])); sys.exit(0)
    kind = sys.argv[1]
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    na = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
    form = {'fwd2': '2', 'wgrad2': '2', 'fwd3': '1', 'fwd0': '0'}.get(kind)
    env_a = dict(os.environ, **({'IGAN_CONV_PLANES': form} if form else {}))
    if os.environ.get('AGGRESSOR_LIB'):        # a variant build for the aggressors only
        env_a['IGAN_LIB'] = os.environ['AGGRESSOR_LIB']
    for k in [k for k in os.environ if k.startswith('AGGRESSOR_ENV_')]:      # AGGRESSOR_ENV_X=v -> X=v in the aggressors only
        env_a[k[len('AGGRESSOR_ENV_'):]] = os.environ[k]
    env_v = dict(os.environ, **({'IGAN_LIB': os.environ['VICTIM_LIB']} if os.environ.get('VICTIM_LIB') else {}))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--victim', str(seconds)], env=env_v, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(nv)]
    if kind != 'none':
        ps += [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--aggressor', kind, str(seconds)], env=env_a, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(na)]
    print('== aggressor %s: %d victims, %d aggressors, %.0f s' % (kind, nv, 0 if kind == 'none' else na, seconds))
    for p in ps:
        out = p.communicate(timeout=600)[0]
        for ln in out.splitlines():
            if ln.startswith('VICTIM') or ln.startswith('AGGRESSOR'):
                print('   ' + ln)
