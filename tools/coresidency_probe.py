#!/usr/bin/env python3
"""Does a kernel of ANOTHER PROCESS disturb dense_small_kernel?  (Round 5: with eight ranks on one GPU two eager executions of a training op differed, and
the first differing call was a dense layer of the mapping network / of D's head given IDENTICAL inputs -- tools/r5_trace8b.sh; never with one process.)
`victims` processes repeat dense-layer calls (forward M = 48, data gradient M = 24 / 12 / 3, the shapes that differed) and compare every result with their
first; `aggressors` processes run one kind of kernel in a loop meanwhile:
    fwd2 / fwd3 / fwd0   the 3x3 convolution of G 128 Conv1 at N = 6 in the fp16 form / bf16 form / on the fp32 instruction (IGAN_CONV_PLANES per child)
    wgrad2               its weight gradient in the fp16 form
    dense                the victims' own calls (a process of the same kind)
    none                 nothing (victims beside victims only)
usage: python tools/coresidency_probe.py <aggressor kind> [victims = 4] [aggressors = 4] [seconds = 20]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def victim(seconds):
    import torch
    from inclusivegan_amd import hip_ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(11 + os.getpid() % 7)
    geom = hip_ops.ConvGeom(1, 1, 1, 1, 0, 0)
    cases = []
    for name, M, K, N, wt in [('fwd 48x512->512', 48, 512, 512, False), ('dgrad 24x512->8192', 24, 512, 8192, True), ('dgrad 12x512->8192', 12, 512, 8192, True),
                              ('dgrad 3x512->512', 3, 512, 512, True), ('fwd 24x512->512', 24, 512, 512, False)]:
        x = torch.randn(M, K, 1, 1, generator=g).to(dev)
        w = (torch.randn(1, 1, N, K, generator=g) if wt else torch.randn(1, 1, K, N, generator=g)).to(dev) / K ** 0.5
        run = (lambda x=x, w=w, N=N, wt=wt: hip_ops.conv2d_raw(x, w, hip_ops.dgrad_geom(geom) if wt else geom, (1, 1), N, w_transposed=wt))
        cases.append((name, run, run().clone(), torch.zeros((), device=dev, dtype=torch.int64), torch.zeros((), device=dev, dtype=torch.float64)))
    torch.cuda.synchronize()
    t0 = time.time()
    rounds = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            for name, run, first, nbad, worst in cases:
                y = run()
                ne = (y != first)
                nbad += ne.any().to(torch.int64)
                worst.copy_(torch.maximum(worst, ((y.double() - first.double()).abs().max() / first.double().abs().max())))
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
