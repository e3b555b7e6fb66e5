#!/usr/bin/env python3
"""Are the piece-form kernels bit-reproducible when SEVERAL PROCESSES share the GPU?  (Round 4: under the two-piece fp16 form the 8-ranks-on-one-GPU
bench reported a replay that differed from its eager execution in the last bits, one rank alone never did.)  Starts `procs` children; each repeats
the same forward / data-gradient / weight-gradient calls (same seeds in every child) `rounds` times and reports every digest that differs from its
own first round; the parent also compares the children's first rounds with each other.
usage: [IGAN_CONV_PLANES=2] python tools/planes_contention.py [procs = 8] [rounds = 30]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rounds):
    if os.environ.get('CONTENTION_GRAPHS') == '1':
        return child_graphs(rounds)
    import torch
    from inclusivegan_amd import hip_ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5 + int(os.environ.get('CONTENTION_SEED', '0')))       # CONTENTION_DIFFERENT_DATA=1: every child its own data (same shapes, same addresses)

    def dig(t):
        return hashlib.sha1(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]

    cases = []
    for name, N, Cin, H, Cout, stride, up, pad, out in [('32x32 C512', 24, 512, 32, 512, 1, 1, 1, 32), ('128x128 C128', 12, 128, 128, 128, 1, 1, 1, 128),
                                                        ('up 32->65 C512->256', 8, 512, 32, 256, 1, 2, 2, 65), ('s2 65->32 C256->512', 8, 256, 65, 512, 2, 1, 0, 32),
                                                        ('16x16 C512 sliced', 18, 512, 16, 512, 1, 1, 1, 16)]:
        x = torch.randn(N, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(3, 3, Cin, Cout, generator=g) / (9 * Cin) ** 0.5).to(dev)
        s = (torch.rand(N, Cin, generator=g) + 0.5).to(dev)
        d = (torch.rand(N, Cout, generator=g) + 0.5).to(dev)
        dy = torch.randn(N, Cout, out, out, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        cases.append((name, x, w, s, d, dy, hip_ops.ConvGeom(3, 3, stride, up, pad, pad), out, H, Cin, Cout))
    first = {}
    bad = {}
    keep = {}
    shown = []
    for r in range(rounds):
        for name, x, w, s, d, dy, geom, out, H, Cin, Cout in cases:
            xp = hip_ops.to_pieces(x, s)
            y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d, x_pieces=xp)
            dyp = hip_ops.to_pieces(dy, d)
            dx = hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True, in_scale=d, x_pieces=dyp)
            dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d, x_pieces=xp, dy_pieces=dyp)
            y2 = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d)          # images made by the library in its workspace
            for what, t in (('fwd', y), ('dgrad', dx), ('wgrad', dw), ('fwd(own image)', y2)):
                k = (name, what)
                h = dig(t)
                if k not in first:
                    first[k] = h
                    keep[k] = t.detach().clone()
                elif first[k] != h:
                    bad[k] = bad.get(k, 0) + 1
                    if len(shown) < 12:         # where and how large: bounding box of the differing elements in the logical [N, C, H, W] (or [KH, KW, Cin, Cout]) index space
                        df = (t.detach() != keep[k])
                        idx = df.nonzero()
                        rel = float((t.detach().double() - keep[k].double()).abs().max() / keep[k].double().abs().max())
                        shown.append('%s %s round %d: %d of %d elements differ, max diff / max |value| %.2e, index box lo %s hi %s' % (
                            name, what, r, idx.shape[0], t.numel(), rel, idx.min(0).values.tolist(), idx.max(0).values.tolist()))
    print('FIRST ' + ' '.join('%s' % first[k] for k in sorted(first)))
    print('BAD %d %s' % (sum(bad.values()), sorted(bad.items())))
    for line in shown:
        print('DIFF ' + line)


def child_graphs(rounds):
    """The same calls captured in a hipGraph per shape: every round compares the REPLAY's digests with the first eager execution's."""
    import inclusivegan_amd  # noqa: F401
    import torch
    from inclusivegan_amd import hip_ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)

    def dig(t):
        return hashlib.sha1(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]

    cases = []
    for name, N, Cin, H, Cout, stride, up, pad, out in [('32x32 C512', 24, 512, 32, 512, 1, 1, 1, 32), ('128x128 C128', 12, 128, 128, 128, 1, 1, 1, 128),
                                                        ('64x64 C256', 24, 256, 64, 256, 1, 1, 1, 64), ('up 64->129 C256->128', 12, 256, 64, 128, 1, 2, 2, 129),
                                                        ('s2 129->64 C128->256', 12, 128, 129, 256, 2, 1, 0, 64), ('16x16 C512 sliced', 18, 512, 16, 512, 1, 1, 1, 16)]:
        x = torch.randn(N, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(3, 3, Cin, Cout, generator=g) / (9 * Cin) ** 0.5).to(dev)
        s = (torch.rand(N, Cin, generator=g) + 0.5).to(dev)
        d = (torch.rand(N, Cout, generator=g) + 0.5).to(dev)
        dy = torch.randn(N, Cout, out, out, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        cases.append((name, x, w, s, d, dy, hip_ops.ConvGeom(3, 3, stride, up, pad, pad), out, H, Cin, Cout))

    def run(case):
        name, x, w, s, d, dy, geom, out, H, Cin, Cout = case
        xp = hip_ops.to_pieces(x, s)
        y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d, x_pieces=xp)
        dyp = hip_ops.to_pieces(dy, d)
        dx = hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True, in_scale=d, x_pieces=dyp)
        dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d, x_pieces=xp, dy_pieces=dyp)
        return y, dx, dw

    side = torch.cuda.Stream()
    graphs_ = []
    want = {}
    with torch.cuda.stream(side):
        for case in cases:
            outs = run(case)
            torch.cuda.synchronize()
            for what, t in zip(('fwd', 'dgrad', 'wgrad'), outs):
                want[(case[0], what)] = dig(t)
    pool = None
    for case in cases:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, pool=pool, stream=side):
            outs = run(case)
        pool = pool or gr.pool()
        graphs_.append((case[0], gr, outs))
    bad = {}
    for r in range(rounds):
        for name, gr, outs in graphs_:
            gr.replay()
        torch.cuda.synchronize()
        for name, gr, outs in graphs_:      # all graphs share one pool: digest right after ITS replay
            gr.replay()
            torch.cuda.synchronize()
            for what, t in zip(('fwd', 'dgrad', 'wgrad'), outs):
                if dig(t) != want[(name, what)]:
                    bad[(name, what)] = bad.get((name, what), 0) + 1
    print('FIRST ' + ' '.join(want[k] for k in sorted(want)))
    print('BAD %d %s' % (sum(bad.values()), sorted(bad.items())))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]))
        sys.exit(0)
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    diff = os.environ.get('CONTENTION_DIFFERENT_DATA') == '1'
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--child', str(rounds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                           env=dict(os.environ, CONTENTION_SEED=str(i if diff else 0))) for i in range(procs)]
    outs = [p.communicate(timeout=1500)[0] for p in ps]
    firsts = set()
    for i, o in enumerate(outs):
        lines = o.strip().splitlines()
        f = [ln for ln in lines if ln.startswith('FIRST ')]
        b = [ln for ln in lines if ln.startswith('BAD ')]
        firsts.add(f[0] if f else 'missing')
        print('child %d: %s' % (i, b[0] if b else 'no output'))
        for ln in lines:
            if ln.startswith('DIFF '):
                print('    ' + ln)
    print('form %s, %d processes x %d rounds: children agree on the first round: %s' % (os.environ.get('IGAN_CONV_PLANES', 'default'), procs, rounds, len(firsts) == 1))
