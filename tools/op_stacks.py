#!/usr/bin/env python3
"""Which source lines launch the torch element-wise glue kernels of a training op: torch profiler with Python stacks,
device time of aten ops grouped by the innermost frame inside this package.
usage: python tools/op_stacks.py {G_train|G_reg|D_train|D_reg} [top]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

import tools.op_profile as OP  # noqa: E402


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else 'G_train'
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    run = OP.build(op)
    run(); run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        run()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.self_device_time_total <= 0:
            continue
        frame = '?'
        for fr in (ev.stack or []):
            if 'inclusivegan_amd' in fr and 'torch/' not in fr:
                frame = fr.replace(ROOT + '/', '')
                break
        a = agg[(ev.name, frame)]
        a[0] += 1
        a[1] += ev.self_device_time_total
    tot = sum(v[1] for v in agg.values())
    print('%s: aten ops with device time: %.1f us total' % (op, tot))
    for (name, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%8.1f us %4d  %-28s %s' % (t, n, name, frame))


if __name__ == '__main__':
    main()
