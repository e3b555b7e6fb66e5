#!/usr/bin/env python3
"""Which source lines launch the small torch element-wise / reduce / copy kernels of a training op?  Runs one op's device
work (tools/op_profile.py build) under torch.profiler with Python stacks and aggregates device time of every kernel that is
NOT one of the library's own (names not starting with the csrc kernels) by (kernel family, innermost inclusivegan_amd frame).
usage: python tools/glue_attrib.py {G_train|G_reg|D_train|D_reg} [top]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

import op_profile  # noqa: E402
from prof_summary import short  # noqa: E402

OWN = ('conv_fwd', 'maxpool2x2', 'conv_wgrad_kernel', 'conv_fixup', 'plain_reduce', 'upfirdn2d', 'ban_', 'fba_kernel', 'scale_dot', 'lpips_kernel',
       'mbstd', 'dense_small', 'thin_', 'adam', 'finite_check', 'ema_kernel', 'sumsq', 'bcast_mul', 'row_sqnorm', 'nn1', 'stamp', 'bias_grad')


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else 'G_train'
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    run = op_profile.build(op)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        run()
        torch.cuda.synchronize()
    # map: correlation of launches to python stacks via the CPU op events' stacks
    agg = defaultdict(lambda: [0, 0.0])
    total_own = total_glue = 0.0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        for k in ev.kernels:
            name = short(k.name)
            dur = k.duration
            if any(name.startswith(o) or o in name[:40] for o in OWN):
                total_own += dur
                continue
            total_glue += dur
            frame = next((f for f in (ev.stack or []) if 'inclusivegan_amd' in f and 'tfutil.py' not in f), None)
            if frame is None:       # backward (engine thread) or no Python stack: name the enclosing autograd node / Function instead
                par, chain = ev.cpu_parent, []
                while par is not None:
                    if not par.name.startswith('aten::'):
                        chain.append(par.name.replace('autograd::engine::evaluate_function: ', 'bwd of '))
                    par = par.cpu_parent
                frame = (chain[-1] if chain else 'top level') + '  ' + str(ev.input_shapes)[:60]
            frame = frame.replace(ROOT + '/', '')
            a = agg[(name[:44], ev.name[:30], frame[:110])]
            a[0] += 1; a[1] += dur
    print('%s: own kernels %.2f ms, torch glue kernels %.2f ms (eager, one call)' % (op, total_own / 1e3, total_glue / 1e3))
    for (name, opn, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%8.1f us %4d x  %-44s %-30s %s' % (t, n, name, opn, frame))


if __name__ == '__main__':
    main()
