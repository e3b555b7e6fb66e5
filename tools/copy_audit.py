#!/usr/bin/env python3
"""Which calls make hip_ops.nhwc() / .contiguous() actually copy (layout conversions on the hot path): wraps the helpers,
runs one training op eagerly and prints (shape, strides, caller) of every real copy.
usage: python tools/copy_audit.py {G_train|D_train|G_reg|D_reg}"""
import os
import sys
import traceback
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inclusivegan_amd import hip_ops  # noqa: E402
import tools.op_profile as OP  # noqa: E402


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else 'G_train'
    run = OP.build(op)
    run(); run()
    torch.cuda.synchronize()
    seen = Counter()
    orig = hip_ops.nhwc

    def audited(x):
        if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
            fr = [f for f in traceback.extract_stack()[:-1] if 'inclusivegan_amd' in f.filename]
            where = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:])
            seen[(tuple(x.shape), tuple(x.stride()), where)] += 1
        return orig(x)

    hip_ops.nhwc = audited
    run()
    torch.cuda.synchronize()
    hip_ops.nhwc = orig
    for (shape, stride, where), n in sorted(seen.items(), key=lambda kv: -kv[1] * int(torch.tensor(kv[0][0]).prod())):
        print('%3d x %-22s strides %-28s %s' % (n, shape, stride, where))


if __name__ == '__main__':
    main()
