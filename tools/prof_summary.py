#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (grouped, shortened kernel names).
usage: python tools/prof_summary.py <kernel_stats.csv> [top]"""
import csv
import re
import sys


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'at::native::', '', n)
    m = re.match(r'(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|elementwise_kernel|reduce_kernel|distribution_elementwise_grid_stride_kernel)<', n)
    if m:
        f = re.search(r'(CUDAFunctorOnSelf_add|CUDAFunctor_add|MulFunctor|DivFunctor|FillFunctor|rsqrt|sqrt|pow_tensor|direct_copy|sum_functor|MeanOps|normal|uniform|where|softplus|abs|neg|sigmoid|log|exp|clamp|compare|BUnaryFunctor|AUnaryFunctor|BinaryFunctor)[A-Za-z_]*', n)
        return 'torch:%s:%s' % (m.group(1).replace('_kernel', '').replace('vectorized_elementwise', 'vec_elt').replace('elementwise_manual_unroll', 'elt_unroll'), f.group(0) if f else '?')
    return n.split('(')[0][:80]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    agg = {}
    for r in rows:
        k = short(r['Name'])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r['Calls']); a[1] += int(r['TotalDurationNs'])
    tot = sum(v[1] for v in agg.values())
    print('total kernel time %.3f s over %d launches' % (tot / 1e9, sum(v[0] for v in agg.values())))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%6.2f%% %9d calls %9.1f us avg  %s' % (100.0 * t / tot, c, t / c / 1e3, k))


if __name__ == '__main__':
    main()
