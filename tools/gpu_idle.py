#!/usr/bin/env python3
"""How busy is the device during the steady state of a run?  Reads a rocprofv3 *_kernel_trace.csv, takes the last
`fraction` of the launches (the timed iterations of bench.py), merges overlapping kernel intervals and reports
busy time / wall time, the distribution of the gaps between consecutive kernels, and the kernels that precede the
largest gaps (a host synchronisation shows up as a long gap after one specific kernel).
usage: python tools/gpu_idle.py <kernel_trace.csv> [fraction=0.4] [top=15]"""
import csv
import sys
from collections import Counter

from prof_summary import short


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    rows = rows[int(len(rows) * (1.0 - frac)):]
    iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows]
    wall = max(e for _, e, _ in iv) - iv[0][0]
    busy = 0
    cur_s, cur_e = iv[0][0], iv[0][1]
    gaps = []
    prev_name = iv[0][2]
    for s, e, name in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev_name, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev_name = name
    busy += cur_e - cur_s
    print('%d launches over %.1f ms: device busy %.1f ms = %.1f %%, idle %.1f ms in %d gaps' % (
        len(iv), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps)))
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1000), (1000, 1e9)):
        sel = [g for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
        print('  gaps %5g-%-5g us: %6d  total %8.2f ms' % (lo, hi if hi < 1e9 else float('inf'), len(sel), sum(g[0] for g in sel) / 1e6))
    print('largest gaps (us, after -> before):')
    for g in sorted(gaps, reverse=True)[:top]:
        print('  %9.1f  %-40s -> %s' % (g[0] / 1e3, g[1][:40], g[2][:60]))
    # per-kernel time inside the window (steady state): what the non-conv remainder of a step is made of
    per = Counter(); cnt = Counter()
    for st, e, name in iv:
        per[name] += e - st; cnt[name] += 1
    tot = sum(per.values())
    print('kernel time inside the window: %.1f ms; by kernel:' % (tot / 1e6))
    for name, t in per.most_common(45):
        print('  %6.2f %%  %9.2f ms  %7d launches  %8.1f us avg  %s' % (100.0 * t / tot, t / 1e6, cnt[name], t / cnt[name] / 1e3, name[:90]))
    c = Counter()
    for g in gaps:
        if g[0] > 20e3:
            c[(g[1][:40], g[2][:40])] += g[0]
    print('idle time in gaps > 20 us by (kernel before -> kernel after):')
    for (a, b), t in c.most_common(top):
        print('  %9.2f ms  %-40s -> %s' % (t / 1e6, a, b))


if __name__ == '__main__':
    main()
