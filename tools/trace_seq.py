#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_trace.csv into the ordered launch sequence of its last N launches:
one line per launch (start offset us, duration us, grid, short name), for reading which small kernels
sit between which big ones.  usage: python tools/trace_seq.py <kernel_trace.csv> <out.txt> [last_fraction]"""
import csv
import sys

from prof_summary import short


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    rows = rows[int(len(rows) * (1.0 - frac)):]
    t0 = int(rows[0]['Start_Timestamp'])
    with open(sys.argv[2], 'w') as f:
        for r in rows:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            grid = '%sx%sx%s' % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'))
            f.write('%9.1f %7.1f %-16s %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, grid, short(r['Kernel_Name'])))


if __name__ == '__main__':
    main()
