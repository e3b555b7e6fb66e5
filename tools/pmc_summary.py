#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs found under a directory.
usage: python tools/pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    filt = sys.argv[2:]
    acc = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if filt and not any(s in k for s in filt):
                continue
            a = acc.setdefault((k, r['Counter_Name']), [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    for (k, c), (n, v) in sorted(acc.items()):
        print('%-60s %-28s n=%-5d avg=%.6g' % (k[:60], c, n, v / n))


if __name__ == '__main__':
    main()
