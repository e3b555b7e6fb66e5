#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for the M0 hazard of LDS-DMA: `buffer_load_* ... lds` reads M0 (the LDS base address) and the
ISA requires ONE wait state between a scalar-ALU write of M0 and the instruction that reads it (the compiler's own `s_mov_b32 m0, sN; s_nop 0;
buffer_load ... lds` sequences show the rule).  Reports, per kernel, every LDS-DMA instruction whose immediately preceding instruction writes M0.
usage: m0_hazard_scan.py file.s [kernel-name substring]"""
import re
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    kernel = None
    prev = None          # previous real instruction (text), reset at labels: a label is a join point -- the predecessor is unknown, report it separately
    prev_is_label = False
    stats = {}
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            kernel = m.group(1); prev = None; prev_is_label = False
            continue
        if kernel is None or not s or s.startswith(';') or s.startswith('.') and not s.endswith(':'):
            continue
        if re.match(r'^\.?\w+:', s):          # basic-block label
            prev_is_label = True
            continue
        ins = s.split(';')[0].strip()
        if not ins:
            continue
        if re.match(r'buffer_load_\w+ .*\blds\b', ins):
            st = stats.setdefault(kernel, dict(dma=0, hazard=[], after_label=0))
            st['dma'] += 1
            if prev is not None and re.match(r's_\w+ m0\b', prev) and not prev_is_label:
                st['hazard'].append((prev, ins))
            if prev_is_label:
                st['after_label'] += 1
        prev = ins
        prev_is_label = False
    bad = 0
    for k, st in stats.items():
        if filt and filt not in k:
            continue
        print('%-90s LDS-DMA sites %3d   M0 written by the instruction right before: %d' % (k[:90], st['dma'], len(st['hazard'])))
        for p, i in st['hazard']:
            print('      %s   ->   %s' % (p, i))
        bad += len(st['hazard'])
    print('TOTAL hazard sites: %d' % bad)


if __name__ == '__main__':
    main()
