#!/usr/bin/env python3
"""Order of the LDS-relevant instructions in the main loop of a piece kernel, from hipcc's assembly (every ds_read* that consumes
LDS-DMA-written bytes with the s_waitcnt vmcnt(N) + s_barrier that orders it).
usage: lds_hazard_table.py file.s <mangled-kernel-substring>
Prints the prologue's DMA issues, then the loop body from its header to its back edge: wait counters, barriers, LDS reads, LDS-DMA issues, matrix
instructions (counted), with assembly line numbers."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    hdr = [i for i, l in enumerate(body) if 'Inner Loop Header' in l]
    print('kernel %s: %d lines, inner loop headers at %s' % (lines[start].split(':')[0], len(body), [h + 1 for h in hdr]))
    pat = re.compile(r'\b(s_waitcnt|s_barrier|ds_read\w*|ds_write\w*|buffer_load_\w+|v_mfma\w*|s_cbranch\w*|s_branch|s_endpgm|global_store\w*|buffer_store\w*|s_mov_b32 m0|s_add_i32 m0|s_or_b32 m0)\b')
    def show(lo, hi, title):
        print('---- ' + title)
        mf = 0
        for i in range(lo, hi):
            s = body[i].strip()
            if s.startswith(';') and 'ASM' not in s:
                continue
            if re.match(r'^\.LBB\w+:', s):
                if mf: print('            ... %d matrix instructions' % mf); mf = 0
                print('%6d  %s' % (i + 1, s.split(';')[0].strip()))
                continue
            m = pat.search(s)
            if not m:
                continue
            if m.group(1).startswith('v_mfma'):
                mf += 1
                continue
            if m.group(1).startswith('buffer_load') and ' lds' not in s:
                continue
            if mf: print('            ... %d matrix instructions' % mf); mf = 0
            print('%6d      %s' % (i + 1, s.split(';')[0].strip()[:100]))
        if mf: print('            ... %d matrix instructions' % mf)
    first = hdr[0]
    dma0 = next(i for i, l in enumerate(body) if re.search(r'buffer_load_\w+ .* lds', l))
    show(dma0 - 3, first, 'prologue: first LDS-DMA to the loop header')
    # the loop body: from the header to the last branch back to it
    lab = re.match(r'^(\.LBB\w+):', body[first].strip()).group(1)
    back = max(i for i, l in enumerate(body) if re.search(r's_c?branch\w* ' + re.escape(lab) + r'\b', l) or re.search(r's_branch ' + re.escape(lab) + r'\b', l)) if any(lab in l for l in body[first + 1:]) else first + 200
    # blocks between the header and the epilogue that jump back (the loop's tail may be laid out BEFORE the header)
    tail_lo = max(0, first - 80)
    show(first, min(len(body), max(back + 1, first + 140)), 'loop body from its header')
    show(tail_lo, first, 'block laid out in front of the header (the loop\'s tail: back edge)')


if __name__ == '__main__':
    main()
