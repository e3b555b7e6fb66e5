#!/usr/bin/env python3
"""Build-time check of conv_fwd_planes_w4_kernel's hand-issued fragment reads.

The four-wave tile of the fp16 form issues its LDS fragment reads as `asm volatile("ds_read_b128 %0, ...")` WITHOUT a wait: the compiler does not count them,
and the step's single `s_waitcnt ... lgkmcnt(0)` (also inline assembly, naming the registers as its outputs) is what makes their data valid.  That is only
correct while the compiler never copies, spills or otherwise touches a destination register between the read and the wait -- nothing in the language says
so; the generated code does (239 VGPRs, no spill).  This script compiles csrc/conv2d_mfma.hip to gfx950 assembly and fails (exit 1) if, in that kernel,

  * the kernel spills vector registers or uses scratch (.vgpr_spill_count, .private_segment_fixed_size != 0, any scratch_ instruction; scalar spills go to
    lanes of a vector register by v_writelane / v_readlane, which the scan below sees like any other instruction), or
  * any instruction between a `ds_read_b128 v[a:b], ...` and the next `s_waitcnt` with lgkmcnt(0) reads or writes one of v[a:b]
    (linear scan in program order; a basic-block label does not clear the pending set: conservative).

usage: python tools/ds_read_check.py [--asm file.s]    (tests/test_build_checks.py runs it)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = 'conv_fwd_planes_w4_kernel'


def assembly(path=None):
    if path:
        return open(path).read()
    src = os.path.join(ROOT, 'inclusivegan_amd', 'csrc', 'conv2d_mfma.hip')
    out = os.path.join(tempfile.mkdtemp(prefix='dsread_'), 'conv2d_mfma.s')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-Wno-inline-asm', '-Wno-unused-function', '-Wno-unused-variable',
                    src, '-o', out], check=True)
    return open(out).read()


def regs_of(text):
    """VGPR indices named in an operand string: v12, v[98:101]."""
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', text):
        out.add(int(a))
    return out


def check(asm):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        if re.match(r'^_Z\w*%s\w*:' % KERNEL, ln):
            start = i
            break
    assert start is not None, 'kernel %s not found in the assembly' % KERNEL
    problems = []
    pending = {}        # register -> line number of the ds_read that targets it
    reads = waits = 0
    for i in range(start + 1, len(lines)):
        s = lines[i].strip()
        if s.startswith('.Lfunc_end') or s.startswith('.size') or re.match(r'^_Z\w+:', lines[i]):
            break
        ins = s.split(';')[0].strip()
        if not ins or ins.startswith('.') or ins.endswith(':'):
            continue
        if re.match(r'scratch_|buffer_(load|store)\w* .*\boffen\b.*\bs\[?0', ins) or ins.startswith('scratch_'):
            problems.append('line %d: scratch access (%s)' % (i + 1, ins))
        m = re.match(r'ds_read_b128\s+(v\[\d+:\d+\])\s*,\s*(.*)', ins)
        if m:
            reads += 1
            touched = regs_of(m.group(2)) & set(pending)
            if touched:
                problems.append('line %d: address of a ds_read_b128 uses unlanded registers %s' % (i + 1, sorted(touched)))
            for r in regs_of(m.group(1)):
                if r in pending:
                    problems.append('line %d: ds_read_b128 into v%d which a read of line %d has not landed in yet' % (i + 1, r, pending[r]))
                pending[r] = i + 1
            continue
        if ins.startswith('s_waitcnt') and re.search(r'lgkmcnt\(0\)', ins):
            waits += 1
            pending.clear()
            continue
        if pending and not ins.startswith('s_'):
            touched = regs_of(ins) & set(pending)
            if touched:
                problems.append('line %d: `%s` touches v%s before the wait that lands the read of line %d' % (i + 1, ins, sorted(touched), pending[min(touched)]))
    meta = {}
    in_kernel = False
    for ln in lines:           # the kernel's entry in the amdhsa.kernels metadata
        t = ln.strip()
        if t.startswith('.name:') or t.startswith('- .name:'):
            in_kernel = KERNEL in t
        if in_kernel:
            m = re.match(r'-?\s*\.(vgpr_spill_count|sgpr_spill_count|vgpr_count|private_segment_fixed_size|agpr_count):\s*(\d+)', t)
            if m:
                meta[m.group(1)] = int(m.group(2))
    for k in ('vgpr_spill_count', 'private_segment_fixed_size'):
        if meta.get(k, 0) != 0:
            problems.append('%s = %d' % (k, meta[k]))
    return dict(reads=reads, waits=waits, meta=meta, problems=problems)


def main():
    path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == '--asm' else None
    r = check(assembly(path))
    print('%s: %d ds_read_b128 sites, %d lgkmcnt(0) waits, metadata %s' % (KERNEL, r['reads'], r['waits'], r['meta']))
    for p in r['problems']:
        print('PROBLEM ' + p)
    print('OK' if not r['problems'] else 'FAILED: %d problems' % len(r['problems']))
    sys.exit(1 if r['problems'] else 0)


if __name__ == '__main__':
    main()
