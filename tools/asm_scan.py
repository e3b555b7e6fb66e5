#!/usr/bin/env python3
"""Static audit of the generated gfx950 code of every kernel in csrc/ for the access shapes that were hit under co-residency with another PROCESS's fp16 tile
kernel (round 5, profiles/r05_replay_mismatch.txt: dense_small_kernel's flat / scratch accesses and its strided 16-byte loads): per kernel
    scratch bytes (.private_segment_fixed_size), scratch_* and flat_* instructions, vector / scalar spills,
    16-byte global / buffer loads (candidates for the strided-request shape; whether a wave's request is strided is a property of the addressing, listed for review).
usage: python tools/asm_scan.py [file.hip ...]   (default: every csrc/*.hip)   exit 1 if any kernel has scratch or flat accesses"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assemble(src, outdir):
    out = os.path.join(outdir, os.path.basename(src)[:-4] + '.s')
    subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-Wno-inline-asm',
                    '-Wno-unused-function', '-Wno-unused-variable', src, '-o', out], check=True, stderr=subprocess.DEVNULL)
    return out


def scan(path):
    kernels, cur = {}, None
    meta_name = None
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            cur = kernels.setdefault(m.group(1), dict(scratch=0, flat=0, b128=0, insts=0))
            continue
        s = ln.strip()
        if cur is not None and (s.startswith('.Lfunc_end') or s.startswith('.section')):
            cur = None
        if cur is not None and s and not s.startswith(('.', ';')) and not s.endswith(':'):
            op = s.split()[0]
            cur['insts'] += 1
            cur['scratch'] += op.startswith('scratch_')
            cur['flat'] += op.startswith('flat_')
            cur['b128'] += bool(re.match(r'(global_load_dwordx4|buffer_load_dwordx4|global_load_b128)', op)) and ' lds' not in s
        m = re.match(r'-?\s*\.name:\s*(\S+)', s)
        if m:
            meta_name = m.group(1)
        m = re.match(r'-?\s*\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s*(\d+)', s)
        if m and meta_name in kernels:
            kernels[meta_name][m.group(1)] = int(m.group(2))
    return kernels


def demangle(names):
    try:
        r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'inclusivegan_amd', 'csrc', '*.hip')))
    tmp = tempfile.mkdtemp(prefix='asm_scan_')
    bad = 0
    print('%-16s %-78s %6s %8s %5s %7s %6s %6s' % ('file', 'kernel', 'VGPRs', 'scratchB', 'flat', 'scratch', 'vspill', 'b128'))
    for src in srcs:
        k = scan(assemble(src, tmp))
        names = demangle(list(k))
        for n, v in k.items():
            if 'vgpr_count' not in v:      # device functions without kernel metadata
                continue
            short = re.sub(r'\(anonymous namespace\)::', '', names[n]).split('(')[0]
            flag = v.get('private_segment_fixed_size', 0) or v['flat'] or v['scratch'] or v.get('vgpr_spill_count', 0)
            bad += bool(flag)
            print('%-16s %-78s %6d %8d %5d %7d %6d %6d%s' % (os.path.basename(src), short[:78], v['vgpr_count'], v.get('private_segment_fixed_size', 0), v['flat'], v['scratch'],
                                                             v.get('vgpr_spill_count', 0), v['b128'], '   <-- scratch / flat' if flag else ''))
    print('kernels with scratch or flat accesses: %d' % bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
