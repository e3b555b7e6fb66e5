"""CPU (build container: hipcc cross-compiles gfx950): checks on the GENERATED code of kernels whose correctness rests on something the language does not
promise.  tools/ds_read_check.py: the four-wave fp16 tile issues its LDS fragment reads from inline assembly without a wait; no instruction may
touch a destination register before the step's lgkmcnt(0) wait lands it, and the kernel must not spill vector registers."""
import os
import shutil

import pytest

from tools import ds_read_check


@pytest.mark.skipif(not (os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('hipcc')), reason='needs hipcc')
def test_w4_tile_fragment_reads_are_untouched_until_their_wait():
    r = ds_read_check.check(ds_read_check.assembly())
    assert r['reads'] >= 16 and r['waits'] >= 2, r
    assert r['meta'].get('vgpr_spill_count', 0) == 0 and r['meta'].get('vgpr_count', 999) <= 256, r['meta']
    assert not r['problems'], '\n'.join(r['problems'])


def test_the_scan_sees_a_touched_register():
    asm = '''
_Z25conv_fwd_planes_w4_kernelN4igan8ConvArgsE:
\tds_read_b128 v[98:101], v5 offset:16
\tv_mov_b32_e32 v7, v99
\ts_waitcnt vmcnt(4) lgkmcnt(0)
\tds_read_b128 v[102:105], v5
\ts_waitcnt lgkmcnt(0)
\tv_mov_b32_e32 v7, v103
.Lfunc_end0:
'''
    r = ds_read_check.check(asm)
    assert r['reads'] == 2 and len(r['problems']) == 1 and 'v[99]' in r['problems'][0], r
