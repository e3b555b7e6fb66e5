"""CPU (build container: hipcc cross-compiles gfx950): checks on the GENERATED code of kernels whose correctness rests on something the language does not
promise.  tools/ds_read_check.py: the four-wave fp16 tile issues its LDS fragment reads from inline assembly without a wait; no instruction may
touch a destination register before the step's lgkmcnt(0) wait lands it, and the kernel must not spill vector registers."""
import os
import shutil

import pytest

from tools import asm_scan, ds_read_check

needs_hipcc = pytest.mark.skipif(not (os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('hipcc')), reason='needs hipcc')


@pytest.fixture(scope='module')
def conv_asm(tmp_path_factory):
    """gfx950 assembly of csrc/conv2d_mfma.hip (one 40 s compile for the checks below)."""
    path = tmp_path_factory.mktemp('asm') / 'conv2d_mfma.s'
    path.write_text(ds_read_check.assembly())
    return str(path)


@needs_hipcc
def test_w4_tile_fragment_reads_are_untouched_until_their_wait(conv_asm):
    r = ds_read_check.check(open(conv_asm).read())
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


@needs_hipcc
def test_convolution_kernels_have_no_flat_access_and_no_scratch_on_any_configured_path(conv_asm):
    """Round 6 (profiles/r06_coresidency.txt): under co-residency with another process's tile kernel it was a kernel's flat / scratch accesses that returned
    wrong data, and spills inside a loop cost time in any case.  No kernel of conv2d_mfma.hip may have a flat instruction; scratch is tolerated only in the two
    instantiations no BASELINE configuration reaches (the four-wave scaled weight gradient on RAGGED channel counts)."""
    kernels = {k: v for k, v in asm_scan.scan(conv_asm).items() if 'vgpr_count' in v}
    assert len(kernels) > 30
    names = asm_scan.demangle(list(kernels))
    flat = [names[k] for k, v in kernels.items() if v['flat']]
    assert not flat, flat
    scratch = sorted(names[k].replace('(anonymous namespace)::', '').split('(')[0] for k, v in kernels.items()
                     if v.get('private_segment_fixed_size', 0) or v['scratch'] or v.get('vgpr_spill_count', 0))
    assert scratch == ['void conv_wgrad_kernel<128, 128, 2, 2, false, 1>', 'void conv_wgrad_kernel<128, 128, 2, 2, false, 2>'], scratch
