#!/usr/bin/env python3
"""Turns the per-pass summaries of tools/collect_pmc.sh into the evidence files bench.py reads back:
profiles/<tag>_pmc.txt (all summaries + the reading) and <tag>_pmc_conv_headline.json / _pmc_dominant.json / _pmc_upfirdn.json.
gfx950 corrections per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE reports half the bytes of wide coalesced
reads (doubled here); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
Round 4: kernels are matched by FAMILY (template name without arguments, as bench.py groups them), every JSON records the sha256 of the
kernel sources the passes were taken on (kernel_source_sha16 -- bench.py refuses a figure from another version), and the weight-gradient
family gets its own file (<tag>_pmc_wgrad.json: SQ counters + traffic on the three largest layers).
usage: python tools/pmc_to_json.py <tag> [dominant kernel family as bench.py prints it]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r'(.+?)\s{2,}(\S+)\s+n=(\d+)\s+avg=(\S+)', line.rstrip())
        if m:
            out[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))
    return out


def norm(name):       # rocprof prints default template arguments too: "<false, true, 0>" -> "<false, true>"
    return re.sub(r', 0>$', '>', name)


def family(name):
    return name.split('<')[0].split(' (')[0].strip()


def main():
    tag = sys.argv[1]
    dominant = family(sys.argv[2]) if len(sys.argv) > 2 else 'conv_fwd_planes_w4_kernel'
    d = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag, 'pmc')
    P = lambda n: parse(os.path.join(d, 'pmc_%s.txt' % n))
    sha_path = os.path.join(d, 'kernel_source_sha16.json')
    sha = json.load(open(sha_path)) if os.path.exists(sha_path) else {}
    sq1, sq2, sq3 = P('sq1'), P('sq2'), P('sq3')
    # the headline call is several kernels in the piece form (images + tile kernel): the figures below are the TILE kernel's
    tile = [k for (k, c) in sq1 if k.startswith('conv_fwd')]
    kname = norm(tile[0])
    one = lambda tab, ctr: next(v[1] for (k, c), v in tab.items() if c == ctr and norm(k) == kname)
    call_sum = lambda tab, ctr: sum(v[1] for (k, c), v in tab.items() if c == ctr)          # whole call (the launches run 1:1)
    gui = one(sq3, 'GRBM_GUI_ACTIVE') / 8.0
    busy = one(sq1, 'SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024)
    wc = one(sq1, 'SQ_WAVE_CYCLES')
    fx, wx = call_sum(P('fetch_xcd'), 'FETCH_SIZE'), call_sum(P('write_xcd'), 'WRITE_SIZE')
    fp = call_sum(P('fetch_plain'), 'FETCH_SIZE')
    hit = lambda t: one(t, 'TCC_HIT_sum') / (one(t, 'TCC_HIT_sum') + one(t, 'TCC_MISS_sum'))
    l2x, l2p = hit(P('l2_xcd')), hit(P('l2_plain'))
    head = dict(kernel=kname, shape='modulated conv 128x128 3x3 Cin=Cout=128 batch 6', FETCH_SIZE_KiB=fx, WRITE_SIZE_KiB=wx, fetch_correction=2.0,
                traffic_bytes_per_launch=int((2 * fx + wx) * 1024), FETCH_SIZE_KiB_plain_block_order=fp, l2_hit_rate=round(l2x, 3),
                l2_hit_rate_plain_block_order=round(l2p, 3), mfma_busy_frac=round(busy, 3), cycles_per_launch=round(gui),
                traffic_note='FETCH / WRITE summed over the kernels of one call (piece form: x image + filter image + tile kernel)',
                kernel_source_sha16=sha, source='profiles/%s_pmc.txt (rocprofv3 --pmc, separate passes, tools/collect_pmc.sh)' % tag)
    # dominant instantiation over the eager device work of G_train + D_train
    fsum = wsum = n = 0
    for op in ('G_train', 'D_train'):
        f, w = P(op + '_FETCH_SIZE'), P(op + '_WRITE_SIZE')
        for (k, c), (cnt, avg) in f.items():
            if family(norm(k)) == dominant:
                fsum += cnt * avg; n += cnt
        for (k, c), (cnt, avg) in w.items():
            if family(norm(k)) == dominant:
                wsum += cnt * avg
    dom = dict(kernel=dominant, launches_sampled=n, FETCH_SIZE_KiB=round(fsum / max(n, 1), 1), WRITE_SIZE_KiB=round(wsum / max(n, 1), 1), fetch_correction=2.0,
               traffic_bytes_per_launch=int((2 * fsum + wsum) / max(n, 1) * 1024),
               note="average over this family's launches in the eager device work of 3 G steps and 3 D steps (tools/op_profile.py under --pmc); the regulariser steps are not sampled",
               kernel_source_sha16=sha, source='profiles/%s_pmc.txt' % tag)
    first = lambda tab, ctr: next(v[1] for (k, c), v in tab.items() if c == ctr)
    uf, uw = first(P('upfirdn_FETCH_SIZE'), 'FETCH_SIZE'), first(P('upfirdn_WRITE_SIZE'), 'WRITE_SIZE')
    up = dict(kernel='upfirdn2d_fir4_kernel<8, 2>', FETCH_SIZE_KiB=uf, WRITE_SIZE_KiB=uw, fetch_correction=2.0, traffic_bytes_per_launch=int((2 * uf + uw) * 1024),
              note='average over the three 128x128 call sites of tools/kernel_bench.py upfirdn 6 (the sites bench.py times)', kernel_source_sha16=sha, source='profiles/%s_pmc.txt' % tag)
    # weight-gradient family on the three largest layers
    wg_layers = []
    for i, layer in enumerate(('G 32 Conv1 (N24 C512)', 'G 64 Conv1 (N24 C256)', 'G 128 Conv1 (N24 C128)'), 1):
        t = parse(os.path.join(d, 'pmc_wgrad_layer%d.txt' % i))
        tr = parse(os.path.join(d, 'pmc_wgrad_layer%d_traffic.txt' % i))
        ks = sorted({k for (k, c) in t if 'wgrad' in k})
        if not ks:
            continue
        k = ks[0]
        g = lambda ctr: next((v[1] for (kk, c), v in t.items() if kk == k and c == ctr), float('nan'))
        gui_w = g('GRBM_GUI_ACTIVE') / 8.0
        fsz = sum(v[1] for (kk, c), v in tr.items() if c == 'FETCH_SIZE' and 'wgrad' in kk)
        wsz = sum(v[1] for (kk, c), v in tr.items() if c == 'WRITE_SIZE' and 'wgrad' in kk)
        wg_layers.append(dict(layer=layer, kernel=norm(k), cycles_per_launch=round(gui_w), mfma_busy_frac=round(g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui_w * 1024), 3),
                              lds_bank_conflict_cycles=g('SQ_LDS_BANK_CONFLICT'), salu_per_mfma=round(g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA'), 2),
                              valu_per_mfma=round((g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'), 2), wait_inst_lds=g('SQ_WAIT_INST_LDS'),
                              FETCH_SIZE_KiB=fsz, WRITE_SIZE_KiB=wsz, fetch_correction=2.0, traffic_bytes_per_launch=int((2 * fsz + wsz) * 1024)))
    wgj = dict(kernel=wg_layers[0]['kernel'] if wg_layers else None, layers=wg_layers, kernel_source_sha16=sha,
               note='tools/pmc_layer.sh + tools/pmc_traffic.sh on tools/conv_layers.py shapes; counters per launch of the weight-gradient tile kernel, traffic incl. its reduce', source='profiles/%s_pmc.txt' % tag)
    prof = os.path.join(ROOT, 'profiles')
    for name, obj in (('pmc_conv_headline', head), ('pmc_dominant', dom), ('pmc_upfirdn', up), ('pmc_wgrad', wgj)):
        with open(os.path.join(prof, '%s_%s.json' % (tag, name)), 'w') as f:
            json.dump(obj, f, indent=1)
    with open(os.path.join(prof, '%s_pmc.txt' % tag), 'w') as f:
        f.write('# rocprofv3 --pmc passes (separate runs, --kernel-trace only), MI355X; collected by tools/collect_pmc.sh %s, assembled by tools/pmc_to_json.py.\n' % tag)
        f.write('#   headline shape:  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/kernel_bench.py conv 6 5\n')
        f.write('#       (modulated conv 128x128 3x3 Cin=Cout=128 batch 6; IGAN_XCD_REMAP=0 for the "plain block order" passes)\n')
        f.write('#   upfirdn2d:       ... -- python3 tools/kernel_bench.py upfirdn 6 5\n')
        f.write('#   training ops:    ... -- python3 tools/op_profile.py {G_train|D_train} 2      (eager device work of the op, 3 calls)\n')
        for n_ in sorted(os.listdir(d)):
            f.write('## %s\n' % n_)
            f.write(open(os.path.join(d, n_)).read())
        f.write('\n# reading (GRBM_GUI_ACTIVE summed over 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled per the gfx950 correction):\n')
        f.write('#  headline modconv, B=6 (M=98304, N=128, K=1152), %s:\n' % kname)
        f.write('#    cycles/launch = %.4g;  MFMA busy = %.4g / (cycles * 1024 SIMDs) = %.1f %% of cycles\n' % (gui, one(sq1, 'SQ_VALU_MFMA_BUSY_CYCLES'), busy * 100))
        mops = one(sq2, 'SQ_INSTS_VALU_MFMA_MOPS_F32') + sum(next((v[1] for (k, c), v in sq2.items() if c == ctr and norm(k) == kname), 0.0)
                                                             for ctr in ('SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F16'))
        f.write('#    MFMA op count = %.4g MOPS * 512 = %.4g FLOP issued (algorithmic 2*M*N*K = 2.899e10; the piece forms issue 3 fp16 / 6 bf16 products per fp32 product)\n' % (mops, mops * 512))
        f.write('#    wave cycles: issue-stalled %.1f %% (SQ_WAIT_INST_ANY), parked on s_waitcnt / s_barrier %.1f %% (SQ_WAIT_ANY), issuing %.1f %%; LDS bank conflicts %g\n' % (
            one(sq1, 'SQ_WAIT_INST_ANY') / wc * 100, one(sq1, 'SQ_WAIT_ANY') / wc * 100, one(sq1, 'SQ_ACTIVE_INST_ANY') / wc * 100, one(sq1, 'SQ_LDS_BANK_CONFLICT')))
        f.write('#    instructions: %.3g MFMA, %.3g VALU (incl. MFMA), %.3g LDS, %.3g SALU, %.3g VMEM reads, %.3g VMEM writes\n' % (
            one(sq3, 'SQ_INSTS_MFMA'), one(sq2, 'SQ_INSTS_VALU'), one(sq2, 'SQ_INSTS_LDS'), one(sq2, 'SQ_INSTS_SALU'), one(sq2, 'SQ_INSTS_VMEM_RD'), one(sq2, 'SQ_INSTS_VMEM_WR')))
        f.write('#    HBM-side traffic = 2*%.0f + %.0f KiB = %.1f MiB per launch with the XCD-aware block order (L2 hit rate %.1f %%), 2*%.0f + %.0f = %.1f MiB with the plain order (%.1f %%);\n' % (
            fx, wx, (2 * fx + wx) / 1024, l2x * 100, fp, wx, (2 * fp + wx) / 1024, l2p * 100))
        f.write('#      algorithmic: 48 MiB in + 48 MiB out + 0.6 MiB weights -> MFMA-bound\n')
        f.write('#  dominant kernel of the step, %s: fetch 2*%.0f + write %.0f KiB = %.1f MiB per launch (average over %d launches of G_train + D_train)\n' % (
            dominant, dom['FETCH_SIZE_KiB'], dom['WRITE_SIZE_KiB'], dom['traffic_bytes_per_launch'] / 2 ** 20, n))
        f.write('#  upfirdn2d (average over the three 128x128 call sites): fetch 2*%.0f + write %.0f KiB = %.1f MB per launch against 168.0 MB algorithmic (in + out)\n' % (
            uf, uw, up['traffic_bytes_per_launch'] / 1e6))
    print(json.dumps(head)); print(json.dumps(dom)); print(json.dumps(up)); print(json.dumps(wgj))


if __name__ == '__main__':
    main()
