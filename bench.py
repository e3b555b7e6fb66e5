#!/usr/bin/env python3
"""Throughput benchmark of the hot path: training img/s of StyleGAN2+IMLE (config-e-Gskip-Dresnet)
on synthetic CelebA-shaped 128x128 batches, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... --
     or bare: the script then starts its N ranks itself as a child torch.distributed.run)

A "step" is one iteration of the reference's host loop (training/training_loop.py:466-482): G step,
G path-length reg every 4th, D step + Gs EMA, D R1 reg every 16th; it advances the image counter by
2*minibatch_size (:481), which is what "img" means in img/s.  The IMLE refresh (candidate generation
+ nearest-neighbour assignment, :353-406) runs once before the timed region and is reported
separately (`imle_refresh_s`), as BASELINE.md prescribes.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the conv family timed per call INSIDE the replayed training graphs with device-side time stamps, grouped by kernel
                family; `kernel` = the family with the largest total time: algorithmic FLOPs / time vs its peak (piece families: the 2.5 PFLOP/s
                fp16 / bf16 dense peak over the form's products per fp32 product -- 3 for the default two-piece fp16 form = 833.3 fp32-equivalent
                TFLOP/s, 6 for the bf16-piece form = 416.7; fp32-instruction families: 157.3 TFLOP/s), plus
                `wgrad`, `families`; the north-star shape (128x128 Conv1) alone and the HBM-bound upfirdn2d (GB/s vs 8 TB/s) with HIP
                events, sustained; `traffic` only from a --pmc pass taken on the kernels as they are now;
  second_line_exact_fp32 / line_bf16_pieces / line_fp16_pairs: the same steady state with the other convolution forms (child runs), labelled;
  cpu_baseline  the CPU oracle (oracle/, PyTorch-CPU fp32) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
_LAUNCH_AFFINITY = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None      # before this rank confines itself to its GPU's socket
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense" (v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD)
# matrix instructions per fp32 product in a piece form (csrc/conv2d_mfma.hip): 1 = three bf16 pieces: a0b0, a0b1, a1b0, a0b2, a1b1, a2b0;
# 2 = two fp16 pieces under per-pixel / per-channel power-of-two scales (no per-tensor window: DESIGN.md section 4): p0p0, p0p1, p1p0.  fp32-equivalent peak = 2500 / products (fp16 and bf16 MFMA run at the same rate)
PIECE_PRODUCTS_BY_FORM = {1: 6, 2: 3}
PIECE_DTYPE = {0: 'f32',
               1: 'f32 (3x3 convs: exact 3-piece bf16 split, fp32 sums)',
               2: 'f32-equivalent (3x3 convs: 2-piece fp16 split, operands to <= 1 ulp of fp32, scales per pixel / per channel: no per-tensor window; fp32 sums)'}


def piece_form():
    """Which form the large 3x3 convolutions run in (csrc/conv2d_mfma.hip planes_mode; igan_conv_piece_form() is the library's own answer):
    IGAN_CONV_PLANES=0 -> 0 (fp32 matrix instruction everywhere), 1 -> 1 (three bf16 pieces), unset / 2 -> 2 (two fp16 pieces: the default)."""
    v = os.environ.get('IGAN_CONV_PLANES')
    try:
        m = int(v) if v is not None else DEFAULT_PIECE_FORM
    except ValueError:
        m = 0
    return 0 if m == 0 else (1 if m == 1 else 2)


DEFAULT_PIECE_FORM = 2


def piece_form_on():
    return piece_form() != 0


def piece_products():
    return PIECE_PRODUCTS_BY_FORM[piece_form()]


def piece_form_peak():
    return BF16_DENSE_PEAK_TFLOPS / piece_products()


def family_of(kernel_name):
    """Kernel FAMILY = the template name without its arguments (and without the '(+ reduce)' / '(shared image)' notes): all
    instantiations of conv_wgrad_kernel<...> are one family."""
    return kernel_name.split('<')[0].split(' (')[0].strip()


def family_peak(family):
    return piece_form_peak() if 'planes' in family else F32_MATRIX_PEAK_TFLOPS


def file_sha16(path):
    import hashlib
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(suffix, want_kernel=None, sources=('conv2d_mfma.hip',)):
    """traffic_bytes_per_launch of the newest committed `--pmc` pass profiles/rNN_<suffix> -- ONLY if that pass was taken on the kernels
    as they are now: the file records the sha256 of the kernel sources it was collected with (tools/pmc_to_json.py), and a file that
    records none, or another one, is refused (-> (None, reason)).  (a constant read from a stale file.)"""
    path = _latest_profile(suffix)
    if not path:
        return None, 'no profiles/rNN_%s' % suffix
    with open(path) as f:
        pj = json.load(f)
    have = pj.get('kernel_source_sha16', {})
    for src in sources:
        now = file_sha16(os.path.join(ROOT, 'inclusivegan_amd', 'csrc', src))
        if have.get(src) != now:
            return None, '%s was collected on another version of csrc/%s (recorded %s, now %s): re-run tools/collect_pmc.sh' % (os.path.basename(path), src, have.get(src), now)
    if want_kernel is not None and family_of(pj.get('kernel', '')) != family_of(want_kernel):
        return None, '%s is about %s, not %s' % (os.path.basename(path), pj.get('kernel'), want_kernel)
    return pj['traffic_bytes_per_launch'], os.path.basename(path)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=32)
    p.add_argument('--warmup', type=int, default=4)
    p.add_argument('--resolution', type=int, default=128)
    p.add_argument('--minibatch-gpu', type=int, default=6)
    p.add_argument('--data-size', type=int, default=30000, help='images in the data set = reals of one IMLE refresh (BASELINE config 4: 30 000); rounded down to a multiple of 2 * minibatch_gpu * gpus')
    p.add_argument('--num-samples-factor', type=int, default=10)
    p.add_argument('--lpips-weight', type=float, default=2.5)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--no-variant-line', action='store_true', help='skip the labelled measurements with the other convolution forms (child runs of this script: exact fp32 and the other piece form)')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend for --gpus > 1 (nccl = RCCL; gloo for the one-GPU tests)')
    p.add_argument('--one-gpu', action='store_true', help='test hook: every rank on device 0 (needs --backend gloo: RCCL refuses two ranks on one device)')
    p.add_argument('--revalidate-every', type=int, default=0, help='stress check: every N iterations (untimed work inside the loop) replay each captured op against its eager execution again; the results land in hip_graphs.checks')
    p.add_argument('--op-times', action='store_true', help='also report the mean device time of each training op (HIP events)')
    p.add_argument('--conv-shapes', default=None, metavar='FILE', help='write the per-shape table of the conv family inside the replayed graphs (device stamps) to FILE')
    return p.parse_args()


def headline_shape_roofline(device, batch, reps=40, warm_s=0.4):
    """The north-star shape: modulated conv 128x128 Conv1 (GEMM M = batch*128*128, N = 128, K = 1152;
    SURVEY.md section 8a/8d), timed alone with HIP events on the launch stream.  The shape first runs back to back for
    `warm_s` seconds: the device leaves its idle clock only after a few hundred milliseconds of load, and a measurement
    taken during that ramp (as in round 1) reads ~20 % low."""
    import torch
    from inclusivegan_amd import hip_ops
    cin = cout = 128
    res = 128
    x = torch.randn(batch, cin, res, res, device=device).contiguous(memory_format=torch.channels_last)
    w = torch.randn(3, 3, cin, cout, device=device) / 34.0
    s = torch.rand(batch, cin, device=device) + 0.5
    d = torch.rand(batch, cout, device=device) + 0.5
    geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(20):
            hip_ops.conv2d_raw(x, w, geom, (res, res), cout, in_scale=s, out_scale=d)
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()     # same stream the kernels are launched on (torch's current stream)
    for _ in range(reps):
        hip_ops.conv2d_raw(x, w, geom, (res, res), cout, in_scale=s, out_scale=d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * batch * res * res * cout * cin * 9
    achieved = flops / (ms * 1e-3) / 1e12
    peak = piece_form_peak() if piece_form_on() else F32_MATRIX_PEAK_TFLOPS
    hip_ops.launch_log = names = []          # the tile kernel this call runs, as the library names it (igan_conv2d_kernel_name)
    hip_ops.conv2d_raw(x, w, geom, (res, res), cout, in_scale=s, out_scale=d)
    hip_ops.launch_log = None
    tile_kernel = names[0][0]
    out = dict(shape='modulated conv 128x128 3x3 Cin=Cout=128 batch %d (M=%d N=128 K=1152)' % (batch, batch * res * res),
               kernel=(tile_kernel + ' (whole call: x image + filter image + tile kernel)') if piece_form_on() else 'conv_fwd_dma_kernel<false, true>',
               achieved=round(achieved, 2), peak=round(peak, 1), frac=round(achieved / peak, 4),
               flops_per_launch=flops, us_per_launch=round(ms * 1e3, 1), traffic=None)
    # HBM-side bytes per launch of this shape from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per
    # the gfx950 correction + WRITE_SIZE); only valid for the batch they were taken at.
    if batch == 6:
        out['traffic'], out['traffic_source'] = pmc_traffic('pmc_conv_headline.json')
    return out


def sustained_instruction_rate(kind):
    """tools/mfma_rate's line for v_mfma_f32_32x32x16_<kind> (bare register loop on random operands, two waves per SIMD, in-kernel clock) from the newest
    committed profiles/rNN_mfma_rate.txt: the clock the chip settles at under that instruction, and the rate at that clock."""
    import re
    path = _latest_profile('mfma_rate.txt')
    if not path:
        return None
    for line in open(path):
        m = re.match(r'mfma_f32_32x32x16_%s\s+2 waves/SIMD.*after [\d.]+ s\s+([\d.]+) TFLOP/s\s+in-kernel clock (\d+) MHz' % kind, line)
        if m:
            return dict(instruction='v_mfma_f32_32x32x16_%s' % kind, tflops=float(m.group(1)), clock_mhz=int(m.group(2)), source=os.path.basename(path))
    return None


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present (evidence files are named per round)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_' + suffix)))
    return hits[-1] if hits else None


def hbm_kernel_roofline(device, batch, warm_s=0.2, reps=40):
    """The HBM-bound kernel north_star names: upfirdn2d at its three 128x128 call sites (FIR after the up-convolution in G,
    FIR before the two strided convolutions in D), sustained, HIP events on the launch stream.  Algorithmic bytes =
    (numel_in + numel_out) * 4."""
    import numpy as np
    import torch
    from inclusivegan_amd import hip_ops
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64
    sites = [('G Conv0_up post-filter', (batch, 129, 129, 128), k * 4, 1, 1),
             ('D Conv1_down pre-filter', (2 * batch, 128, 128, 128), k, 2, 2),
             ('D Skip pre-filter', (2 * batch, 128, 128, 128), k, 1, 1)]
    tot_b = tot_s = 0.0
    per_site = {}
    for name, shape, kk, p0, p1 in sites:
        x = torch.randn(*shape, device=device)
        fn = lambda: hip_ops.upfirdn2d_raw(x, kk, 1, 1, 1, 1, p0, p1, p0, p1)
        y = fn()
        t0 = time.time()
        while time.time() - t0 < warm_s:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        by = (x.numel() + y.numel()) * 4.0
        per_site[name] = round(by / sec / 1e9, 1)
        tot_b += by; tot_s += sec
    gbs = tot_b / tot_s / 1e9
    out = dict(bound='hbm', kernel='upfirdn2d_fir4_kernel', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s',
               frac=round(gbs / HBM_PEAK_GBS, 4), bytes_per_launch=round(tot_b / len(sites)), us_per_launch=round(tot_s / len(sites) * 1e6, 1),
               sites_GBps=per_site, traffic=None)
    if batch == 6:
        out['traffic'], out['traffic_source'] = pmc_traffic('pmc_upfirdn.json', sources=('upfirdn2d.hip',))
    return out


def step_roofline(stamp, steps, shapes_file=None):
    """Roofline of the conv family over the launches of real training iterations, measured with the device running exactly as in the
    timed region: the training ops' hipGraphs are re-captured with a pair of device-side time stamps around every conv-family call
    (hip_ops.StampLog: one-wave kernels reading the 100 MHz counter in stream order, plus a fold kernel per graph that accumulates the
    durations), then replayed for further iterations.  Calls are grouped by kernel FAMILY (template name without its arguments); the
    family with the largest total time is `kernel`.  achieved = sum of algorithmic FLOPs / sum of call durations (a call of the piece
    form includes the filter-image kernel, the x image when the call wrote its own, and the fix-up / reduce launches; images shared
    between calls are the family 'to_planes_kernel', FLOP-free time inside conv_family_tflops).  Peak: 157.3 TFLOP/s for the families on
    the fp32 matrix instruction; for the bf16-piece families the bf16 dense peak / 6 products = 416.7 fp32-equivalent TFLOP/s."""
    totals = stamp.totals_us()
    replays = {}
    for step, (first, count) in steps.items():
        for i in range(first, first + count):
            replays[i] = step.replays
    fam = {}
    for i, (name, flops, us) in enumerate(totals):
        n = replays.get(i, 0)
        if n == 0 or us <= 0:
            continue
        a = fam.setdefault(family_of(name), [0, 0.0, 0.0])
        a[0] += n
        a[1] += flops * n
        a[2] += us * 1e-6
    fam_secs = sum(v[2] for v in fam.values())
    fam_flops = sum(v[1] for v in fam.values())

    def entry(name):
        calls, flops, secs = fam[name]
        ach = flops / secs / 1e12
        pk = family_peak(name)
        return dict(kernel=name, achieved=round(ach, 2), peak=round(pk, 1), unit='TFLOP/s', frac=round(ach / pk, 4), launches=calls,
                    avg_launch_us=round(secs / calls * 1e6, 1), flops_per_launch=round(flops / calls), share_of_conv_time=round(secs / fam_secs, 3))
    mfma = {k: v for k, v in fam.items() if v[1] > 0 and k.startswith('conv_')}
    name = max(mfma, key=lambda k: mfma[k][2])
    if shapes_file:
        with open(shapes_file, 'w') as f:
            f.write('# conv family inside the replayed training graphs (device stamps), %d stamped iterations; sorted by time\n' % max(replays.values()))
            f.write('# %-78s %-58s %8s %10s %8s\n' % ('shape', 'kernel', 'launches', 'total us', 'TFLOP/s'))
            for shape, kname, n, us, tf in stamp.shape_table(replays):
                f.write('%-80s %-58s %8d %10.1f %8.1f\n' % (shape, kname[:58], n, us, tf))
            f.write('# total %.1f ms, %.1f TFLOP/s\n' % (fam_secs * 1e3, fam_flops / max(fam_secs, 1e-12) / 1e12))
            f.write('# per family: ' + '; '.join('%s %.1f ms %.1f TFLOP/s' % (k, v[2] * 1e3, v[1] / v[2] / 1e12) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])) + '\n')
    out = dict(bound='mfma', **entry(name))
    out['peak_note'] = ('%s dense peak %.0f / %d piece products (fp32-equivalent)' % ('fp16' if piece_form() == 2 else 'bf16', BF16_DENSE_PEAK_TFLOPS, piece_products())) if 'planes' in name else 'f32 matrix peak (v_mfma_f32_32x32x2_f32)'
    if 'planes' in name:        # what the piece products' matrix instruction itself sustains on this chip (committed measurement, not a live one): context for `frac`, never the `peak`
        sus = sustained_instruction_rate('f16' if piece_form() == 2 else 'bf16')
        if sus is not None:
            out['matrix_instruction_sustained'] = dict(sus, fp32_equivalent_tflops=round(sus['tflops'] / piece_products(), 1),
                                                       frac_of_sustained=round(out['achieved'] / (sus['tflops'] / piece_products()), 4))
    out['conv_family_tflops'] = round(fam_flops / max(fam_secs, 1e-12) / 1e12, 2)
    out['conv_family_ms_per_iteration'] = round(fam_secs * 1e3 / max(replays.values()), 3)
    out['timing'] = 'device stamps inside the replayed hipGraphs'
    out['traffic'] = None
    out['families'] = {k: entry(k) if v[1] > 0 else dict(kernel=k, launches=v[0], avg_launch_us=round(v[2] / v[0] * 1e6, 1), share_of_conv_time=round(v[2] / fam_secs, 3))
                       for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}
    wg = [k for k in mfma if 'wgrad' in k]
    if wg:
        out['wgrad'] = entry(max(wg, key=lambda k: mfma[k][2]))
    return out


def cpu_baseline(resolution, batch, lpips_weight):
    """CPU oracle on a bounded sample of the SAME workload mix as the GPU line: one G step, one D step, one G
    path-length step and one D R1 step (forward + backward each) are timed once at the same per-GPU batch, and an iteration costs
    t_G + t_D + t_Greg / 4 + t_Dreg / 16 (lazy regularisation: training_loop.py:474-479)."""
    import numpy as np
    import torch
    from oracle import loss as OL
    from oracle.misc import SeededRandom
    from inclusivegan_amd.dnnlib import tflib
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    kw = dict(num_channels=3, resolution=resolution, label_size=0, fmap_base=8192, device='cpu')
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    Lp = tflib.Network('lpips', func_name='inclusivegan_amd.metrics.lpips.vgg16_zhang_perceptual', resolution=resolution, device='cpu', seed=3)
    gp = {n: v.detach().clone().requires_grad_(v.requires_grad) for n, v in G.vars.items()}
    dp = {n: v.detach().clone().requires_grad_(v.requires_grad) for n, v in D.vars.items()}
    lp = {n: v.detach() for n, v in Lp.vars.items()}
    cfg = dict(resolution=resolution, num_channels=3, fmap_base=8192, G_arch='skip', D_arch='resnet', fused_modconv=False)
    rand = SeededRandom(0)
    g = torch.Generator().manual_seed(1)
    reals = lambda n: torch.rand(n, 3, resolution, resolution, generator=g) * 2 - 1
    lat = lambda n: torch.nn.functional.normalize(torch.randn(n, 512, generator=g), dim=1)
    gtr = [p for p in gp.values() if p.requires_grad]
    dtr = [p for p in dp.values() if p.requires_grad]

    def G_step():
        loss, _, _ = OL.G_loss(gp, {k: v.detach() for k, v in dp.items()}, lp, cfg, rand, batch, reals(batch), lat(batch), reals(batch), lat(batch), lpips_weight, phase='loss', state={})
        torch.autograd.grad(loss.mean(), gtr, allow_unused=True)

    def D_step():
        loss, _, _ = OL.D_loss({k: v.detach() for k, v in gp.items()}, dp, cfg, rand, batch, reals(2 * batch), gamma=100, phase='loss', state={})
        torch.autograd.grad(loss.mean(), dtr, allow_unused=True)

    def G_reg():
        _, reg, _ = OL.G_loss(gp, {k: v.detach() for k, v in dp.items()}, lp, cfg, rand, batch, None, lat(batch), None, lat(batch), lpips_weight, phase='reg', state={})
        torch.autograd.grad((reg * 4).mean(), gtr, allow_unused=True)

    def D_reg():
        _, reg, _ = OL.D_loss({k: v.detach() for k, v in gp.items()}, dp, cfg, rand, batch, reals(2 * batch), gamma=100, phase='reg', state={})
        torch.autograd.grad((reg * 16).mean(), dtr, allow_unused=True)

    t = {}
    t_all = time.time()
    for name, fn in (('G', G_step), ('D', D_step), ('G_reg', G_reg), ('D_reg', D_reg)):
        t0 = time.time()
        fn()
        t[name] = time.time() - t0
    per_iter = t['G'] + t['D'] + t['G_reg'] / 4 + t['D_reg'] / 16
    return dict(value=round(2 * batch / per_iter, 4), unit='img/s', cores=cores, kind='port',
                sample='one G step %.1f s + one D step %.1f s + one path-length step %.1f s / 4 + one R1 step %.1f s / 16 (forward+backward each, the GPU line\'s lazy-regularisation mix) '
                       'at minibatch_gpu=%d, %dx%d, PyTorch-CPU fp32 oracle, %.1f s of CPU work' % (t['G'], t['D'], t['G_reg'], t['D_reg'], batch, resolution, resolution, time.time() - t_all),
                op_seconds={k: round(v, 2) for k, v in t.items()})


def knn_cpu_baseline(num_points=30000, dim=3072, num_queries=256):
    """The nearest-neighbour stage on the host: the REFERENCE's own Prioritized-DCI library (oracle/_ref, built from
    /root/reference/dci_code/src by oracle/Makefile) with the training-time parameters (training_loop.py:197,368,398:
    DCI(dim, 3, 15), add(num_levels 3, field_of_view 10, prop_to_retrieve 0.002), query(k 1, field_of_view 200,
    prop_to_retrieve 1.0)) on a bounded sample -- Stacked-MNIST-sized rows (dim 3072), 30 000 candidates -- with all host
    cores (OpenMP).  Returns queries/s for index construction + queries; the GPU leg runs the same problem."""
    import numpy as np
    from oracle import dci_ref
    if not dci_ref.available():
        return dict(value=None, unit='queries/s', cores=0, kind='reference', sample='oracle/_ref/libdci_ref.so not built')
    cores = os.cpu_count() or 1
    rng = np.random.RandomState(0)
    data = rng.uniform(-1, 1, size=(num_points, dim))
    queries = rng.uniform(-1, 1, size=(num_queries, dim))
    t0 = time.time()
    d = dci_ref.DCIRef(dim, 3, 15)
    d.add(data, num_levels=3, field_of_view=10, prop_to_retrieve=0.002)
    t_add = time.time() - t0
    t0 = time.time()
    d.query(queries, num_neighbours=1, field_of_view=200, prop_to_retrieve=1.0)
    t_q = time.time() - t0
    d.close()
    return dict(value=round(num_queries / t_q, 2), unit='queries/s', cores=cores, kind='reference',
                sample='reference DCI library (dci_code/src/dci.c, OpenMP, %d threads): %d candidates x %d dims, add %.1f s, %d queries in %.1f s'
                       % (cores, num_points, dim, t_add, num_queries, t_q), add_s=round(t_add, 2), num_points=num_points, dim=dim)


def knn_gpu(device, num_points=30000, dim=3072, num_queries=3000):
    """The same problem on the HIP path (exact 1-NN): queries/s, candidates resident."""
    import torch
    from inclusivegan_amd.dci_code.dci import DCI
    g = torch.Generator(device='cpu').manual_seed(0)
    data = (torch.rand(num_points, dim, generator=g) * 2 - 1).to(device)
    q = (torch.rand(num_queries, dim, generator=g) * 2 - 1).to(device)
    db = DCI(dim, 3, 15, device=device)
    db.add(data)
    db.query_device(q[:256])
    torch.cuda.synchronize()
    t0 = time.time()
    db.query_device(q)
    torch.cuda.synchronize()
    return round(num_queries / (time.time() - t0), 1)


def cpu_baseline_subprocess(resolution, batch, lpips_weight, timeout_s=600):
    """Runs cpu_baseline() in a child process that never touches HIP (devices hidden), with a time
    limit, so that a slow host cannot hold the benchmark hostage."""
    import subprocess
    code = ('import json,sys; sys.path.insert(0, %r); import bench; '
            'b = bench.cpu_baseline(%d, %d, %r); b["knn"] = bench.knn_cpu_baseline(); '
            'print("CPU_BASELINE " + json.dumps(b))' % (ROOT, resolution, batch, lpips_weight))
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        # the CPU baseline uses the host as the launcher gave it to us, not the one socket this rank pinned itself to
        unpin = (lambda: os.sched_setaffinity(0, _LAUNCH_AFFINITY)) if _LAUNCH_AFFINITY else None
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout_s, env=env, preexec_fn=unpin)
        for line in r.stdout.splitlines():
            if line.startswith('CPU_BASELINE '):
                return json.loads(line[len('CPU_BASELINE '):])
        return dict(value=None, unit='img/s', cores=0, kind='port', sample='cpu baseline failed: %s' % r.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='img/s', cores=0, kind='port', sample='cpu baseline exceeded %d s' % timeout_s)


FORM_LABEL = {
    0: 'exact fp32: every convolution on v_mfma_f32_32x32x2_f32 (IGAN_CONV_PLANES=0); everything else as in the headline',
    1: '3x3 convolutions (forward, data gradient, weight gradient) as 3 bf16 pieces x 6 products with fp32 sums (IGAN_CONV_PLANES=1)',
    2: '3x3 convolutions as 2 fp16 pieces x 3 products under per-pixel / per-channel power-of-two scales, fp32 sums (IGAN_CONV_PLANES=2)',
}
FORM_KEY = {0: 'second_line_exact_fp32', 1: 'line_bf16_pieces', 2: 'line_fp16_pairs'}


def second_line(args, form, timeout_s=900):
    """A labelled line next to the headline: the same steady-state measurement with the convolution form `form` -- a child process,
    because the switch is read once per process (small data set: the refresh is not the subject).  The headline runs whichever form is the
    default (or the environment asks for); the EXACT-fp32-instruction run (IGAN_CONV_PLANES=0) is always reported, and so is the other
    piece form."""
    import subprocess
    other = str(form)
    env = dict(os.environ, IGAN_CONV_PLANES=other)
    # at least 200 timed iterations after 40 of warm-up: short-window rates swing with the thermal state the headline run left behind
    steps, warmup = max(args.steps, 200), max(args.warmup, 40)
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', str(warmup), '--data-size', '1152',
           '--minibatch-gpu', str(args.minibatch_gpu), '--resolution', str(args.resolution), '--lpips-weight', str(args.lpips_weight),
           '--no-cpu-baseline', '--no-variant-line']
    label = FORM_LABEL[form]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # a failed second run never takes the headline with it
        return {'label': label, 'error': repr(e)[:200]}
    roof = d.get('roofline', {})
    return {'label': label, 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'warmup': d['warmup'], 'dtype': d['dtype'],
            'data_size': 1152, 'hip_graphs': d.get('hip_graphs'), 'dominant_kernel': roof.get('kernel'), 'dominant_kernel_tflops': roof.get('achieved'),
            'dominant_kernel_frac': roof.get('frac'), 'conv_family_tflops': roof.get('conv_family_tflops')}


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %7.1fs] %s' % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) as a CHILD process -- nothing in this parent has touched the GPU --
    pass its output through and exit with its code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: required for RCCL between processes on these hosts
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    import inclusivegan_amd  # noqa: F401 -- first: sets the HIP runtime flags the captured graphs need, before anything touches the GPU
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # --one-gpu / --backend gloo: for boxes with a single GPU (tests/test_gpu_dist.py) -- the launch path, the rank slicing, the
    # barriers and the max-over-ranks timing are the ones of a real multi-GPU run.
    if args.one_gpu:
        local_rank = 0
    backend = args.backend
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    from inclusivegan_amd import hostaffinity
    host_threads = hostaffinity.limit_host_threads()         # a small CPU pool: graph replays are host submissions, and a 128-thread OpenMP pool starves them
    pinned = hostaffinity.pin_to_device_node(local_rank)      # IGAN_PIN_NUMA=1 only
    log('host: %d intra-op CPU threads, NUMA pinning %s' % (host_threads, ('%d cpus' % len(pinned)) if pinned else 'off'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, '--gpus %d but WORLD_SIZE is %d' % (args.gpus, world)

    from inclusivegan_amd import _abi
    _abi.get_plugin()   # fail loudly if the HIP extension is missing
    from inclusivegan_amd.dnnlib import EasyDict
    from inclusivegan_amd.training import training_loop as TL

    B = args.minibatch_gpu
    # the loop needs data_size % (2 * minibatch_size) == 0 (training_loop.py:338-340 walks the set in whole double minibatches)
    data_size = args.data_size // (2 * B * world) * (2 * B * world)
    state = dict(t_start=None, t_end=None, refresh=[], iters=0, graphs=None)
    profile_iters = 0 if args.no_roofline else 16     # stamped iterations (graphs re-captured with device time stamps) after the timed region

    def barrier_sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def on_iteration(info):
        state['iters'] += 1
        log('iteration %d done' % state['iters'])
        if args.revalidate_every and state['iters'] % args.revalidate_every == 0:
            info['revalidate_graphs']('iteration %d' % state['iters'])
        if state['iters'] == args.warmup and args.warmup > 0:
            info['drain']()         # the loop hands its device work to a submission thread: everything up to here has been issued ...
            barrier_sync()          # ... and has run
            state['t_start'] = time.perf_counter()
        if state['iters'] == args.warmup + args.steps:
            info['drain']()
            barrier_sync()
            state['t_end'] = time.perf_counter()
            if profile_iters == 0:
                return True
            # roofline leg: the same iterations with the graphs re-captured around device-side time stamps
            from inclusivegan_amd import hip_ops
            from inclusivegan_amd.dnnlib.tflib.graphs import GraphedStep
            # every rank re-captures in lockstep (the graphs contain the gradient exchange under RCCL); the stamps of the other
            # ranks are taken and dropped, so all ranks replay the same graphs
            state['stamp'] = hip_ops.StampLog(device)
            state['stamp_steps'] = {}
            hip_ops.stamp_log = state['stamp']
            GraphedStep.after_capture = lambda step: state['stamp_steps'].__setitem__(step, state['stamp'].fold())
            GraphedStep.generation += 1
        if state['iters'] == args.warmup + args.steps + profile_iters:
            info['drain']()
            torch.cuda.synchronize()
            # the stamped graphs are new captures (all four ops have been re-captured within 16 iterations): they get the same
            # replay-equals-eager check as the first ones.  The stamp totals are frozen first -- the check's replays would add
            # samples the replay counters do not know about -- and its eager side runs without stamps.
            from inclusivegan_amd import hip_ops
            frozen = state['stamp'].totals_us()
            state['stamp'].totals_us = lambda: frozen
            hip_ops.stamp_log = None
            info['revalidate_graphs']('after the stamped re-capture')
            torch.cuda.synchronize()
            return True
        return False

    def on_refresh(seconds):
        state['refresh'].append(seconds)
        log('IMLE refresh %.1f s' % seconds)
        if args.warmup == 0 and state['t_start'] is None:     # the refresh opens iteration 1: start timing right after it
            barrier_sync()
            state['t_start'] = time.perf_counter()

    kwargs = dict(
        G_args=EasyDict(func_name='training.networks_stylegan2.G_main', init_mul=1.0, fmap_base=8 << 10, architecture='skip'),
        D_args=EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature', fmap_base=8 << 10, architecture='resnet'),
        G_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8), D_opt_args=EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8),
        G_loss_args=EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=args.lpips_weight),
        D_loss_args=EasyDict(func_name='training.loss.D_logistic_r1', gamma=100),
        dataset_args=EasyDict(resolution=args.resolution, num_channels=3, label_size=40, label_kind='attributes'),
        sched_args=EasyDict(G_lrate_base=0.002, D_lrate_base=0.002, minibatch_gpu_base=B, minibatch_size_base=B * world),
        tf_config={'rnd.np_random_seed': 1000},
        total_kimg=10 ** 6, data_size=data_size, num_epochs=10000,
        init_staleness=10, num_samples_factor=args.num_samples_factor, knn_perturb_factor=0.05, candidate_batch_size=256,
        hooks=dict(on_iteration=on_iteration, on_refresh=on_refresh, on_graphs=lambda g: state.__setitem__('graphs', g), async_ok=True,
                   **({'op_times': state.setdefault('op_times', {}), 'op_host_times': state.setdefault('op_host_times', {})} if args.op_times else {})),
    )
    log('starting training loop')
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the loop's tick lines and layer tables go to stderr: stdout carries the ONE JSON line
        TL.training_loop(**kwargs)
    log('timed region done')

    elapsed = torch.tensor([state['t_end'] - state['t_start']], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    imgs = 2 * B * world * args.steps
    out = {
        'metric': 'training img/sec (whole node), CelebA 128x128 StyleGAN2+IMLE',
        'value': round(imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': PIECE_DTYPE[piece_form()],
        'data': 'synthetic',
        'config': {'workload': 'CelebA-shaped %dx%d StyleGAN2+IMLE, config-e-Gskip-Dresnet (fmap_base 8192), minibatch_gpu %d, '
                               'NN_rec_lpips_weight %g, lazy reg G/4 D/16, random-init weights' % (args.resolution, args.resolution, B, args.lpips_weight),
                   'global_batch': B * world, 'images_per_step': 2 * B * world, 'parallelism': 'dp%d' % world,
                   'data_size': data_size, 'num_samples_factor': args.num_samples_factor, 'init_staleness': 10},
        'imle_refresh_s': round(state['refresh'][0], 3) if state['refresh'] else None,
        # the four training ops run as replayed hipGraphs, each validated bit for bit against its eager execution after capture
        'hip_graphs': state['graphs'] if state['graphs'] is not None else {'captured': False},
        'host': {'async_submit': os.environ.get('IGAN_ASYNC_SUBMIT', '0') == '1' and not args.op_times, 'cpu_threads': host_threads, 'pinned_to_gpu_numa_node': None if pinned is None else '%d cpus' % len(pinned)},
    }
    if world > 1:
        from inclusivegan_amd.dnnlib.tflib import optimizer as _opt
        nccl = torch.cuda.nccl.version() if backend == 'nccl' and hasattr(torch.cuda, 'nccl') else None
        out['rccl'] = {'ranks': torch.distributed.get_world_size(), 'backend': torch.distributed.get_backend(),
                       'in_graph': bool(state['graphs'] and state['graphs'].get('captured') and _opt.collectives_capturable()),
                       'version': '.'.join(str(v) for v in nccl) if isinstance(nccl, (tuple, list)) else nccl}
    if state['refresh']:
        # `value` is the steady state between refreshes (the refresh is reported separately); one refresh serves
        # data_size * init_staleness images (training_loop.py:354), so over the first period the throughput is
        period = data_size * 10
        out['amortised_img_s'] = round(period / (period / out['value'] + state['refresh'][0]), 3)
    if piece_form() == 2:
        # how often the two-piece fp16 form met an element outside the window in which its two pieces hold the operand to 2^-23: more than 2^26 below the largest of its
        # OWN scale group (round 5: a pixel's channel vector in the row images of the forward / data-gradient kernel; a channel's pixels -- the summed axis -- in the
        # column images of the weight gradient; never a whole tensor)
        import ctypes
        v = (ctypes.c_ulonglong * 4)()
        _abi.check(_abi.get_plugin().igan_debug_f16_window_by_kind(v, 0))
        frac = lambda b, n: (b / n) if n else 0.0
        out['fp16_pairs_window'] = {'elements_imaged': v[1] + v[3], 'nonzero_elements_below_exact_window': v[0] + v[2], 'fraction': frac(v[0] + v[2], v[1] + v[3]),
                                    'row_images': {'imaged': v[1], 'below': v[0], 'fraction': frac(v[0], v[1]), 'scale_group': 'pixel (channel vector)'},
                                    'column_images': {'imaged': v[3], 'below': v[2], 'fraction': frac(v[2], v[3]), 'scale_group': 'channel (its pixels: the summed axis)'}}
    if args.op_times:
        torch.cuda.synchronize()
        out['op_ms'] = {k: round(sum(a.elapsed_time(b) for a, b in v[2:]) / max(len(v) - 2, 1), 3) for k, v in state['op_times'].items()}
        out['op_calls'] = {k: len(v) for k, v in state['op_times'].items()}
        out['op_host_ms'] = {k: round(sum(v[2:]) / max(len(v) - 2, 1) * 1e3, 3) for k, v in state['op_host_times'].items()}     # host time inside the op's call
    if rank == 0:
        if not args.no_roofline:
            from inclusivegan_amd import hip_ops
            from inclusivegan_amd.dnnlib.tflib.graphs import GraphedStep
            hip_ops.stamp_log = None
            GraphedStep.after_capture = None
            log('aggregating %d stamped conv launches' % len(state['stamp'].entries))
            out['roofline'] = step_roofline(state['stamp'], state['stamp_steps'], args.conv_shapes)
            if B == 6:
                out['roofline']['traffic'], out['roofline']['traffic_source'] = pmc_traffic('pmc_dominant.json', want_kernel=out['roofline']['kernel'])
            out['roofline']['headline_shape'] = headline_shape_roofline(device, B)
            out['roofline']['hbm_kernel'] = hbm_kernel_roofline(device, B)
        if world == 1 and not args.no_cpu_baseline:
            log('timing the CPU oracle baseline')
            out["cpu_baseline"] = cpu_baseline_subprocess(args.resolution, B, args.lpips_weight)        # the GPU line's own minibatch_gpu (round 5; 2 until then)
            knn = out['cpu_baseline'].get('knn')
            if isinstance(knn, dict) and knn.get('value'):
                knn['gpu_queries_per_s'] = knn_gpu(device, knn['num_points'], knn['dim'])
        if world == 1 and not args.no_variant_line:
            # the headline form over the labelled lines' longer window (>= 200 iterations after 40 of warm-up): the contract's 20-step window holds one R1 step where
            # 1.25 are expected (lazy regularisation every 16th iteration) and reads ~0.5 % high; this figure does not
            log('sustained window of the headline form (child run)')
            sus = second_line(args, piece_form())
            out['sustained'] = {k: sus.get(k) for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'data_size', 'hip_graphs', 'error') if k in sus}
            out['sustained_img_s'] = sus.get('value')
            for form in (0, 1, 2):
                if form != piece_form():
                    log('labelled line: convolution form %d (child run)' % form)
                    out[FORM_KEY[form]] = second_line(args, form)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
