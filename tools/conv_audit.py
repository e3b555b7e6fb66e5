#!/usr/bin/env python3
"""Every convolution call INSIDE a second-order training step, audited against fp64 on sampled output elements (VERDICT r05 next #1b).

The path-length step (training/loss.py:55-89) or the R1 step (:107-111) runs once on the HIP path with hip_ops.conv2d_raw / conv2d_wgrad_raw wrapped:
after each call, `--samples` output elements are re-evaluated from the call's OWN fp32 inputs in fp64 (the definition in include/igan_hip.h, vectorised
gathers on the device; tests/test_conv_audit_reference.py holds it to oracle/conv_sample.py), and the call's error is reported as

    rel     rms(err) / rms(ref)                         the figure the per-kernel parity tests bound
    bias    mean(err) / rms(ref)                        SIGNED: a coherent shift does not average out in the sums over pixels that follow
    z       mean(err) / (std(err) / sqrt(m))            how many standard errors the shift is away from zero (|z| <~ 3: no shift seen)
    zplane  the same per sampled (sample, channel) plane: largest |z| and the rms of the planes' z (1 = independent errors)
    emag    rms of err / sum_k |a_k b_k|                against the element's own magnitude sum

The arithmetic form is the process's (IGAN_CONV_PLANES = 0 / 1 / 2 ...): run once per form.
usage: python tools/conv_audit.py [--res 128] [--fmap 8192] [--B 6] [--op G_reg|D_reg|D_loss] [--state file.npz] [--samples 1024] [--planes 8] [--min-k 1152]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def conv_reference(x, w, geom, idx, w_transposed=False, in_scale=None, out_scale=None):
    """fp64 value and magnitude sum of y[n, co, oy, ox] for idx = (n, co, oy, ox) index tensors [m].  x: logical [N, Cin, H, W]."""
    n, co, oy, ox = idx
    N, Cin, H, Wd = x.shape
    m = n.numel()
    acc = torch.zeros(m, dtype=torch.float64, device=x.device)
    mag = torch.zeros(m, dtype=torch.float64, device=x.device)
    for ky in range(geom.kh):
        v = oy * geom.stride + ky - geom.pad_y
        for kx in range(geom.kw):
            u = ox * geom.stride + kx - geom.pad_x
            ok = (v >= 0) & (v % geom.up == 0) & (v // geom.up < H) & (u >= 0) & (u % geom.up == 0) & (u // geom.up < Wd)
            vv = (v // geom.up).clamp(0, H - 1)
            uu = (u // geom.up).clamp(0, Wd - 1)
            xv = x[n, :, vv, uu].double()                       # [m, Cin]
            if in_scale is not None:
                xv = (x[n, :, vv, uu] * in_scale[n]).double()     # the kernels round x * in_scale to fp32 once
            if w_transposed:
                wv = w[geom.kh - 1 - ky, geom.kw - 1 - kx][co, :].double()      # [m, Cin]
            else:
                wv = w[ky, kx][:, co].t().double()
            p = xv * wv
            acc += p.sum(1) * ok
            mag += p.abs().sum(1) * ok
    f = torch.full((m,), float(geom.alpha), dtype=torch.float64, device=x.device)
    if out_scale is not None:
        f = f * out_scale[n, co].double()
    return acc * f, mag * f.abs()


def wgrad_reference(x, dy, geom, idx, in_scale=None, out_scale=None):
    """fp64 value and magnitude sum of dw[ky, kx, ci, co] for idx = (ky, kx, ci, co) index tensors [m]."""
    ky, kx, ci, co = idx
    N, Cin, H, Wd = x.shape
    _, Cout, OH, OW = dy.shape
    m = ky.numel()
    out = torch.zeros(m, dtype=torch.float64, device=x.device)
    mag = torch.zeros(m, dtype=torch.float64, device=x.device)
    oy = torch.arange(OH, device=x.device)
    ox = torch.arange(OW, device=x.device)
    for a in range(geom.kh):
        v = oy * geom.stride + a - geom.pad_y
        oky = (v >= 0) & (v % geom.up == 0) & (v // geom.up < H)
        for b in range(geom.kw):
            sel = ((ky == a) & (kx == b)).nonzero().flatten()
            if sel.numel() == 0:
                continue
            u = ox * geom.stride + b - geom.pad_x
            okx = (u >= 0) & (u % geom.up == 0) & (u // geom.up < Wd)
            if not bool(oky.any()) or not bool(okx.any()):
                continue
            vi, ui = (v[oky] // geom.up), (u[okx] // geom.up)
            xs = x[:, ci[sel]][:, :, vi][:, :, :, ui]                                  # [N, g, oy', ox']
            if in_scale is not None:
                xs = xs * in_scale[:, ci[sel]][:, :, None, None]
            g = dy[:, co[sel]][:, :, oky.nonzero().flatten()][:, :, :, okx.nonzero().flatten()]
            if out_scale is not None:
                g = g * out_scale[:, co[sel]][:, :, None, None]
            p = xs.double() * g.double()
            out[sel] = p.sum(dim=(0, 2, 3)) * float(geom.alpha)
            mag[sel] = p.abs().sum(dim=(0, 2, 3)) * abs(float(geom.alpha))
    return out, mag


def stats(err, ref, mag, planes):
    m = err.numel()
    rms_ref = float(ref.pow(2).mean().sqrt()) + 1e-300
    sd = float(err.std()) + 1e-300
    out = dict(rel=float(err.pow(2).mean().sqrt()) / rms_ref, bias=float(err.mean()) / rms_ref, z=float(err.mean()) / (sd / m ** 0.5),
               emag=float((err / mag.clamp_min(1e-300))[mag > 0].pow(2).mean().sqrt()) if bool((mag > 0).any()) else 0.0)
    if planes > 1:
        e = err.reshape(planes, -1)
        zp = e.mean(1) / (e.std(1) / e.shape[1] ** 0.5).clamp_min(1e-300)
        zp = torch.where(e.std(1) > 0, zp, torch.zeros_like(zp))        # a plane of one pixel (1x1 maps) has no spread
        out['zplane_max'] = float(zp.abs().max())
        out['zplane_rms'] = float(zp.pow(2).mean().sqrt())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--fmap', type=int, default=8192)
    ap.add_argument('--B', type=int, default=6)
    ap.add_argument('--op', default='G_reg')
    ap.add_argument('--samples', type=int, default=1024)
    ap.add_argument('--planes', type=int, default=8)
    ap.add_argument('--state', default=None, help='a state file of tests/reg_forms.py (tools/reg_forms.py --keep-state) instead of the initialisation state')
    ap.add_argument('--min-k', type=int, default=1152, help='audit calls with taps * Cin >= this (the piece forms start at 1152)')
    a = ap.parse_args()
    import inclusivegan_amd  # noqa: F401
    from inclusivegan_amd import hip_ops, _abi
    from tests import reg_forms as RF
    lib = _abi.get_plugin()
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device='cpu').manual_seed(77)
    rows = []
    orig_conv, orig_wgrad = hip_ops.conv2d_raw, hip_ops.conv2d_wgrad_raw
    busy = [False]

    def kname_conv(x, cout, out_hw, geom, w_transposed, in_scale, out_scale):
        n, cin, h, wd = x.shape
        p = _abi.Conv2DParams(x=1 << 20, w=1 << 20, y=1 << 20, in_scale=(1 << 20 if in_scale is not None else None), out_scale=(1 << 20 if out_scale is not None else None),
                              workspace=None, workspace_floats=0, N=n, H=h, W=wd, Cin=cin, OH=out_hw[0], OW=out_hw[1], Cout=cout, KH=geom.kh, KW=geom.kw, stride=geom.stride,
                              up=geom.up, pad_y=geom.pad_y, pad_x=geom.pad_x, w_transposed=1 if w_transposed else 0, splits=1, alpha=1.0, bias=None, act=0, act_alpha=0.0, act_gain=1.0)
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
        return buf.value.decode().split('<')[0]

    def kname_wgrad(x, dy, geom, in_scale, out_scale):
        n, cin, h, wd = x.shape
        _, cout, oh, ow = dy.shape
        p = _abi.Conv2DWgradParams(x=1 << 20, dy=1 << 20, dw=1 << 20, in_scale=(1 << 20 if in_scale is not None else None), out_scale=(1 << 20 if out_scale is not None else None),
                                   workspace=None, workspace_floats=0, N=n, H=h, W=wd, Cin=cin, OH=oh, OW=ow, Cout=cout, KH=geom.kh, KW=geom.kw, stride=geom.stride, up=geom.up,
                                   pad_y=geom.pad_y, pad_x=geom.pad_x, splits=1, alpha=1.0)
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_wgrad_kernel_name(ctypes.byref(p), buf, 128))
        return buf.value.decode().split('<')[0]

    def conv_wrapper(x, w, geom, out_hw, cout, w_transposed=False, in_scale=None, out_scale=None, bias=None, act=None, noise=None, strength=None, x_pieces=None, colmax=None):
        y = orig_conv(x, w, geom, out_hw, cout, w_transposed=w_transposed, in_scale=in_scale, out_scale=out_scale, bias=bias, act=act, noise=noise, strength=strength,
                      x_pieces=x_pieces, colmax=colmax)
        if busy[0] or x.device.type != 'cuda' or x.dim() != 4 or geom.kh * geom.kw * x.shape[1] < a.min_k:
            return y
        busy[0] = True
        with torch.no_grad():
            yc = y if act is None else orig_conv(x, w, geom, out_hw, cout, w_transposed=w_transposed, in_scale=in_scale, out_scale=out_scale)     # the convolution without its fused epilogue
            N = x.shape[0]
            P = a.planes
            per = a.samples // P
            n = torch.randint(0, N, (P,), generator=gen).to(dev).repeat_interleave(per)
            co = torch.randint(0, cout, (P,), generator=gen).to(dev).repeat_interleave(per)
            oy = torch.randint(0, out_hw[0], (P * per,), generator=gen).to(dev)
            ox = torch.randint(0, out_hw[1], (P * per,), generator=gen).to(dev)
            ref, mag = conv_reference(x, w, geom, (n, co, oy, ox), w_transposed, in_scale, out_scale)
            got = yc[n, co, oy, ox].double()
            s = stats(got - ref, ref, mag, P)
            s.update(kind=('dgrad' if w_transposed else 'fwd') + ('+s' if in_scale is not None else '') + ('+d' if out_scale is not None else '') + ('+act' if act is not None else ''),
                     shape='N%d %dx%d C%d -> %dx%d C%d s%d u%d' % (N, x.shape[2], x.shape[3], x.shape[1], out_hw[0], out_hw[1], cout, geom.stride, geom.up),
                     kernel=kname_conv(x, cout, out_hw, geom, w_transposed, in_scale, out_scale))
            rows.append(s)
        busy[0] = False
        return y

    def wgrad_wrapper(x, dy, geom, in_scale=None, out_scale=None, x_pieces=None, dy_pieces=None, x_colmax=None, dy_colmax=None):
        dw = orig_wgrad(x, dy, geom, in_scale=in_scale, out_scale=out_scale, x_pieces=x_pieces, dy_pieces=dy_pieces, x_colmax=x_colmax, dy_colmax=dy_colmax)
        if busy[0] or x.device.type != 'cuda' or geom.kh * geom.kw * x.shape[1] < a.min_k:
            return dw
        busy[0] = True
        with torch.no_grad():
            m = min(a.samples, 512)
            ky = torch.randint(0, geom.kh, (m,), generator=gen).to(dev)
            kx = torch.randint(0, geom.kw, (m,), generator=gen).to(dev)
            ci = torch.randint(0, x.shape[1], (m,), generator=gen).to(dev)
            co = torch.randint(0, dy.shape[1], (m,), generator=gen).to(dev)
            ref, mag = wgrad_reference(x, dy, geom, (ky, kx, ci, co), in_scale, out_scale)
            got = dw[ky, kx, ci, co].double()
            s = stats(got - ref, ref, mag, 1)
            s.update(kind='wgrad' + ('+s' if in_scale is not None else '') + ('+d' if out_scale is not None else ''),
                     shape='N%d %dx%d C%d -> %dx%d C%d s%d u%d' % (x.shape[0], x.shape[2], x.shape[3], x.shape[1], dy.shape[2], dy.shape[3], dy.shape[1], geom.stride, geom.up),
                     kernel=kname_wgrad(x, dy, geom, in_scale, out_scale))
            rows.append(s)
        busy[0] = False
        return dw

    if a.state:
        state = RF.load_state(a.state)
    else:
        state, _ = RF.init_state(dev, a.res, a.fmap, a.B, (0.0,))       # records the draws (unpatched), pl_mean = 0
    other = []
    orig_ban_bwd, orig_mb_bwd = hip_ops.bias_act_noise_bwd_raw, hip_ops.mbstd_bwd_raw

    def ban_bwd_wrapper(dy, y, noise, act_idx, alpha, gain, want_db):
        out = orig_ban_bwd(dy, y, noise, act_idx, alpha, gain, want_db)
        if y.device.type == 'cuda' and act_idx in (1, 2, 3):
            with torch.no_grad():
                d64, y64 = dy.double().reshape(y.shape), y.double()
                slope = {1: torch.ones_like(y64), 2: (y64 > 0).double(), 3: torch.where(y64 > 0, torch.ones_like(y64), torch.full_like(y64, float(alpha)))}[act_idx]
                ref = d64 * slope * float(gain)
                e = float((out[0].double() - ref).norm() / ref.norm().clamp_min(1e-300))
                eb = float((out[1].double() - ref.sum(dim=(0, 2, 3) if y.dim() == 4 else 0)).norm() / ref.sum(dim=(0, 2, 3) if y.dim() == 4 else 0).norm().clamp_min(1e-300)) if out[1] is not None else 0.0
                other.append('ban_bwd   %-22s act %d  dx rel %.2e  db rel %.2e  zeros in y %d' % (tuple(y.shape), act_idx, e, eb, int((y == 0).sum())))
        return out

    def mb_bwd_wrapper(x, dy, g):
        out = orig_mb_bwd(x, dy, g)
        with torch.no_grad():
            x64 = x.double().detach().requires_grad_(True)
            with torch.enable_grad():
                yc = hip_ops.mbstd_composite(x64, g)
            ref, = torch.autograd.grad(yc, [x64], dy.double())
            other.append('mbstd_bwd %-22s G %d  dx rel %.2e' % (tuple(x.shape), g, float((out.double() - ref).norm() / ref.norm())))
        return out

    hip_ops.conv2d_raw, hip_ops.conv2d_wgrad_raw = conv_wrapper, wgrad_wrapper
    hip_ops.bias_act_noise_bwd_raw, hip_ops.mbstd_bwd_raw = ban_bwd_wrapper, mb_bwd_wrapper
    try:
        RF.hip_ops_of_state(state, dev, ops=(a.op,))
    finally:
        hip_ops.conv2d_raw, hip_ops.conv2d_wgrad_raw = orig_conv, orig_wgrad
        hip_ops.bias_act_noise_bwd_raw, hip_ops.mbstd_bwd_raw = orig_ban_bwd, orig_mb_bwd
    form = int(lib.igan_conv_piece_form())
    print('# conv calls of one %s step at %dx%d, fmap %d, minibatch_gpu %d; arithmetic form %d (IGAN_CONV_PLANES); %d samples per call in %d (sample, channel) planes'
          % (a.op, state['cfg']['res'], state['cfg']['res'], state['cfg']['fmap'], state['cfg']['B'], form, a.samples, a.planes))
    print('%-3s %-12s %-44s %-28s %9s %10s %7s %11s %9s' % ('#', 'kind', 'shape', 'kernel', 'rel', 'bias', 'z', 'zplane max/rms', 'emag'))
    for i, s in enumerate(rows):
        print('%-3d %-12s %-44s %-28s %9.2e %10.2e %7.2f %5.1f /%5.2f %9.2e' % (i, s['kind'], s['shape'], s['kernel'], s['rel'], s['bias'], s['z'],
                                                                              s.get('zplane_max', 0.0), s.get('zplane_rms', 0.0), s['emag']))
    for line in other:
        print('# other: ' + line)
    by = {}
    for s in rows:
        by.setdefault(s['kernel'], []).append(s)
    print('# by kernel family: calls, rms of rel, mean bias, rms of z (1 = no shift), largest |z|')
    for k, v in sorted(by.items()):
        print('%-32s %3d  rel %.2e  bias %+.2e  z rms %.2f  max %.2f' % (k, len(v), float(np.sqrt(np.mean([s['rel'] ** 2 for s in v]))), float(np.mean([s['bias'] for s in v])),
                                                                     float(np.sqrt(np.mean([s['z'] ** 2 for s in v]))), max(abs(s['z']) for s in v)))


if __name__ == '__main__':
    main()
