"""GPU: the piece forms of the forward / data-gradient / weight-gradient convolution (csrc/conv2d_mfma.hip conv_fwd_planes_kernel,
conv_wgrad_planes_kernel) against fp64: three bf16 pieces / six products (IGAN_CONV_PLANES=1, the default of the first half of round 4) and two
fp16 pieces under per-pixel / per-channel power-of-two scales / three products (IGAN_CONV_PLANES=2 or unset: the default).  IGAN_CONV_PLANES=0 restores
the fp32 instruction everywhere.  The switch is read once per process, so the checks run in child processes with the
switch stated explicitly.  Tolerances are those of the exact-fp32 path's own full-size tests
(3e-6 relative to the output rms per element, tests/test_gpu_fullsize.py), plus the property that made the variant acceptable at
all: no coherent shift of the outputs (|mean error| below 5e-8 rms; profiles/r03_bf16_split_rounding.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, %r)
from inclusivegan_amd import hip_ops, _abi
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(7)
lib = _abi.get_plugin()
seen = set()
_orig = lib.igan_conv2d
def check(name, N, Cin, H, Cout, k, stride, up, pad, out, transposed=False, scales=False, act=None):
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) / (k * k * Cin) ** 0.5
    s = (torch.rand(N, Cin, generator=g) + 0.5) if scales else None
    d = (torch.rand(N, Cout, generator=g) + 0.5) if scales else None
    geom = hip_ops.ConvGeom(k, k, stride, up, pad, pad)
    xs = x.double() * (s.double()[:, :, None, None] if scales else 1.0)
    wd = w.double().permute(3, 2, 0, 1)             # OIHW
    # include/igan_hip.h's formula stated with torch ops, as tests/test_gpu_ops.py _conv_oracle: zero-stuff, pad, correlate
    xu = xs
    if up > 1:
        xu = torch.zeros(N, Cin, (H - 1) * up + 1, (H - 1) * up + 1, dtype=torch.float64)
        xu[:, :, ::up, ::up] = xs
    need = (out - 1) * stride + k
    xp = F.pad(xu, [pad, max(need - pad - xu.shape[3], 0), pad, max(need - pad - xu.shape[2], 0)])
    want = F.conv2d(xp, wd, stride=stride)[:, :, :out, :out]
    if scales:
        want = want * d.double()[:, :, None, None]
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    got = hip_ops.conv2d_raw(xd, w.to(dev), geom, (out, out), Cout, in_scale=(s.to(dev) if scales else None), out_scale=(d.to(dev) if scales else None))
    assert tuple(got.shape) == tuple(want.shape), (name, got.shape, want.shape)
    err = got.double().cpu() - want
    rms = float(want.pow(2).mean().sqrt())
    rel, mean, mx = float(err.pow(2).mean().sqrt()) / rms, float(err.mean()) / rms, float(err.abs().max()) / rms
    print('%%-28s rel L2 %%.2e  mean %%+.2e  max %%.2e' %% (name, rel, mean, mx))
    assert rel < 3e-7 and abs(mean) < 5e-8 and mx < 3e-6, (name, rel, mean, mx)

def kernel_of(N, Cin, H, Cout, k, stride, up, pad, out):
    p = _abi.Conv2DParams(x=1 << 20, w=1 << 20, y=1 << 20, in_scale=None, out_scale=None, workspace=None, workspace_floats=0, N=N, H=H, W=H, Cin=Cin, OH=out, OW=out,
                          Cout=Cout, KH=k, KW=k, stride=stride, up=up, pad_y=pad, pad_x=pad, w_transposed=0, splits=1, alpha=1.0, bias=None, act=0, act_alpha=0.0, act_gain=1.0)
    buf = ctypes.create_string_buffer(128)
    _abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
    return buf.value.decode()

cases = [
    ('3x3 32x32 C256 modulated', 3, 256, 32, 256, 3, 1, 1, 1, 32, dict(scales=True)),
    ('3x3 16x16 C512 (sliced)', 9, 512, 16, 512, 3, 1, 1, 1, 16, dict()),
    ('3x3 s2 33->16 C256->512', 9, 256, 33, 512, 3, 2, 1, 0, 16, dict()),
    ('3x3 up2 16->33 C512->256', 8, 512, 16, 256, 3, 1, 2, 2, 33, dict(scales=True)),
    ('3x3 24x24 C128->384 ragged', 4, 128, 24, 384, 3, 1, 1, 1, 24, dict()),
    ('3x3 8x8 C512 at 24 samples (1536 rows: the test lowers the row threshold to 1024, the product keeps 2048)', 24, 512, 8, 512, 3, 1, 1, 1, 8, dict(scales=True)),
]
# shapes the variant leaves to the fp32 kernels: 1x1 (Skip, the distance GEMM), shallow reductions, few rows
for args in ((2, 128, 64, 256, 1, 1, 1, 0, 64), (2, 64, 32, 128, 3, 1, 1, 1, 32), (3, 512, 8, 512, 3, 1, 1, 1, 8)):
    assert not kernel_of(*args).startswith('conv_fwd_planes'), args
cases = cases
for name, N, Cin, H, Cout, k, stride, up, pad, out, kw in cases:
    assert kernel_of(N, Cin, H, Cout, k, stride, up, pad, out).startswith('conv_fwd_planes'), (name, kernel_of(N, Cin, H, Cout, k, stride, up, pad, out))
    check(name, N, Cin, H, Cout, k, stride, up, pad, out, **kw)
# gradients through the operator surface: data gradient = the same kernel with the filter transposed
from inclusivegan_amd.hip_ops import conv2d
x = torch.randn(3, 256, 32, 32, generator=g)
w = torch.randn(3, 3, 256, 256, generator=g) / 48.0
dy = torch.randn(3, 256, 32, 32, generator=g)
xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
wdv = w.to(dev).requires_grad_(True)
y = conv2d(xd, wdv, hip_ops.ConvGeom(3, 3, 1, 1, 1, 1), (32, 32))
y.backward(dy.to(dev).contiguous(memory_format=torch.channels_last))
x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
F.conv2d(x64, w64.permute(3, 2, 0, 1), padding=1).backward(dy.double())
for nm, got, want in (('dx', xd.grad, x64.grad), ('dw', wdv.grad, w64.grad)):
    e = got.double().cpu() - want
    rel = float(e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print('%%-28s rel L2 %%.2e' %% (nm, rel))
    assert rel < 1e-6, (nm, rel)
# weight gradient in piece form (conv_wgrad_planes_kernel): plain, modulated (both scales), stride 2, up 2, ragged pixel axis
def wgrad_case(name, N, Cin, H, Cout, k, stride, up, pad, out, scales=False):
    x = torch.randn(N, Cin, H, H, generator=g)
    dy = torch.randn(N, Cout, out, out, generator=g)
    s = (torch.rand(N, Cin, generator=g) + 0.5) if scales else None
    d = (torch.rand(N, Cout, generator=g) + 0.5) if scales else None
    xs = (x.double() * (s.double()[:, :, None, None] if scales else 1.0))
    dys = dy.double() * (d.double()[:, :, None, None] if scales else 1.0)
    w0 = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    xu = xs
    if up > 1:
        xu = torch.zeros(N, Cin, (H - 1) * up + 1, (H - 1) * up + 1, dtype=torch.float64)
        xu[:, :, ::up, ::up] = xs
    need = (out - 1) * stride + k
    xp = F.pad(xu, [pad, max(need - pad - xu.shape[3], 0), pad, max(need - pad - xu.shape[2], 0)])
    y = F.conv2d(xp, w0.permute(3, 2, 0, 1), stride=stride)[:, :, :out, :out]
    (y * dys).sum().backward()
    want = w0.grad
    geom = hip_ops.ConvGeom(k, k, stride, up, pad, pad)
    got = hip_ops.conv2d_wgrad_raw(x.to(dev).contiguous(memory_format=torch.channels_last), dy.to(dev).contiguous(memory_format=torch.channels_last), geom,
                                   in_scale=(s.to(dev) if scales else None), out_scale=(d.to(dev) if scales else None))
    err = got.double().cpu() - want
    rms = float(want.pow(2).mean().sqrt())
    rel, mean, mx = float(err.pow(2).mean().sqrt()) / rms, float(err.mean()) / rms, float(err.abs().max()) / rms
    print('%%-28s rel L2 %%.2e  mean %%+.2e  max %%.2e' %% ('wgrad ' + name, rel, mean, mx))
    assert rel < 5e-7 and abs(mean) < 1e-7 and mx < 5e-6, (name, rel, mean, mx)

wgrad_case('3x3 32x32 C256 modulated', 3, 256, 32, 256, 3, 1, 1, 1, 32, scales=True)
wgrad_case('3x3 24x24 C128->384', 4, 128, 24, 384, 3, 1, 1, 1, 24)
wgrad_case('3x3 s2 33->16 C256->512', 9, 256, 33, 512, 3, 2, 1, 0, 16)
wgrad_case('3x3 up2 16->33 C512->256', 8, 512, 16, 256, 3, 1, 2, 2, 33, scales=True)
wgrad_case('3x3 19x19 C160->224 ragged', 6, 160, 19, 224, 3, 1, 1, 1, 19)
wgrad_case('3x3 8x8 C512 at 24 samples (1536 summed pixels)', 24, 512, 8, 512, 3, 1, 1, 1, 8, scales=True)
print('PLANES-VARIANT-OK')
'''


@pytest.mark.parametrize('form', ['1', '2'], ids=['bf16_x3_six_products', 'fp16_x2_three_products'])
def test_piece_forms_against_fp64(cuda_device, form):
    """IGAN_CONV_PLANES=1: three bf16 pieces, six products.  =2: two fp16 pieces under per-pixel / per-channel power-of-two scales, three products
    (the default form; channel counts that are not powers of two stay on the fp32 kernels there) -- same shapes, same tolerances."""
    env = dict(os.environ, IGAN_CONV_PLANES=form, IGAN_PLANES_MIN_ROWS='1024', IGAN_WGRAD_PLANES_MIN_ROWS='1024')      # the 8x8 cases run in the piece form (the product's threshold since round 6, stated here so that the cases do not move with it)
    r = subprocess.run([sys.executable, '-c', CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0 and 'PLANES-VARIANT-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


SHARE_CHILD = r'''
import hashlib, sys
import torch
sys.path.insert(0, %r)
from inclusivegan_amd import hip_ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(11)
def dig(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.detach().float().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()
out = []
geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
# modulated layer (ModConv2dFn), fused synthesis layer (ModConvBanFn) and plain layer with epilogue (ConvBiasActFn): forward + first-order backward
x = torch.randn(4, 256, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(3, 3, 256, 256, generator=g) / 48.0).to(dev).requires_grad_(True)
s = (torch.rand(4, 256, generator=g) + 0.5).to(dev).requires_grad_(True)
d = (torch.rand(4, 256, generator=g) + 0.5).to(dev).requires_grad_(True)
dy = torch.randn(4, 256, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
y = hip_ops.ModConv2dFn.apply(x, w, s, d, geom, (32, 32))
gx, gw, gs, gd = torch.autograd.grad(y, [x, w, s, d], dy)
out.append(dig(y, gx, gw, gs, gd))
b = torch.randn(256, generator=g).to(dev).requires_grad_(True)
y = hip_ops.ConvBiasActFn.apply(x, w, b, geom, (32, 32), 3, 0.2, 2 ** 0.5)
gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
out.append(dig(y, gx, gw, gb))
noise = torch.randn(1, 1, 32, 32, generator=g).to(dev)
strength = torch.tensor(0.3, device=dev, requires_grad=True)
y = hip_ops.ModConvBanFn.apply(x, w, s, d, b, noise, strength, geom, (32, 32), 3, 0.2, 2 ** 0.5)
gx, gw, gs, gd, gb = torch.autograd.grad(y, [x, w, s, d, b], dy)
out.append(dig(y, gx, gw, gs, gd, gb))
print('DIGESTS ' + ' '.join(out))
'''


def test_four_wave_tile_is_bit_identical_to_the_eight_wave_tile(cuda_device):
    """The fp16 form's forward / data-gradient tile runs on four waves with register-prefetched fragments by default (conv_fwd_planes_w4_kernel); IGAN_F16_W4=0
    is the eight-wave tile (conv_fwd_planes_kernel<2, true>).  Another schedule of the same products and folds: forward outputs and every gradient of a modulated
    layer, a fused synthesis layer and a plain layer with epilogue must agree bit for bit (profiles/r05_w4_tile.txt section 4 has the digests of five larger shapes)."""
    digests = {}
    for w4 in ('0', '1'):
        env = dict(os.environ, IGAN_CONV_PLANES='2', IGAN_F16_W4=w4)
        r = subprocess.run([sys.executable, '-c', SHARE_CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('DIGESTS ')]
        assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-3000:]
        digests[w4] = lines[-1]
    assert digests['0'] == digests['1'], digests


@pytest.mark.parametrize('form', ['1', '2'], ids=['bf16_x3', 'fp16_x2'])
def test_shared_piece_images_change_nothing(cuda_device, form):
    """What the calls of a layer hand each other must change nothing: the bf16 form's piece images written once per layer (hip_ops.to_pieces -> x_pieces / dy_pieces,
    ABI v5) against every convolution call writing its own (IGAN_PIECES_SHARE=0); the fp16 form's channel maxima left by the forward / data-gradient call's row image
    (x_colmax, ABI v8) against the weight gradient finding them by passes of its own (IGAN_COLMAX_SHARE=0) -- the same arithmetic on the same images and the same scales
    (a maximum does not depend on the order), so outputs and all gradients are bit-identical."""
    digs = []
    for share in ('1', '0'):
        env = dict(os.environ, IGAN_CONV_PLANES=form, IGAN_PIECES_SHARE=share, IGAN_COLMAX_SHARE=share)
        r = subprocess.run([sys.executable, '-c', SHARE_CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        digs.append([l for l in r.stdout.splitlines() if l.startswith('DIGESTS ')][-1])
    assert digs[0] == digs[1], digs


FP32_CHILD = r'''
import ctypes, sys
import torch
sys.path.insert(0, %r)
from inclusivegan_amd import hip_ops, _abi
dev = torch.device('cuda', 0)
lib = _abi.get_plugin()
form = {'1': 2, 'bf16': 1, '0': 0}[sys.argv[1]]
assert lib.igan_conv_piece_form() == form         # nothing in the environment = the two-piece fp16 form (ABI v7)
assert lib.igan_conv_pieces_wanted(3, 3, 256, 256) == int(form != 0) and lib.igan_conv_pieces_wanted(1, 1, 256, 256) == 0 and lib.igan_conv_pieces_wanted(3, 3, 64, 256) == 0
# ABI v8: only the bf16 form takes caller-written images (the fp16 form scales a tensor per pixel for one kernel and per channel for the other)
assert lib.igan_pieces_image_ok(4, 1024, 256) == int(form == 1) and lib.igan_pieces_image_ok(1, 1024, 256) == 0 and lib.igan_pieces_image_ok(4, 1024, 144) == 0
p = _abi.Conv2DParams(x=1 << 20, w=1 << 20, y=1 << 20, in_scale=None, out_scale=None, workspace=None, workspace_floats=0, N=4, H=32, W=32, Cin=256, OH=32, OW=32,
                      Cout=256, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1, w_transposed=0, splits=1, alpha=1.0, bias=None, act=0, act_alpha=0.0, act_gain=1.0)
buf = ctypes.create_string_buffer(128)
_abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
print('KERNEL ' + buf.value.decode())
g = torch.Generator().manual_seed(3)
x = torch.randn(4, 256, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(3, 3, 256, 256, generator=g) / 48.0).to(dev)
geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1)
y = hip_ops.conv2d_raw(x, w, geom, (32, 32), 256)
want = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu().permute(3, 2, 0, 1), padding=1)
err = float((y.double().cpu() - want).abs().max() / want.abs().max())
assert err < 3e-6, err
# an image that is not the image of this tensor is refused, never read (ABI v6)
xp = hip_ops.to_pieces(x)
if form == 2:
    assert xp is None
    fake = hip_ops.PieceImage(torch.zeros(x.numel() * 6 // 4, device=dev), x.numel() * 6)
    for call, word in ((lambda: hip_ops.conv2d_raw(x, w, geom, (32, 32), 256, x_pieces=fake), 'x_pieces'),
                       (lambda: hip_ops.conv2d_wgrad_raw(x, x, geom, dy_pieces=fake), 'dy_pieces')):
        try:
            call()
        except Exception as e:
            assert word in str(e), e
        else:
            raise AssertionError('the fp16 form accepted a caller-written image')
    assert lib.igan_to_pieces(None, x.data_ptr(), None, fake.data_ptr(), 4, 1024, 256) == 3       # IGAN_ERR_UNSUPPORTED
    # ABI v8: the channel maxima a convolution call leaves (x_colmax) and the weight gradient takes: bit-identical to its own passes, also when the
    # call that was asked for them does not write a row image itself (a 1x1 convolution: the maxima come from a pass of their own)
    assert lib.igan_colmax_floats(4, 1024, 256) == 4 + 1024 * 256 and lib.igan_colmax_floats(4, 1024, 144) == 0
    dy = torch.randn(4, 256, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    xcm, dycm = hip_ops.colmax_buffer(x), hip_ops.colmax_buffer(dy)
    y3 = hip_ops.conv2d_raw(x, w, geom, (32, 32), 256, colmax=xcm)
    hip_ops.conv2d_raw(dy, w, hip_ops.dgrad_geom(geom), (32, 32), 256, w_transposed=True, colmax=dycm)
    assert torch.equal(y, y3) and int(xcm[0]) == 256 and float(xcm[4:4 + 256 * 256].reshape(256, 256).max()) == float(x.abs().max())
    dw0 = hip_ops.conv2d_wgrad_raw(x, dy, geom)
    dw1 = hip_ops.conv2d_wgrad_raw(x, dy, geom, x_colmax=xcm, dy_colmax=dycm)
    assert torch.equal(dw0, dw1)
    w1 = (torch.randn(1, 1, 256, 256, generator=g) / 16.0).to(dev)
    xcm2 = hip_ops.colmax_buffer(x)
    hip_ops.conv2d_raw(x, w1, hip_ops.ConvGeom(1, 1, 1, 1, 0, 0), (32, 32), 256, colmax=xcm2)
    assert torch.equal(hip_ops.conv2d_wgrad_raw(x, dy, geom, x_colmax=xcm2, dy_colmax=dycm), dw0)
elif form == 1:
    assert xp is not None and xp.nbytes == x.numel() * 6
    y2 = hip_ops.conv2d_raw(x, w, geom, (32, 32), 256, x_pieces=xp)
    assert torch.equal(y, y2)
    xp.nbytes -= 96
    try:
        hip_ops.conv2d_raw(x, w, geom, (32, 32), 256, x_pieces=xp)
    except Exception as e:
        assert 'x_pieces' in str(e), e
    else:
        raise AssertionError('a piece image of the wrong size was accepted')
    dy = torch.randn(4, 256, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dyp = hip_ops.to_pieces(dy)
    dyp.nbytes += 6
    try:
        hip_ops.conv2d_wgrad_raw(x, dy, geom, dy_pieces=dyp)
    except Exception as e:
        assert 'dy_pieces' in str(e), e
    else:
        raise AssertionError('a dy piece image of the wrong size was accepted')
else:
    assert xp is None
print('SWITCH-OK')
'''


@pytest.mark.parametrize('on', ['1', 'bf16', '0'], ids=['default_piece_form', 'bf16_piece_form', 'exact_fp32'])
def test_switch_selects_the_form_and_piece_images_are_size_checked(cuda_device, on):
    """Default (nothing in the environment) = the piece form for the large 3x3 layers; IGAN_CONV_PLANES=0 = the fp32 instruction
    everywhere (the labelled second bench line).  The library, not the host, decides which tensors get an image
    (igan_conv_pieces_wanted / igan_pieces_image_ok), and it refuses an image whose byte size is not that of the tensor."""
    env = {k: v for k, v in os.environ.items() if k != 'IGAN_CONV_PLANES'}
    if on != '1':
        env['IGAN_CONV_PLANES'] = {'0': '0', 'bf16': '1'}[on]
    r = subprocess.run([sys.executable, '-c', FP32_CHILD % ROOT, on], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'SWITCH-OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    kernel = [l for l in r.stdout.splitlines() if l.startswith('KERNEL ')][-1]
    assert ('planes' in kernel) == (on != '0'), kernel
