"""CPU: the C-ABI library loads without a GPU and exports every symbol include/igan_hip.h declares;
argument validation returns the reference's error class before anything touches a device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'igan_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(igan_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    from inclusivegan_amd import _abi
    lib = _abi.get_plugin()
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_abi.SIGNATURES) == declared
    header = open(os.path.join(ROOT, 'include', 'igan_hip.h')).read()
    assert lib.igan_abi_version() == _abi.ABI_VERSION == int(re.search(r'#define IGAN_ABI_VERSION (\d+)', header).group(1))
    # the library and the binding agree on every struct size (a stale .so must not load)
    for which, struct in enumerate(_abi.STRUCTS):
        assert lib.igan_struct_size(which) == ctypes.sizeof(struct), struct.__name__
    assert lib.igan_struct_size(len(_abi.STRUCTS)) == 0


def test_struct_layouts_match_header_field_order():
    from inclusivegan_amd import _abi
    text = open(os.path.join(ROOT, 'include', 'igan_hip.h')).read()
    for cname, cls in [('igan_upfirdn2d_params', _abi.UpFirDn2DParams), ('igan_fused_bias_act_params', _abi.FusedBiasActParams),
                       ('igan_conv2d_params', _abi.Conv2DParams), ('igan_conv2d_wgrad_params', _abi.Conv2DWgradParams),
                       ('igan_dense_params', _abi.DenseParams), ('igan_dense_wgrad_params', _abi.DenseWgradParams),
                       ('igan_taps_params', _abi.TapsParams)]:
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                names.append(re.findall(r'([A-Za-z_0-9]+)\s*$', part.strip())[0])
        assert names == [f[0] for f in cls._fields_], cname
    # the enums the dense entry points switch on
    for name, val in re.findall(r'(IGAN_DENSE_[A-Z_]+) = (\d+)', text):
        assert getattr(_abi, name[len('IGAN_'):]) == int(val), name
    assert _abi.DENSE_MAX_GROUPS == int(re.search(r'#define IGAN_DENSE_MAX_GROUPS (\d+)', text).group(1))


def test_invalid_arguments_are_rejected_without_a_device():
    from inclusivegan_amd import _abi
    lib = _abi.get_plugin()
    k = np.ones((4, 4), np.float32)
    dummy = ctypes.c_void_p(16)   # never dereferenced: validation fails first
    p = _abi.UpFirDn2DParams(x=16, k=k.ctypes.data, y=16, upx=0, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0,
                             majorDim=1, inH=4, inW=4, minorDim=4, kernelH=4, kernelW=4, outH=1, outW=1)
    assert lib.igan_upfirdn2d(None, ctypes.byref(p)) == _abi.IGAN_ERR_INVALID_ARGUMENT
    assert b'upx and upy must be at least 1x1' in lib.igan_last_error()
    with pytest.raises(ValueError):
        _abi.check(_abi.IGAN_ERR_INVALID_ARGUMENT)
    p.upx = 1
    p.padx0 = -8
    assert lib.igan_upfirdn2d(None, ctypes.byref(p)) == _abi.IGAN_ERR_INVALID_ARGUMENT
    assert b'output must be at least 1x1' in lib.igan_last_error()
    fp = _abi.FusedBiasActParams(x=16, b=None, ref=None, y=16, grad=1, act=3, alpha=0.2, gain=1.0, sizeX=8, sizeB=0, stepB=1)
    assert lib.igan_fused_bias_act(None, ctypes.byref(fp)) == _abi.IGAN_ERR_INVALID_ARGUMENT   # ref missing for grad=1
    cp = _abi.Conv2DParams(x=16, w=16, y=16, N=1, H=4, W=4, Cin=4, OH=4, OW=4, Cout=4, KH=3, KW=3, stride=2, up=2, pad_y=1, pad_x=1)
    assert lib.igan_conv2d(None, ctypes.byref(cp)) == _abi.IGAN_ERR_INVALID_ARGUMENT
    cp.stride = 1; cp.up = 3
    assert lib.igan_conv2d(None, ctypes.byref(cp)) == _abi.IGAN_ERR_UNSUPPORTED
    del dummy


def test_plans_are_host_only():
    from inclusivegan_amd import _abi
    lib = _abi.get_plugin()
    # Tiles are dealt to 256 CUs.  4x4 layer of config-e at batch 6: M = 96 -> 4 tiles of 128x128, all sliced
    # along K; 128x128 layer at batch 6: 768 tiles = 3 whole rounds -> nothing to slice; a 32x32 layer at
    # batch 18 (VGG conv3): 288 tiles -> only the 32 tiles of the second round are sliced.
    small = _abi.Conv2DParams(x=16, w=16, y=16, N=6, H=4, W=4, Cin=512, OH=4, OW=4, Cout=512, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1)
    big = _abi.Conv2DParams(x=16, w=16, y=16, N=6, H=128, W=128, Cin=128, OH=128, OW=128, Cout=128, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1)
    mid = _abi.Conv2DParams(x=16, w=16, y=16, N=18, H=32, W=32, Cin=256, OH=32, OW=32, Cout=256, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1)
    s, sl, ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    # the large 3x3 layers run in the bf16-piece form by default: their plans add the piece images of x and of the filter (6 B per element)
    on = lib.igan_conv_pieces_wanted(3, 3, 128, 128)
    assert on == (0 if os.environ.get('IGAN_CONV_PLANES') == '0' else 1)
    form = lib.igan_conv_piece_form()
    a256 = lambda b: (b + 255) // 256 * 256
    if form == 2:       # fp16 form: a row image of x (4 B per element + 1 / S per pixel) and a filter image (4 B per element + 1 / S per output channel + column-maximum partials)
        img = lambda p: (a256(4 * p.N * p.H * p.W * p.Cin + 4 * p.N * p.H * p.W) + a256(4 * p.KH * p.KW * p.Cin * p.Cout + 4 * p.Cout + 4 * p.KH * p.KW * 2 * p.Cout)) // 4
        wimg = lambda P, C: a256(4 * P * C + 8 * C + 4 * 1024 * C) // 4      # weight gradient: a column image per operand (+ 1 / S and S per channel, 1024 rows of partials)
    else:
        img = lambda p: on * (p.N * p.H * p.W * p.Cin * 6 // 4 + p.KH * p.KW * p.Cin * p.Cout * 6 // 4)
        wimg = lambda P, C: on * (P * C * 6 // 4)
    assert lib.igan_conv2d_plan(ctypes.byref(small), ctypes.byref(s), ctypes.byref(sl), ctypes.byref(ws)) == 0
    assert s.value > 1 and sl.value == 4 and ws.value == 4 * s.value * 128 * 128
    assert lib.igan_conv2d_plan(ctypes.byref(big), ctypes.byref(s), ctypes.byref(sl), ctypes.byref(ws)) == 0
    assert s.value == 1 and sl.value == 0 and ws.value == img(big)
    assert lib.igan_conv2d_plan(ctypes.byref(mid), ctypes.byref(s), ctypes.byref(sl), ctypes.byref(ws)) == 0
    assert s.value > 1 and sl.value in (32, 288) and ws.value == sl.value * s.value * 128 * 128 + img(mid)
    wg = _abi.Conv2DWgradParams(x=16, dy=16, dw=16, N=6, H=128, W=128, Cin=128, OH=128, OW=128, Cout=128, KH=3, KW=3, stride=1, up=1, pad_y=1, pad_x=1)
    assert lib.igan_conv2d_wgrad_plan(ctypes.byref(wg), ctypes.byref(s), ctypes.byref(ws)) == 0
    assert s.value > 1 and ws.value == s.value * 9 * 128 * 128 + 2 * wimg(6 * 128 * 128, 128)


def test_no_cpu_fallback():
    import torch
    from inclusivegan_amd import hip_ops
    with pytest.raises(RuntimeError, match='no CPU path'):
        hip_ops.upfirdn2d_raw(torch.zeros(1, 4, 4, 4), np.ones((4, 4), np.float32), 1, 1, 1, 1, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match='no CPU path'):
        hip_ops.conv2d_raw(torch.zeros(1, 4, 4, 4), torch.zeros(3, 3, 4, 4), hip_ops.ConvGeom(3, 3, 1, 1, 1, 1), (4, 4), 4)


def test_small_dense_path_is_bounded_by_its_32_bit_offsets():
    """dense_small_kernel reads its operands through buffer descriptors with 32-bit byte offsets and an out-of-range marker of
    0x7FFFFFF0, so an operand of 2 GiB or more must not reach it -- DCI's threshold re-rank (dci.py: one query row against 8192 candidates) does with
    an unprojected dimension of 256*256*3.  The dispatcher sends such a call to the MFMA tiles; the direct entry point rejects it."""
    from inclusivegan_amd import _abi
    lib = _abi.get_plugin()

    def kernel(M, K, N, wt):
        p = _abi.Conv2DParams(x=1 << 20, w=1 << 20, y=1 << 20, in_scale=None, out_scale=None, workspace=None, workspace_floats=0, N=M, H=1, W=1, Cin=K, OH=1, OW=1,
                              Cout=N, KH=1, KW=1, stride=1, up=1, pad_y=0, pad_x=0, w_transposed=wt, splits=1, alpha=1.0, bias=None, act=0, act_alpha=0.0, act_gain=1.0)
        buf = ctypes.create_string_buffer(128)
        _abi.check(lib.igan_conv2d_kernel_name(ctypes.byref(p), buf, 128))
        return buf.value.decode()

    assert kernel(1, 3072, 8192, 1).startswith('dense_small_kernel')             # CelebA-shaped 32x32x3: 100 MB of candidates
    assert kernel(1, 49152, 8192, 1).startswith('dense_small_kernel')            # 128x128x3: 1.6 GB, still below the marker
    assert kernel(64, 512, 512, 0).startswith('dense_small_kernel')
    for K in (196608, 65536):                                                     # 256x256x3: 6.4 GB; exactly 2 GiB: refused by igan_conv2d as a whole ...
        with pytest.raises(ValueError, match='2 GiB'):
            kernel(1, K, 8192, 1)
    from inclusivegan_amd.dci_code.dci import DCI                                 # ... so DCI folds fewer candidates per pass at such a dimension
    for dim in (3072, 49152, 196608):
        db = DCI(dim, device='cpu')
        assert db.cand_chunk * dim * 4 < 0x7FFFFFF0 and db.query_chunk * dim * 4 < 0x7FFFFFF0
        assert (db.cand_chunk, db.query_chunk) == ((8192, 4096) if dim <= 49152 else (2729, 2729))
    d = _abi.DenseParams()
    d.x = d.w = d.y = 1 << 20
    d.M, d.K, d.N, d.ldx, d.ldy, d.w_transposed = 1, 196608, 8192, 196608, 8192, 1
    d.prologue, d.epilogue, d.alpha = _abi.DENSE_PRO_NONE, _abi.DENSE_EPI_SCALE, 1.0
    assert lib.igan_dense_small(None, ctypes.byref(d)) != 0
    assert b'2 GiB' in lib.igan_last_error()
