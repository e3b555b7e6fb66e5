"""ctypes binding of include/igan_hip.h (libigan_hip.so).

This is the only place the product touches native code.  It plays the role of
`dnnlib/tflib/custom_ops.py:87-167` (`get_plugin`) in the reference: locate the
compiled plugin, load it once per process, hand out callables.  Unlike the
reference there is no JIT: the library is built ahead of time for gfx950 by
`__graft_entry__.build()` / `make -C inclusivegan_amd/csrc`.

There is deliberately NO fallback: if the shared library is missing or a symbol
is absent, importing a kernel raises.  The CPU oracle under `oracle/` is test
infrastructure and is never imported from here.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IGAN_LIB selects another build of the same library (A/B runs of compile-time kernel variants: make VARIANT=...)
LIB_PATH = os.environ.get('IGAN_LIB') or os.path.join(_HERE, 'csrc', 'libigan_hip.so')

IGAN_OK = 0
IGAN_ERR_INVALID_ARGUMENT = 1
IGAN_ERR_HIP = 2
IGAN_ERR_UNSUPPORTED = 3

c_float_p = ctypes.POINTER(ctypes.c_float)


class UpFirDn2DParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('k', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('upx', ctypes.c_int), ('upy', ctypes.c_int), ('downx', ctypes.c_int), ('downy', ctypes.c_int),
        ('padx0', ctypes.c_int), ('padx1', ctypes.c_int), ('pady0', ctypes.c_int), ('pady1', ctypes.c_int),
        ('majorDim', ctypes.c_int), ('inH', ctypes.c_int), ('inW', ctypes.c_int), ('minorDim', ctypes.c_int),
        ('kernelH', ctypes.c_int), ('kernelW', ctypes.c_int),
        ('outH', ctypes.c_int), ('outW', ctypes.c_int),
    ]


class FusedBiasActParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('b', ctypes.c_void_p), ('ref', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('grad', ctypes.c_int), ('act', ctypes.c_int),
        ('alpha', ctypes.c_float), ('gain', ctypes.c_float),
        ('sizeX', ctypes.c_int), ('sizeB', ctypes.c_int), ('stepB', ctypes.c_int),
    ]


class Conv2DParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('in_scale', ctypes.c_void_p), ('out_scale', ctypes.c_void_p),
        ('workspace', ctypes.c_void_p), ('workspace_floats', ctypes.c_size_t),
        ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('Cin', ctypes.c_int),
        ('OH', ctypes.c_int), ('OW', ctypes.c_int), ('Cout', ctypes.c_int),
        ('KH', ctypes.c_int), ('KW', ctypes.c_int),
        ('stride', ctypes.c_int), ('up', ctypes.c_int),
        ('pad_y', ctypes.c_int), ('pad_x', ctypes.c_int),
        ('w_transposed', ctypes.c_int), ('splits', ctypes.c_int), ('sliced_tiles', ctypes.c_int), ('alpha', ctypes.c_float),
        ('bias', ctypes.c_void_p), ('act', ctypes.c_int), ('act_alpha', ctypes.c_float), ('act_gain', ctypes.c_float),
        ('noise', ctypes.c_void_p), ('noise_strength', ctypes.c_void_p), ('noise_bcast', ctypes.c_int),
        ('x_pieces', ctypes.c_void_p), ('x_pieces_bytes', ctypes.c_size_t),
        ('x_colmax', ctypes.c_void_p),
        ('w_pieces', ctypes.c_void_p), ('w_pieces_bytes', ctypes.c_size_t),
    ]

    def __init__(self, *args, **kwargs):
        kwargs.setdefault('alpha', 1.0)     # a plain convolution
        super().__init__(*args, **kwargs)


class Conv2DWgradParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p),
        ('in_scale', ctypes.c_void_p), ('out_scale', ctypes.c_void_p),
        ('workspace', ctypes.c_void_p), ('workspace_floats', ctypes.c_size_t),
        ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('Cin', ctypes.c_int),
        ('OH', ctypes.c_int), ('OW', ctypes.c_int), ('Cout', ctypes.c_int),
        ('KH', ctypes.c_int), ('KW', ctypes.c_int),
        ('stride', ctypes.c_int), ('up', ctypes.c_int),
        ('pad_y', ctypes.c_int), ('pad_x', ctypes.c_int),
        ('splits', ctypes.c_int), ('alpha', ctypes.c_float),
        ('x_pieces', ctypes.c_void_p), ('dy_pieces', ctypes.c_void_p), ('x_pieces_bytes', ctypes.c_size_t), ('dy_pieces_bytes', ctypes.c_size_t),
        ('x_colmax', ctypes.c_void_p), ('dy_colmax', ctypes.c_void_p),
    ]

    def __init__(self, *args, **kwargs):
        kwargs.setdefault('alpha', 1.0)     # a plain convolution
        super().__init__(*args, **kwargs)


class DenseParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('x2', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('bias', ctypes.c_void_p), ('e1', ctypes.c_void_p), ('e2', ctypes.c_void_p), ('colsum', ctypes.c_void_p),
        ('ldx', ctypes.c_int), ('ldy', ctypes.c_int),
        ('M', ctypes.c_int), ('K', ctypes.c_int), ('N', ctypes.c_int),
        ('w_transposed', ctypes.c_int),
        ('prologue', ctypes.c_int), ('epilogue', ctypes.c_int),
        ('alpha', ctypes.c_float), ('pro_scale', ctypes.c_float), ('bias_scale', ctypes.c_float),
        ('add_const', ctypes.c_float), ('eps', ctypes.c_float),
    ]


class DenseWgradParams(ctypes.Structure):
    _fields_ = [
        ('a', ctypes.c_void_p), ('b', ctypes.c_void_p), ('b2', ctypes.c_void_p), ('dw', ctypes.c_void_p),
        ('lda', ctypes.c_int),
        ('M', ctypes.c_int), ('K', ctypes.c_int), ('N', ctypes.c_int),
        ('pro_a', ctypes.c_int), ('pro_b', ctypes.c_int),
        ('alpha', ctypes.c_float), ('pro_scale', ctypes.c_float),
    ]


class TapsParams(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('v', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('taps', ctypes.c_int), ('n', ctypes.c_int), ('scale', ctypes.c_float)]


ABI_VERSION = 9      # include/igan_hip.h IGAN_ABI_VERSION
STRUCTS = (UpFirDn2DParams, FusedBiasActParams, Conv2DParams, Conv2DWgradParams, DenseParams, DenseWgradParams, TapsParams)   # igan_struct_size ids

DENSE_MAX_GROUPS = 24
DENSE_PRO_NONE, DENSE_PRO_SQUARE, DENSE_PRO_DEMOD_GRAD = 0, 1, 2
DENSE_EPI_SCALE, DENSE_EPI_BIAS, DENSE_EPI_RSQRT, DENSE_EPI_STYLE_GRAD = 0, 1, 2, 3

_I, _F, _P, _SZ = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes): every symbol include/igan_hip.h declares.
SIGNATURES = {
    'igan_abi_version': (_I, []),
    'igan_struct_size': (_SZ, [_I]),
    'igan_last_error': (ctypes.c_char_p, []),
    'igan_upfirdn2d': (_I, [_P, ctypes.POINTER(UpFirDn2DParams)]),
    'igan_upfirdn2d_f16': (_I, [_P, ctypes.POINTER(UpFirDn2DParams)]),
    'igan_upfirdn2d_ban': (_I, [_P, _P, _P, _P, _I, _P, _I, _F, _F]),
    'igan_fused_bias_act': (_I, [_P, ctypes.POINTER(FusedBiasActParams)]),
    'igan_fused_bias_act_f16': (_I, [_P, ctypes.POINTER(FusedBiasActParams)]),
    'igan_bias_grad_workspace_floats': (_SZ, [_I, _I, _I]),
    'igan_bias_grad': (_I, [_P, _P, _P, _P, _I, _I, _I]),
    'igan_bias_act_noise_workspace_floats': (_SZ, [_I, _I]),
    'igan_bias_act_noise_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F]),
    'igan_bias_act_noise_dd_workspace_floats': (ctypes.c_size_t, [_I, _I, _I]),
    'igan_bias_act_noise_bwd_dd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F]),
    'igan_bias_act_noise_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F]),
    'igan_conv2d_plan': (_I, [ctypes.POINTER(Conv2DParams), ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_SZ)]),
    'igan_conv2d': (_I, [_P, ctypes.POINTER(Conv2DParams)]),
    'igan_conv2d_kernel_name': (_I, [ctypes.POINTER(Conv2DParams), ctypes.c_char_p, _I]),
    'igan_conv2d_wgrad_kernel_name': (_I, [ctypes.POINTER(Conv2DWgradParams), ctypes.c_char_p, _I]),
    'igan_conv2d_wgrad_plan': (_I, [ctypes.POINTER(Conv2DWgradParams), ctypes.POINTER(_I), ctypes.POINTER(_SZ)]),
    'igan_conv2d_wgrad': (_I, [_P, ctypes.POINTER(Conv2DWgradParams)]),
    'igan_debug_f16_window_by_kind': (_I, [_P, _I]),
    'igan_colmax_floats': (ctypes.c_size_t, [_I, _I, _I]),
    'igan_filter_image_bytes': (ctypes.c_size_t, [_I, _I, _I, _I]),
    'igan_filter_image': (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    'igan_to_pieces': (_I, [_P, _P, _P, _P, _I, _I, _I]),
    'igan_conv_pieces_wanted': (_I, [_I, _I, _I, _I]),
    'igan_pieces_image_ok': (_I, [_I, _I, _I]),
    'igan_conv_piece_form': (_I, []),
    'igan_debug_f16_window': (_I, [_P, _P, _I]),
    'igan_dense_small': (_I, [_P, ctypes.POINTER(DenseParams)]),
    'igan_dense_small_wgrad': (_I, [_P, ctypes.POINTER(DenseWgradParams)]),
    'igan_dense_small_grouped': (_I, [_P, ctypes.POINTER(DenseParams), _I]),
    'igan_dense_small_wgrad_grouped': (_I, [_P, ctypes.POINTER(DenseWgradParams), _I]),
    'igan_sumsq_taps_grouped': (_I, [_P, ctypes.POINTER(TapsParams), _I]),
    'igan_bcast_mul_taps_grouped': (_I, [_P, ctypes.POINTER(TapsParams), _I]),
    'igan_sumsq_taps': (_I, [_P, _P, _P, _I, _I]),
    'igan_bcast_mul_taps': (_I, [_P, _P, _P, _P, _I, _I, _F]),
    'igan_scale_dot_workspace_floats': (_SZ, [_I, _I, _I]),
    'igan_scale_dot': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    'igan_lpips_layer_blocks': (_I, [_I, _I]),
    'igan_lpips_layer_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    'igan_lpips_layer_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I]),
    'igan_lpips_pairs_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    'igan_lpips_pairs_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    'igan_maxpool2x2_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I]),
    'igan_maxpool2x2_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I]),
    'igan_mbstd_workspace_floats': (_SZ, [_I, _I, _I, _I, _I]),
    'igan_mbstd_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    'igan_mbstd_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    'igan_row_sqnorm': (_I, [_P, _P, _P, _I, _I]),
    'igan_nn1_update': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    'igan_stamp': (_I, [_P, _P]),
    'igan_debug_set_conv_diag': (None, [_P]),
    'igan_stamp_accumulate': (_I, [_P, _P, _P, _I, _I]),
    'igan_finite_check': (_I, [_P, _P, _I, _P]),
    'igan_adam_step': (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _P]),
    'igan_ema': (_I, [_P, _P, _P, _I, _F]),
    'igan_summary_accumulate': (_I, [_P, _P, _I, _P]),
}

_lib = None
_lock = threading.Lock()


class IganError(RuntimeError):
    """HIP-side failure (reference: errors::Internal -> tf.errors.InternalError)."""


def get_plugin():
    """Load libigan_hip.so once per process (reference: custom_ops.get_plugin)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                'inclusivegan_amd: %s is missing. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                'or `make -C inclusivegan_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
        # The library's HIP runtime must be the one PyTorch drives (streams and device pointers are handed across): both name
        # it libamdhip64.so.7, so whichever copy is loaded first serves both -- and it has to be the one PyTorch ships, because
        # PyTorch loads its own copy by path.  Loading this library first used to leave two runtimes in the process
        # (every launch then failed with hipErrorNoDevice).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is absent
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.igan_abi_version() != ABI_VERSION:
            raise ImportError('inclusivegan_amd: libigan_hip.so has ABI version %d, this binding needs %d -- rebuild it '
                              '(make -C inclusivegan_amd/csrc)' % (lib.igan_abi_version(), ABI_VERSION))
        for which, struct in enumerate(STRUCTS):      # a stale library must never be driven with a newer struct layout
            if lib.igan_struct_size(which) != ctypes.sizeof(struct):
                raise ImportError('inclusivegan_amd: libigan_hip.so was built with sizeof(%s) = %d, this binding declares %d -- '
                                  'rebuild it' % (struct.__name__, lib.igan_struct_size(which), ctypes.sizeof(struct)))
        _lib = lib
    return _lib


def check(rc):
    """Translate a status code into the exception the reference would surface."""
    if rc == IGAN_OK:
        return
    msg = get_plugin().igan_last_error().decode('utf-8', 'replace')
    if rc == IGAN_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)       # tf.errors.InvalidArgumentError
    if rc == IGAN_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise IganError(msg)            # tf.errors.InternalError
