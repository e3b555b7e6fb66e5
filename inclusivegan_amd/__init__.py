"""InclusiveGAN training engine for AMD MI355X (gfx950).

Package layout mirrors the reference tree for the hot path only, so that the
reference's dotted-name plugin mechanism (dnnlib/util.py:251-256) can point at it:

    inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d / fused_bias_act   (custom ops)
    inclusivegan_amd.training.networks_stylegan2 / loss / training_loop
    inclusivegan_amd.dci_code.dci.DCI                               (nearest neighbour)

Native code lives in csrc/ (hand-written HIP for CDNA4) behind the C ABI declared in
include/igan_hip.h and bound in _abi.py.
"""
__version__ = '0.1.0'
