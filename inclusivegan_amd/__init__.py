"""InclusiveGAN training engine for AMD MI355X (gfx950).

Package layout mirrors the reference tree for the hot path only, so that the
reference's dotted-name plugin mechanism (dnnlib/util.py:251-256) can point at it:

    inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d / fused_bias_act   (custom ops)
    inclusivegan_amd.training.networks_stylegan2 / loss / training_loop
    inclusivegan_amd.dci_code.dci.DCI                               (nearest neighbour)

Native code lives in csrc/ (hand-written HIP for CDNA4) behind the C ABI declared in
include/igan_hip.h and bound in _abi.py.
"""
import os as _os

# ROCm's "graph packet capture" (AQL packets of a hipGraph pre-built at instantiation; on by default in the HIP runtime this
# PyTorch ships) replays the training ops' graphs WRONGLY on MI355X: a reduction deep in the second-order backward reads its
# input before the producing kernel of the same replay has written it (found by tests/test_gpu_loop_parity.py; evidence in
# profiles/r03_graph_packet_capture.txt).  The runtime reads its flags at the first HIP call, so the switch has to be in the
# environment before anything touches the GPU; dnnlib/tflib/graphs.py additionally validates every captured op against
# its eager execution and refuses to replay a graph that disagrees.
if _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') is None:
    import sys as _sys
    _t = _sys.modules.get('torch')
    if _t is not None and _t.cuda.is_initialized():
        # the runtime has read its flags already: setting the variable now would silently do nothing
        import warnings as _w
        _w.warn('inclusivegan_amd imported after HIP was initialised: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 cannot take effect in this process; '
                'captured training ops are validated against eager execution and fall back to it if unfaithful (import the package first, or export the variable)')
    _os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'

__version__ = '0.1.0'
