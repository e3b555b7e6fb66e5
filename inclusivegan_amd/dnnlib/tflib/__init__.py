"""The slice of `dnnlib.tflib` the hot path programs against (dnnlib/tflib/__init__.py:7-18)."""
from . import tfutil
from .tfutil import lerp, lerp_clip, slerp, normalize
from .network import Network
from .optimizer import Optimizer, SimpleAdam
