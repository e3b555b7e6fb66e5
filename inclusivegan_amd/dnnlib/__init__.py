"""Minimal stand-in for the reference's `dnnlib` surface used on the hot path
(dnnlib/__init__.py, dnnlib/util.py:35-48,194-256): EasyDict and dotted-name resolution."""
from .util import EasyDict, get_obj_by_name, call_func_by_name, format_time

submit_config = None  # set by training.training_loop callers (reference: dnnlib.submit_config)
