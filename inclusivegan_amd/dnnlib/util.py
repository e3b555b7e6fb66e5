"""EasyDict / dotted-name helpers with the reference's contract (dnnlib/util.py:35-48,194-256)."""
import importlib
from typing import Any


class EasyDict(dict):
    """Convenience class that behaves like a dict but allows access with the attribute syntax."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]


def get_obj_by_name(name: str) -> Any:
    """Resolve 'package.module.attr' trying every split point (dnnlib/util.py:194-240)."""
    parts = name.split('.')
    last_err = None
    for i in range(len(parts) - 1, 0, -1):
        mod_name, attrs = '.'.join(parts[:i]), parts[i:]
        try:
            obj = importlib.import_module(mod_name)
        except ModuleNotFoundError as e:
            # only "this prefix is not a module" is skippable; errors raised *inside* a module are not
            if e.name is not None and (mod_name == e.name or mod_name.startswith(e.name + '.')):
                last_err = e
                continue
            raise
        try:
            for a in attrs:
                obj = getattr(obj, a)
            return obj
        except AttributeError as e:
            last_err = e
    raise ImportError('cannot resolve %r: %s' % (name, last_err))


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    func_obj = get_obj_by_name(func_name)
    assert callable(func_obj)
    return func_obj(*args, **kwargs)


def format_time(seconds) -> str:
    """Seconds -> 's' / 'm s' / 'h m s' / 'd h m' string, same output as dnnlib/util.py:110-122
    (used by the per-tick progress line, training_loop.py:499)."""
    total = int(round(float(seconds)))  # round-half-even like np.rint
    minutes, sec = divmod(total, 60)
    hours, minute = divmod(minutes, 60)
    days, hour = divmod(hours, 24)
    if total < 60:
        return '%ds' % sec
    if total < 3600:
        return '%dm %02ds' % (minute, sec)
    if total < 86400:
        return '%dh %02dm %02ds' % (hour, minute, sec)
    return '%dd %02dh %02dm' % (days, hour, minute)
