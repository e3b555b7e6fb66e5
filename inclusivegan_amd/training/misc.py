"""Host-side helpers the hot loop uses from the reference's `training/misc.py`:
adjust_dynamic_range (:36-41) and the NumPy slerp / normalize pair (:190-203) that perturbs the
matched IMLE latents (training_loop.py:447), plus the snapshot helpers the loop's setup touches: pickle wrappers
(:25-31), image grids (:43-74) and `setup_snapshot_image_grid` (:95-143), which advances the training set's iterator and
fixes how many `grid_latents` are drawn from the host random stream before the IMLE candidates (training_loop.py:171,203).
and the resume bookkeeping that reads a snapshot's kimg / elapsed time back from its run directory's log.txt (:147-187)."""
import numpy as np


def adjust_dynamic_range(data, drange_in, drange_out):
    """Affine map of `data` from drange_in to drange_out, computed with fp32 scale / bias."""
    if drange_in != drange_out:
        lo_in, hi_in = np.float32(drange_in[0]), np.float32(drange_in[1])
        lo_out, hi_out = np.float32(drange_out[0]), np.float32(drange_out[1])
        scale = (hi_out - lo_out) / (hi_in - lo_in)
        bias = lo_out - lo_in * scale
        data = data * scale + bias
    return data


def normalize(v):
    return v / np.sqrt(np.sum(np.square(v), axis=-1, keepdims=True))


def slerp(a, b, t):
    """Spherical interpolation of a batch of vectors; the result is unit-norm."""
    a = normalize(a)
    b = normalize(b)
    d = np.sum(a * b, axis=-1, keepdims=True)
    p = t * np.arccos(d)
    c = normalize(b - d * a)
    d = a * np.cos(p) + c * np.sin(p)
    return normalize(d)


# ----------------------------------------------------------------------------
# Pickle wrappers (misc.py:20-31; the URL cache is not offered: there is no network path here).

# The reference pickles (G, D, Gs) as `dnnlib.tflib.network.Network` objects (training_loop.py:518-519, network.py:255-299).
# Opening such a file here maps that class onto a state holder (tflib.network.PickledNetwork) and `dnnlib.util.EasyDict` onto
# this package's EasyDict; `as_networks()` then builds live networks.  Writing with reference_layout=True emits the same
# class path, so the reference's own `misc.load_pkl` can open the file (given the source text of its network module).

_REF_CLASSES = {('dnnlib.tflib.network', 'Network'): ('inclusivegan_amd.dnnlib.tflib.network', 'PickledNetwork'),
                ('dnnlib.util', 'EasyDict'): ('inclusivegan_amd.dnnlib.util', 'EasyDict'),
                ('dnnlib', 'EasyDict'): ('inclusivegan_amd.dnnlib.util', 'EasyDict')}


def load_pkl(filename):
    import importlib
    import pickle

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            module, name = _REF_CLASSES.get((module, name), (module, name))
            return getattr(importlib.import_module(module), name)

    with open(filename, 'rb') as file:
        return _Unpickler(file, encoding='latin1').load()


def as_networks(obj, device=None):
    """Turn every PickledNetwork inside a loaded object (typically the (G, D, Gs) tuple) into a live Network."""
    from ..dnnlib.tflib.network import PickledNetwork
    if isinstance(obj, PickledNetwork):
        return obj.to_network(device=device)
    if isinstance(obj, (tuple, list)):
        return type(obj)(as_networks(o, device) for o in obj)
    return obj


def save_pkl(obj, filename, reference_layout=False, build_module_src=''):
    """pickle.dump(obj) (misc.py:29-31).  reference_layout=True writes Networks under the reference's class path
    `dnnlib.tflib.network.Network` with its version-4 state, `build_module_src` being the text of the reference's
    training/networks_stylegan2.py (read it from a reference checkout; without it the file still loads here, but the
    reference cannot rebuild the graph)."""
    import pickle
    import sys
    import types
    from ..dnnlib.tflib.network import Network
    if not reference_layout:
        with open(filename, 'wb') as file:
            pickle.dump(obj, file, protocol=pickle.HIGHEST_PROTOCOL)
        return

    class _RefNetwork:      # stands where the reference's class will be when the file is opened there
        def __init__(self, state):
            self._state = state

        def __getstate__(self):
            return self._state

    _RefNetwork.__module__ = 'dnnlib.tflib.network'
    _RefNetwork.__qualname__ = _RefNetwork.__name__ = 'Network'

    def convert(o):
        if isinstance(o, Network):
            st = o.state_v4(build_module_src=build_module_src)
            st['components'] = {k: convert(c) for k, c in st['components'].items()}
            return _RefNetwork(st)
        if isinstance(o, (tuple, list)):
            return type(o)(convert(x) for x in o)
        return o

    saved = {k: sys.modules.get(k) for k in ('dnnlib', 'dnnlib.tflib', 'dnnlib.tflib.network')}
    try:
        for k in saved:
            sys.modules[k] = types.ModuleType(k)
        sys.modules['dnnlib.tflib.network'].Network = _RefNetwork
        with open(filename, 'wb') as file:
            pickle.dump(convert(obj), file, protocol=pickle.HIGHEST_PROTOCOL)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def resume_kimg_time(network_pkl):
    """(kimg, seconds) a snapshot was taken at, from its file name and the run's log.txt next to it (misc.py:147-162)."""
    import os
    path, file = os.path.split(network_pkl)
    kimg = str(int(os.path.splitext(file)[0][-6:]))
    s = 0.0
    if not os.path.isfile('%s/log.txt' % path):       # a snapshot copied away from its run directory: resume at its kimg, clock at zero
        return float(kimg), s
    with open('%s/log.txt' % path, 'r') as f:
        for line in f:
            if all(w in line for w in ('tick', 'kimg', 'minibatch', 'time', 'sec/tick', 'sec/kimg', 'maintenance', 'gpumem')) and kimg in line:
                idx = line.find(kimg)
                kimg = float(line[idx:idx + len(kimg) + 2])
                idx = line.find('time')
                try:
                    s = time_to_seconds(line[idx + 5:idx + 5 + 12])
                except ValueError:
                    # the reference's parser cannot read a single-digit leading field ('7s', '3m 05s': it would crash on resuming a
                    # run younger than ten minutes... which takes this engine seconds to produce); read such strings field by field
                    import re
                    found = dict((u, int(v)) for v, u in re.findall(r'(\d+)([dhms])', line[idx + 5:idx + 5 + 12]))
                    s = float(((found.get('d', 0) * 24 + found.get('h', 0)) * 60 + found.get('m', 0)) * 60 + found.get('s', 0))
                break
    return float(kimg), s


def time_to_seconds(string):
    """'1d 02h 03m' / '04m 05s' style durations of dnnlib.util.format_time -> seconds (misc.py:164-187)."""
    def field(ch, width=2):
        i = string.find(ch)
        if i < 0:
            return 0
        if ch == 'd':
            return int(string[:i])
        if ch == 'h' and i == 1:
            return int(string[:i])
        return int(string[i - width:i])
    return float(((field('d') * 24 + field('h')) * 60 + field('m')) * 60 + field('s'))


# ----------------------------------------------------------------------------
# Snapshot image grids (misc.py:43-74,95-143): which reals fill the grid, and how a batch of images is tiled.

def create_image_grid(images, grid_size=None):
    """[num, (C,) H, W] -> [(C,) grid_h*H, grid_w*W], images laid out row by row; empty cells stay zero."""
    assert images.ndim == 3 or images.ndim == 4
    num, img_h, img_w = images.shape[0], images.shape[-2], images.shape[-1]
    if grid_size is not None:
        grid_w, grid_h = tuple(grid_size)
    else:
        grid_w = max(int(np.ceil(np.sqrt(num))), 1)
        grid_h = max((num - 1) // grid_w + 1, 1)
    grid = np.zeros(list(images.shape[1:-2]) + [grid_h * img_h, grid_w * img_w], dtype=images.dtype)
    for i in range(num):
        col, row = i % grid_w, i // grid_w
        grid[..., row * img_h:(row + 1) * img_h, col * img_w:(col + 1) * img_w] = images[i]
    return grid


def convert_to_pil_image(image, drange=[0, 1]):
    import PIL.Image
    assert image.ndim == 2 or image.ndim == 3
    if image.ndim == 3:
        image = image[0] if image.shape[0] == 1 else image.transpose(1, 2, 0)    # grayscale CHW -> HW, else CHW -> HWC
    image = adjust_dynamic_range(image, drange, [0, 255])
    image = np.rint(image).clip(0, 255).astype(np.uint8)
    return PIL.Image.fromarray(image, 'RGB' if image.ndim == 3 else 'L')


def save_image_grid(images, filename, drange=[0, 1], grid_size=None):
    convert_to_pil_image(create_image_grid(images, grid_size), drange).save(filename)


def apply_mirror_augment(minibatch):
    mask = np.random.rand(minibatch.shape[0]) < 0.5
    minibatch = np.array(minibatch)
    minibatch[mask] = minibatch[mask, :, :, ::-1]
    return minibatch


_GRID_SPANS = {'1080p': (1920, 1080, 3, 2), '4k': (3840, 2160, 7, 4), '8k': (7680, 4320, 7, 4)}     # width, height, min columns, min rows


def setup_snapshot_image_grid(training_set, size='1080p', layout='random'):
    """-> ((gw, gh), reals [gw*gh, C, H, W], labels [gw*gh, label_size]).  'random' = the next gw*gh images of the
    training set's iterator; class layouts ('row_per_class', 'col_per_class', 'class4x4') fill blocks by label arg-max."""
    gw = gh = 1
    if size in _GRID_SPANS:
        span_w, span_h, min_w, min_h = _GRID_SPANS[size]
        gw = np.clip(span_w // training_set.shape[2], min_w, 32)
        gh = np.clip(span_h // training_set.shape[1], min_h, 32)
    reals = np.zeros([gw * gh] + training_set.shape, dtype=training_set.dtype)
    labels = np.zeros([gw * gh, training_set.label_size], dtype=training_set.label_dtype)
    if layout == 'random':
        reals[:], labels[:] = training_set.get_minibatch_np(gw * gh)
    class_layouts = dict(row_per_class=[gw, 1], col_per_class=[1, gh], class4x4=[4, 4])
    if layout in class_layouts:
        bw, bh = class_layouts[layout]
        nw, nh = (gw - 1) // bw + 1, (gh - 1) // bh + 1
        blocks = [[] for _ in range(nw * nh)]
        for _ in range(1000000):
            real, label = training_set.get_minibatch_np(1)
            idx = np.argmax(label[0])
            while idx < len(blocks) and len(blocks[idx]) >= bw * bh:
                idx += training_set.label_size
            if idx < len(blocks):
                blocks[idx].append((real, label))
                if all(len(block) >= bw * bh for block in blocks):
                    break
        for i, block in enumerate(blocks):
            for j, (real, label) in enumerate(block):
                x = (i % nw) * bw + j % bw
                y = (i // nw) * bh + j // bw
                if x < gw and y < gh:
                    reals[x + y * gw] = real[0]
                    labels[x + y * gw] = label[0]
    return (gw, gh), reals, labels
