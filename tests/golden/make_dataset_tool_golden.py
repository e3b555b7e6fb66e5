"""Golden vectors for the two dataset builders, produced by EXECUTING the reference's own functions.

`dataset_tool.py` imports TensorFlow and scikit-image at module level, so `create_mnistrgb` (:307-334) and `create_celeba` (:447-486)
are cut out of the file's syntax tree (read from /root/reference at generation time only) and executed unchanged with a recording
stand-in for `TFRecordExporter` (its `choose_shuffled_order` is the reference's: RandomState(123) shuffle of arange) on seeded
synthetic inputs (tests/util.py synthetic_mnist / synthetic_celeba: the real data sets are not in the image).

Output: tests/golden/dataset_tool_golden.npz -- per image handed to the exporter its SHA-1, and the label arrays.
Run from the repo root:  python tests/golden/make_dataset_tool_golden.py
"""
import ast
import glob
import hashlib
import os
import sys
import tempfile

import numpy as np
import PIL.Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
from tests.util import synthetic_mnist, synthetic_celeba  # noqa: E402

MNIST_IMAGES = 400
CELEBA_CASES = [dict(shuffle=0, num_images=0, num_shifts=0), dict(shuffle=1, num_images=5, num_shifts=2), dict(shuffle=0, num_images=4, num_shifts=0, cx=80, cy=110)]


class Recorder:
    def __init__(self, tfrecord_dir, expected_images, **kw):
        self.expected_images, self.images, self.labels, self.cur_images = expected_images, [], None, 0

    def choose_shuffled_order(self):          # dataset_tool.py:59-62
        order = np.arange(self.expected_images)
        np.random.RandomState(123).shuffle(order)
        return order

    def add_image(self, img):
        self.images.append(np.array(img))
        self.cur_images += 1

    def add_labels(self, labels):
        self.labels = np.array(labels)

    def __enter__(self):
        Recorder.last = self
        return self

    def __exit__(self, *a):
        pass


def main():
    path = os.path.join(REF, 'dataset_tool.py')
    mod = ast.parse(open(path).read(), filename=path)
    fns = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name in ('create_mnistrgb', 'create_celeba')]
    ns = dict(np=np, os=os, glob=glob, PIL=PIL, TFRecordExporter=Recorder, print=lambda *a, **k: None)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    sha = lambda a: np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        synthetic_mnist(os.path.join(tmp, 'mnist'))
        ns['create_mnistrgb'](os.path.join(tmp, 'out'), os.path.join(tmp, 'mnist'), num_images=MNIST_IMAGES, random_seed=123)
        r = Recorder.last
        assert all(i.shape == (3, 32, 32) and i.dtype == np.uint8 for i in r.images) and r.labels.shape == (MNIST_IMAGES, 1000)
        out['mnistrgb_sha1'] = np.stack([sha(i) for i in r.images])
        out['mnistrgb_numbers'] = np.argmax(r.labels, axis=1).astype(np.int32)
        assert np.array_equal(r.labels.sum(axis=1), np.ones(MNIST_IMAGES)) and r.labels.dtype == np.float32
        out['mnistrgb_num_images'] = np.array(MNIST_IMAGES)
        img_dir = synthetic_celeba(tmp)
        cwd = os.getcwd()
        os.chdir(tmp)                             # the reference opens 'celeba/Anno/list_attr_celeba.txt' relative to the working directory
        try:
            for j, kw in enumerate(CELEBA_CASES):
                ns['create_celeba'](os.path.join(tmp, 'out%d' % j), img_dir, **kw)
                r = Recorder.last
                out['celeba_%d_sha1' % j] = np.stack([sha(i) for i in r.images])
                out['celeba_%d_labels' % j] = r.labels
                out['celeba_%d_args' % j] = np.array([kw.get('cx', 89), kw.get('cy', 121), kw['shuffle'], kw['num_images'], kw['num_shifts']])
                assert r.labels.dtype == np.float32 and set(np.unique(r.labels)) <= {0.0, 1.0}
        finally:
            os.chdir(cwd)
    out['celeba_cases'] = np.array(len(CELEBA_CASES))
    p = os.path.join(HERE, 'dataset_tool_golden.npz')
    np.savez_compressed(p, **out)
    print('wrote %s (%.1f KB)' % (p, os.path.getsize(p) / 1e3))


if __name__ == '__main__':
    main()
