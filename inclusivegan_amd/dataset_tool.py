"""The two dataset builders the InclusiveGAN configs are trained from, on top of the TFRecord exporter of training/tfrecord.py
(no TensorFlow): `create_mnistrgb` (Stacked MNIST: three random digits as the R, G, B planes of a 32x32 image, label = the
three-digit number as a 1000-way one-hot; reference dataset_tool.py:307-334) and `create_celeba` (218x178 aligned CelebA PNGs
cropped to 128x128 around (cx, cy) = (89, 121), the 40 binary attributes with -1 -> 0 as labels; :447-486).  Same arguments,
same order of random draws, same files as the reference's tool, so a directory written here is one the reference reads.

    python -m inclusivegan_amd.dataset_tool create_mnistrgb <tfrecord_dir> <mnist_dir> [--num_images N] [--random_seed S]
    python -m inclusivegan_amd.dataset_tool create_celeba   <tfrecord_dir> <celeba_dir> [--cx 89] [--cy 121] [--shuffle 0] [--num_images 0] [--num_shifts 0] [--export_attr 1]

PINNED: tests/golden/dataset_tool_golden.npz holds what the reference's own two functions hand to their exporter on seeded synthetic
inputs (tests/golden/make_dataset_tool_golden.py); tests/test_dataset_tool.py requires the directories written here to hold exactly that.
"""
import argparse
import glob
import gzip
import os
import sys

import numpy as np

from .training.tfrecord import TFRecordExporter

CELEBA_ATTR_FILE = 'celeba/Anno/list_attr_celeba.txt'      # relative to the working directory, as in the reference (:467)


def create_mnistrgb(tfrecord_dir, mnist_dir, num_images=1000000, random_seed=123):
    print('Loading MNIST from "%s"' % mnist_dir)
    with gzip.open(os.path.join(mnist_dir, 'train-images-idx3-ubyte.gz'), 'rb') as f:
        digits = np.frombuffer(f.read(), np.uint8, offset=16).reshape(-1, 28, 28)
    with gzip.open(os.path.join(mnist_dir, 'train-labels-idx1-ubyte.gz'), 'rb') as f:
        digit_labels = np.frombuffer(f.read(), np.uint8, offset=8)
    digits = np.pad(digits, [(0, 0), (2, 2), (2, 2)], 'constant', constant_values=0)          # 28 -> 32 (:315)
    if digits.shape != (60000, 32, 32) or digit_labels.shape != (60000,):
        raise ValueError('create_mnistrgb: expected the 60000 MNIST training digits, got %s / %s' % (digits.shape, digit_labels.shape))
    if (digits.min(), digits.max()) != (0, 255) or (digit_labels.min(), digit_labels.max()) != (0, 9):
        raise ValueError('create_mnistrgb: not MNIST (value ranges)')
    place_value = np.array([1, 10, 100])                                                     # R digit + 10 G digit + 100 B digit (:326)
    numbers = np.empty(num_images, dtype=np.int64)
    with TFRecordExporter(tfrecord_dir, num_images) as tfr:
        rnd = np.random.RandomState(random_seed)
        for i in range(num_images):
            pick = rnd.randint(digits.shape[0], size=3)                                      # one draw of three indices per image (:324)
            tfr.add_image(digits[pick])
            numbers[i] = int(np.dot(digit_labels[pick].astype(np.int64), place_value))
        if numbers.min() != 0 or numbers.max() != 999:                                       # the reference asserts this (:329): the one-hot width is max + 1
            raise ValueError('create_mnistrgb: %d images do not span the numbers 0..999 (got %d..%d); use more images' % (num_images, numbers.min(), numbers.max()))
        onehot = np.zeros((num_images, 1000), dtype=np.float32)
        onehot[np.arange(num_images), numbers] = 1.0
        tfr.add_labels(onehot)


def read_celeba_attributes(attr_file):
    """list_attr_celeba.txt: a count line, a header line with the 40 names, then '<file>.jpg  v1 ... v40' with v in {-1, 1};
    -> {file name: [0 / 1] * 40} (the reference replaces the text '-1' by '0' in the whole line, :474)."""
    table = {}
    with open(attr_file) as f:
        for line in f.readlines()[2:]:
            fields = line.replace('-1', '0').split()
            if fields:
                table[fields[0]] = [int(v) for v in fields[1:]]
    return table


def create_celeba(tfrecord_dir, celeba_dir, cx=89, cy=121, shuffle=0, num_images=0, num_shifts=0, export_attr=1, attr_file=CELEBA_ATTR_FILE):
    import PIL.Image
    print('Loading CelebA from "%s"' % celeba_dir)
    files = sorted(glob.glob(os.path.join(celeba_dir, '*.png')))
    if num_images != 0:                                                                      # :455-460: a head of the list, optionally with the last num_shifts files in place of its tail
        files = files[:num_images] if num_shifts == 0 else files[:num_images - num_shifts] + files[-num_shifts:]
    with TFRecordExporter(tfrecord_dir, len(files)) as tfr:
        order = tfr.choose_shuffled_order() if shuffle else np.arange(len(files))
        for i in order:
            img = np.asarray(PIL.Image.open(files[i]))
            if img.shape != (218, 178, 3):
                raise ValueError('create_celeba: %s is %s, expected the aligned 218x178 RGB images' % (files[i], img.shape))
            tfr.add_image(img[cy - 64:cy + 64, cx - 64:cx + 64].transpose(2, 0, 1))          # 128x128 crop, HWC -> CHW (:467-468)
        if export_attr:
            if not os.path.isfile(attr_file):
                raise FileNotFoundError('create_celeba: attribute file %s not found (export_attr=1)' % attr_file)
            table = read_celeba_attributes(attr_file)
            labels = np.array([table[os.path.basename(f).replace('png', 'jpg')] for f in files]).astype(np.float32)   # listed by .jpg name (:483)
            tfr.add_labels(labels[order])


def execute_cmdline(argv):
    parser = argparse.ArgumentParser(prog=argv[0], description='Dataset builders of the InclusiveGAN configs (TFRecord directories).')
    sub = parser.add_subparsers(dest='command')
    p = sub.add_parser('create_mnistrgb', help='Stacked MNIST (dataset_tool.py:307-334)')
    p.add_argument('tfrecord_dir'); p.add_argument('mnist_dir')
    p.add_argument('--num_images', type=int, default=1000000); p.add_argument('--random_seed', type=int, default=123)
    p = sub.add_parser('create_celeba', help='CelebA 128x128 with attributes (dataset_tool.py:447-486)')
    p.add_argument('tfrecord_dir'); p.add_argument('celeba_dir')
    for name, default in (('cx', 89), ('cy', 121), ('shuffle', 0), ('num_images', 0), ('num_shifts', 0), ('export_attr', 1)):
        p.add_argument('--' + name, type=int, default=default)
    args = parser.parse_args(argv[1:])
    if args.command is None:
        parser.print_help()
        return 1
    kw = vars(args)
    globals()[kw.pop('command')](**kw)
    return 0


if __name__ == '__main__':
    sys.exit(execute_cmdline(sys.argv))
