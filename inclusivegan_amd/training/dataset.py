"""Training-set objects with the interface `training_loop` uses from the reference's
`training/dataset.py` (TFRecordDataset :19-176: shape, dtype, dynamic_range, label_size,
resolution_log2, configure, get_minibatch_tf/np, get_random_labels_tf/np, close; load_dataset :181).

Two sources: `SyntheticDataset` (there is no network for datasets, and the headline metric is defined on synthetic
CelebA- / Stacked-MNIST-shaped batches) and `TFRecordDataset`, which reads the record files + label file that the
reference's dataset_tool.py writes (format: training/tfrecord.py), so a data set prepared for the reference trains here as is.

Images are uint8 [N, C, H, W] ~ U{0..255} (seeded), labels fp32 [N, label_size]: one-hot for
Stacked-MNIST-like sets (dataset_tool.py:332-334) or Bernoulli(0.2) {0,1} attributes for
CelebA-like sets (dataset_tool.py:467-486).  Like the reference with shuffle_mb=0
(training_loop.py:169-170) minibatches walk the set in order and wrap around; get_minibatch_np and
get_minibatch_tf share ONE iterator (dataset.py:148-157), which restarts whenever configure() changes
the minibatch size (:139-145).
"""
import numpy as np
import torch


class SyntheticDataset:
    def __init__(self, resolution=128, num_channels=3, label_size=40, data_size=30000, label_kind='attributes',
                 seed=0, device=None, rank=0, world_size=1, **_ignored):
        self.resolution = int(resolution)
        self.resolution_log2 = int(np.log2(self.resolution))
        assert self.resolution == 2 ** self.resolution_log2
        self.shape = [int(num_channels), self.resolution, self.resolution]
        self.dtype = 'uint8'
        self.dynamic_range = [0, 255]
        self.label_size = int(label_size)
        self.label_dtype = 'float32'
        self.data_size = int(data_size)
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.rank, self.world_size = int(rank), int(world_size)
        rng = np.random.RandomState(seed)
        self._images = rng.randint(0, 256, size=[self.data_size] + self.shape, dtype=np.uint8)
        if self.label_size == 0:
            self._labels = np.zeros([self.data_size, 0], dtype=np.float32)
        elif label_kind == 'onehot':
            idx = rng.randint(0, self.label_size, size=self.data_size)
            self._labels = np.zeros([self.data_size, self.label_size], dtype=np.float32)
            self._labels[np.arange(self.data_size), idx] = 1.0
        else:
            self._labels = (rng.rand(self.data_size, self.label_size) < 0.2).astype(np.float32)
        self._cur_minibatch = -1
        self._cur_lod = -1
        self._cursor = 0
        self._dev_images = None
        self._dev_labels = None

    def close(self):
        pass

    def configure(self, minibatch_size, lod=0):
        """dataset.py:139-145: a CHANGE of minibatch size (or lod) re-initialises the iterator, i.e. the walk restarts at
        image 0; configuring the same values again does nothing."""
        lod = int(np.floor(lod))
        assert minibatch_size >= 1 and lod == 0
        if self._cur_minibatch != int(minibatch_size) or self._cur_lod != lod:
            self._cursor = 0
            self._cur_minibatch = int(minibatch_size)
            self._cur_lod = lod

    def _next_indices(self, n):
        idx = (self._cursor + np.arange(n)) % self.data_size
        self._cursor = int((self._cursor + n) % self.data_size)
        return idx

    def get_minibatch_np(self, minibatch_size, lod=0):  # => images, labels
        self.configure(minibatch_size, lod)
        idx = self._next_indices(minibatch_size)
        return self._images[idx], self._labels[idx]

    def get_minibatch_tf(self):  # => images (uint8, device), labels -- this rank's slice of the global minibatch
        assert self._cur_minibatch > 0
        start = self._cursor
        self._next_indices(self._cur_minibatch)
        per = self._cur_minibatch // self.world_size
        if self._dev_images is None:
            self._dev_images = torch.from_numpy(self._images).to(self.device)
            self._dev_labels = torch.from_numpy(self._labels).to(self.device)
            self._dev_arange = torch.arange(self._cur_minibatch, device=self.device)
        if self._dev_arange.shape[0] < per:
            self._dev_arange = torch.arange(per, device=self.device)
        # indices are formed ON the device from host scalars: no host->device copy, hence no stream synchronisation
        tidx = (self._dev_arange[:per] + (start + self.rank * per)) % self.data_size
        return self._dev_images[tidx], self._dev_labels[tidx]

    def get_random_labels_np(self, minibatch_size, rng=None):  # => labels
        """`rng` None: the GLOBAL NumPy stream, like dataset.py:163-166 (the host loop's draws depend on it).  The metrics pass
        their own generator: the reference draws THEIR labels with TensorFlow ops (dataset.py:159-161), which leaves the NumPy
        stream the training loop reads untouched."""
        if self.label_size > 0:
            return self._labels[(rng or np.random).randint(self._labels.shape[0], size=[minibatch_size])]
        return np.zeros([minibatch_size, 0], np.float32)

    def get_random_labels_tf(self, minibatch_size):  # => labels
        # G and D ignore labels (conditioning is commented out, networks_stylegan2.py:278-284,494-500);
        # a device-side zero block avoids a host->device copy per call.
        return torch.zeros([minibatch_size, self.label_size], device=self.device, dtype=torch.float32)


class TFRecordDataset(SyntheticDataset):
    """The reference's data set object (training/dataset.py:19-176) on the record files `dataset_tool.py` writes, read
    without TensorFlow (training/tfrecord.py): the full-resolution file `*-rNN.tfrecords` with the largest NN, labels from
    `*.labels` (np.load; `max_label_size` = 0 / 'full' / N first components, :83-93), `max_images`, everything resident in
    host memory (CelebA-128 30k = 1.4 GB) and mirrored to the device on first use.  Iterator semantics are inherited:
    sequential walk with wrap-around (`shuffle_mb=0`, the only mode the training loop uses, training_loop.py:169-170), one
    iterator behind get_minibatch_np / get_minibatch_tf, restart on configure() with a new size.  Lower levels of detail
    are progressive-growing inputs (configs a-d) and are not read."""

    def __init__(self, tfrecord_dir, resolution=None, label_file=None, max_label_size=0, max_images=None, repeat=True,
                 shuffle_mb=0, prefetch_mb=2048, buffer_mb=256, num_threads=2, device=None, rank=0, world_size=1, verify=False, **_ignored):
        import glob
        import os
        from . import tfrecord
        if shuffle_mb:
            raise NotImplementedError('TFRecordDataset: shuffling is not built (the training loop runs with shuffle_mb=0)')
        assert os.path.isdir(tfrecord_dir), tfrecord_dir
        tfr_files = sorted(glob.glob(os.path.join(tfrecord_dir, '*.tfrecords')))
        assert len(tfr_files) >= 1
        shapes = [tfrecord.parse_example(next(tfrecord.read_records(f, limit=1))).shape for f in tfr_files]
        best = max(range(len(tfr_files)), key=lambda i: int(np.prod(shapes[i])))
        max_shape = shapes[best]
        self.tfrecord_dir = tfrecord_dir
        self.resolution = int(resolution) if resolution is not None else int(max_shape[1])
        assert self.resolution == max_shape[1] == max_shape[2], 'only the full-resolution level of detail is read'
        self.resolution_log2 = int(np.log2(self.resolution))
        self.shape = [int(max_shape[0]), self.resolution, self.resolution]
        self.dtype = 'uint8'
        self.dynamic_range = [0, 255]
        images = [tfrecord.parse_example(r) for r in tfrecord.read_records(tfr_files[best], verify=verify, limit=max_images)]
        self._images = np.stack(images)
        self.label_file = label_file
        if self.label_file is None:
            guess = sorted(glob.glob(os.path.join(tfrecord_dir, '*.labels')))
            if len(guess):
                self.label_file = guess[0]
        elif not os.path.isfile(self.label_file):
            guess = os.path.join(tfrecord_dir, self.label_file)
            if os.path.isfile(guess):
                self.label_file = guess
        assert max_label_size == 'full' or max_label_size >= 0
        labels = np.zeros([self._images.shape[0], 0], dtype=np.float32)
        if self.label_file is not None and max_label_size != 0:
            labels = np.load(self.label_file)
            assert labels.ndim == 2
        if max_label_size != 'full' and labels.shape[1] > max_label_size:
            labels = labels[:, :max_label_size]
        if max_images is not None and labels.shape[0] > max_images:
            labels = labels[:max_images]
        self._labels = np.ascontiguousarray(labels[:self._images.shape[0]], dtype=np.float32) if labels.shape[1] else \
            np.zeros([self._images.shape[0], 0], dtype=np.float32)
        self.label_size = int(self._labels.shape[1])
        self.label_dtype = 'float32'
        self.data_size = int(self._images.shape[0])
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.rank, self.world_size = int(rank), int(world_size)
        self._cur_minibatch = -1
        self._cur_lod = -1
        self._cursor = 0
        self._dev_images = None
        self._dev_labels = None


def load_dataset(class_name=None, data_dir=None, verbose=False, **kwargs):
    """training/dataset.py:181-197.  `tfrecord_dir` (joined with data_dir) that exists on disk -> TFRecordDataset; otherwise a
    synthetic set of the given `resolution` / `num_channels` / `label_size` (there is no network here for real data: the
    headline metric is defined on synthetic CelebA- / Stacked-MNIST-shaped batches)."""
    import os
    kwargs = dict(kwargs)
    tfr = kwargs.get('tfrecord_dir')
    if tfr is not None:
        path = os.path.join(data_dir, tfr) if data_dir is not None else tfr
        if os.path.isdir(path):
            kwargs['tfrecord_dir'] = path
            kwargs.pop('data_size', None)
            for k in ('num_channels', 'label_kind', 'seed'):
                kwargs.pop(k, None)
            if 'label_size' in kwargs:
                kwargs.pop('label_size')
            ds = TFRecordDataset(**kwargs)
            if verbose:
                print('Streaming data using %s...' % (class_name or 'training.dataset.TFRecordDataset'))
                print('Dataset shape =', np.int32(ds.shape).tolist())
                print('Dynamic range =', ds.dynamic_range)
                print('Label size    =', ds.label_size)
            return ds
        if 'resolution' not in kwargs:
            raise FileNotFoundError('dataset directory %s does not exist (and no synthetic shape was given)' % path)
    kwargs.pop('tfrecord_dir', None)
    kwargs.pop('max_label_size', None)
    kwargs.pop('shuffle_mb', None)
    ds = SyntheticDataset(**kwargs)
    if verbose:
        print('Synthetic dataset shape =', [ds.data_size] + ds.shape, ' label_size =', ds.label_size)
    return ds
