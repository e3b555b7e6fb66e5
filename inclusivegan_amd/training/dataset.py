"""Training-set objects with the interface `training_loop` uses from the reference's
`training/dataset.py` (TFRecordDataset :19-176: shape, dtype, dynamic_range, label_size,
resolution_log2, configure, get_minibatch_tf/np, get_random_labels_tf/np, close; load_dataset :181).

Only the synthetic source is built in this round: there is no network for datasets, and the
headline metric is defined on synthetic CelebA- / Stacked-MNIST-shaped batches.
The TFRecord on-disk format is a "next" row.

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

    def get_random_labels_np(self, minibatch_size):  # => labels
        if self.label_size > 0:      # the GLOBAL NumPy stream, like dataset.py:163-166 (the host loop's draws depend on it)
            return self._labels[np.random.randint(self._labels.shape[0], size=[minibatch_size])]
        return np.zeros([minibatch_size, 0], np.float32)

    def get_random_labels_tf(self, minibatch_size):  # => labels
        # G and D ignore labels (conditioning is commented out, networks_stylegan2.py:278-284,494-500);
        # a device-side zero block avoids a host->device copy per call.
        return torch.zeros([minibatch_size, self.label_size], device=self.device, dtype=torch.float32)


def load_dataset(class_name=None, data_dir=None, verbose=False, **kwargs):
    kwargs = dict(kwargs)
    kwargs.pop('tfrecord_dir', None)
    kwargs.pop('max_label_size', None)
    kwargs.pop('shuffle_mb', None)
    ds = SyntheticDataset(**kwargs)
    if verbose:
        print('Synthetic dataset shape =', [ds.data_size] + ds.shape, ' label_size =', ds.label_size)
    return ds
