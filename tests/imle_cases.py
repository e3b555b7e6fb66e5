"""Inputs shared by tests/golden/make_imle_golden.py (which executes the reference's own loop statements on them) and
tests/test_imle_host.py (which replays them through the product and the oracle restatement): small in-memory data sets with
the iterator semantics of the reference's TFRecordDataset, a fixed stand-in generator, an exact fp64 k-NN."""
import numpy as np


def _case(**kw):
    base = dict(seed=1000, data_size=48, mb=4, num_samples_factor=3, init_staleness=1, candidate_batch_size=16,
                latent_dim=16, shape=[3, 4, 4], label_size=4, attr_names=['Bald', 'Eyeglasses', 'Male', 'Smiling'],
                attr_interesting=None, dist_thres_percentile=100.0, knn_perturb_factor=0.05, minibatch_repeats=4,
                total_img=256, label_p=0.6)
    base.update(kw)
    base['dim'] = int(np.prod(base['shape']))
    return base


CASES = {
    'default': _case(),
    'thres60': _case(dist_thres_percentile=60.0, seed=1001),
    'attr_and': _case(attr_interesting='Bald,Male', seed=1002),
    'attr_one_repeats1': _case(attr_interesting='Smiling', minibatch_repeats=1, seed=1003, init_staleness=2),
}


class FakeDataset:
    """uint8 images [data_size, C, H, W] whose first two bytes spell the image's index; {0,1} attribute labels.
    configure / get_minibatch_np / get_random_labels_np behave like training/dataset.py:139-166 (one iterator, restarted
    when the minibatch size changes; random labels from the GLOBAL NumPy stream)."""

    def __init__(self, case, rng):
        n = case['data_size']
        self.shape = list(case['shape'])
        self.dtype = 'uint8'
        self.dynamic_range = [0, 255]
        self.label_size = case['label_size']
        self.label_dtype = 'float32'
        self.resolution_log2 = int(np.log2(self.shape[1]))
        self.images = rng.randint(0, 256, size=[n] + self.shape).astype(np.uint8)
        flat = self.images.reshape(n, -1)
        flat[:, 0] = np.arange(n) & 255
        flat[:, 1] = np.arange(n) >> 8
        self.labels = (rng.rand(n, self.label_size) < case['label_p']).astype(np.float32)
        self.cursor = 0
        self._mb = -1

    @staticmethod
    def decode_indices(reals):
        flat = np.asarray(reals).reshape(reals.shape[0], -1)
        return (flat[:, 0] + 256 * flat[:, 1]).astype(np.int64)

    def configure(self, minibatch_size, lod=0):
        if self._mb != minibatch_size:
            self.cursor = 0
            self._mb = minibatch_size

    def get_minibatch_np(self, minibatch_size, lod=0):
        self.configure(minibatch_size, lod)
        idx = (self.cursor + np.arange(minibatch_size)) % self.images.shape[0]
        self.cursor = int((self.cursor + minibatch_size) % self.images.shape[0])
        return self.images[idx], self.labels[idx]

    def get_random_labels_np(self, minibatch_size):
        return self.labels[np.random.randint(self.labels.shape[0], size=[minibatch_size])]

    def close(self):
        pass


def fake_generator(case, latents):
    """Stand-in for G.run(latents, labels, is_validation=True): a fixed random linear map + tanh, NCHW float32 in [-1, 1]."""
    w = np.random.RandomState(case['seed'] + 7).randn(case['latent_dim'], case['dim']).astype(np.float32)
    x = np.tanh(np.asarray(latents, dtype=np.float32) @ w / np.float32(np.sqrt(case['latent_dim'])))
    return x.reshape([x.shape[0]] + list(case['shape'])).astype(np.float32)


def exact_knn(data, query, k):
    """fp64 brute force: (idx int32 [nq, k], Euclidean dist float64 [nq, k]), ascending, ties to the lower index."""
    data = np.asarray(data, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    idx = np.empty((query.shape[0], k), dtype=np.int32)
    dist = np.empty((query.shape[0], k), dtype=np.float64)
    for i in range(query.shape[0]):
        diff = data - query[i]
        d2 = np.einsum('ij,ij->i', diff, diff)
        o = np.argsort(d2, kind='stable')[:k]
        idx[i] = o
        dist[i] = np.sqrt(d2[o])
    return idx, dist
