"""Host side of the IMLE term: which (real, label, matched latent) triples feed each iteration.

This is the NumPy bookkeeping of the reference's main loop, `training/training_loop.py:325-464`, kept
bit-exact -- every index, every permutation and every draw from the global NumPy stream equals the
reference's for the same seed (tests/test_imle_host.py drives this class and a literal restatement of
those lines, oracle/training_loop.py, side by side):

  * refresh cadence (:354-356): at the first iteration and whenever `cur_nimg` crosses a multiple of
    `data_size * staleness`; the staleness doubles at every refresh but the first;
  * a refresh draws `get_random_labels_np(data_size * num_samples_factor)` from the global stream (:357),
    runs the nearest-neighbour assignment (`search`: the device work, see training_loop.imle_refresh) which
    walks `training_set_rec` once in `2 * minibatch` steps (:374-403), and advances `cursor` by `data_size`;
  * `dist_thres = percentile(selected_dists, dist_thres_percentile)` (:406);
  * per iteration (:409-441): take the next `2 * minibatch` reals in dataset order with their matched
    latents, keep those within the distance threshold -- or, with `attr_interesting`, those whose listed
    attributes are all 1 (:416-424) -- and carry the surplus over to the next iteration; the carried-over
    rows are dropped when a new pass over the data set begins (`cursor % data_size == 0`);
  * `slerp` the latents towards fresh N(0, I) noise by `knn_perturb_factor` (:447), split in halves, permute
    the first half by a shuffled `order` and the second by the SAME array shuffled once more (:456-464).

Nothing here touches the device.
"""
import numpy as np

from . import misc


# Column order of CelebA's 40 binary attributes (header line of Anno/list_attr_celeba.txt, the order dataset_tool.py:467-486
# writes the label file in): the vocabulary `attr_interesting` names are looked up in when the annotation file itself is not
# at hand (synthetic CelebA-shaped data).
CELEBA_ATTRIBUTES = (
    '5_o_Clock_Shadow Arched_Eyebrows Attractive Bags_Under_Eyes Bald Bangs Big_Lips Big_Nose Black_Hair Blond_Hair Blurry '
    'Brown_Hair Bushy_Eyebrows Chubby Double_Chin Eyeglasses Goatee Gray_Hair Heavy_Makeup High_Cheekbones Male '
    'Mouth_Slightly_Open Mustache Narrow_Eyes No_Beard Oval_Face Pale_Skin Pointy_Nose Receding_Hairline Rosy_Cheeks Sideburns '
    'Smiling Straight_Hair Wavy_Hair Wearing_Earrings Wearing_Hat Wearing_Lipstick Wearing_Necklace Wearing_Necktie Young').split()


def attribute_names(attr_file):
    """The attribute vocabulary of CelebA: second line of list_attr_celeba.txt (training_loop.py:174-180)."""
    with open(attr_file) as f:
        lines = f.readlines()
    return lines[1].split()


class ImleSampler:
    def __init__(self, training_set_rec, latent_candidates, data_size, num_samples_factor, init_staleness,
                 knn_perturb_factor, dist_thres_percentile=100.0, attr_interesting=None, attr_names=None, search=None):
        self.training_set_rec = training_set_rec
        self.latent_candidates = latent_candidates
        self.data_size = int(data_size)
        self.num_samples_factor = int(num_samples_factor)
        self.staleness = init_staleness
        self.knn_perturb_factor = knn_perturb_factor
        self.dist_thres_percentile = dist_thres_percentile
        self.attr_idx = None
        if attr_interesting is not None:
            if attr_names is None:
                raise ValueError('attr_interesting needs the attribute names (celeba/Anno/list_attr_celeba.txt, training_loop.py:174-180)')
            self.attr_idx = [attr_names.index(a) for a in attr_interesting.split(',')]      # ValueError on an unknown name, like :421
        self.search = search
        self.cursor = 0
        self.selected_latents = None
        self.selected_dists = None
        self.nearest_indices = None
        self.dist_thres = None
        self.remained = None            # (reals, labels, latents) rows carried over (:328-330)
        self.tick = None                # first batch of the current pass, kept for the snapshot grids (:449-453)
        self._beginning = False
        self.num_refreshes = 0

    # ------------------------------------------------------------------
    def refresh_due(self, cur_nimg, minibatch_size):
        period = self.data_size * self.staleness
        return self.selected_latents is None or cur_nimg // period != (cur_nimg - minibatch_size * 2) // period    # :354

    def refresh(self, minibatch_size):
        if self.selected_latents is not None:
            self.staleness *= 2                                                                       # :355-356
        num_cand = self.data_size * self.num_samples_factor
        label_candidates = self.training_set_rec.get_random_labels_np(num_cand)                       # :357
        nearest_indices, dists = self.search(self.latent_candidates, label_candidates, minibatch_size)   # :358-402
        nearest_indices = np.asarray(nearest_indices)
        assert nearest_indices.shape == (self.data_size,)
        self.cursor += self.data_size                                                                 # :403, once per query batch
        self.nearest_indices = nearest_indices
        self.selected_latents = self.latent_candidates[nearest_indices]                               # :404
        self.selected_dists = np.asarray(dists, dtype=np.float64)                                     # :405
        self.dist_thres = np.percentile(self.selected_dists, self.dist_thres_percentile)              # :406
        self.num_refreshes += 1

    # ------------------------------------------------------------------
    def next_batch(self, minibatch_size):
        """-> dict(reals_rec_1/2 float32 [mb,C,H,W] in the data range, labels_rec_1/2, latents_rec_1/2, order_1, order_2)."""
        mb2 = minibatch_size * 2
        ds = self.data_size
        fresh_pass = self.cursor % ds == 0
        cur = None if (self.remained is None or fresh_pass) else [np.array(a) for a in self.remained]   # :409-411
        while cur is None or cur[0].shape[0] < mb2:
            reals_t, labels_t = self.training_set_rec.get_minibatch_np(mb2)
            reals_t = reals_t.astype(np.float32)
            pos = self.cursor % ds
            latents_t = self.selected_latents[pos:pos + mb2]
            if self.attr_idx is None:
                keep = self.selected_dists[pos:pos + mb2] <= self.dist_thres                          # :417
            else:
                active = np.ones(labels_t.shape[0])
                for i in self.attr_idx:
                    active *= labels_t[:, i]
                keep = active == 1                                                                    # :419-424
            sel = [reals_t[keep], labels_t[keep], latents_t[keep]]
            if cur is None or self.cursor % ds == 0:
                cur = [np.array(a) for a in sel]                                                      # restart (drops what was carried)
            else:
                cur = [np.concatenate((a, b), axis=0) for a, b in zip(cur, sel)]
            if cur[0].shape[0] > mb2:
                self.remained = [np.array(a[mb2:]) for a in cur]
                cur = [np.array(a[:mb2]) for a in cur]
            else:
                self.remained = None
            if self.cursor % ds == 0:
                self._beginning = True
            self.cursor += mb2

        reals, labels, latents = cur
        noise = np.random.randn(*latents.shape).astype(np.float32)
        latents = misc.slerp(latents, noise, self.knn_perturb_factor)                                 # :447
        if self._beginning:
            self.tick = (np.array(reals), np.array(labels), np.array(latents))
            self._beginning = False
        mb = minibatch_size
        order = np.arange(mb)
        np.random.shuffle(order)                                                                      # :456-457
        order_1 = order.copy()
        out = dict(reals_rec_1=reals[:mb][order], labels_rec_1=labels[:mb][order], latents_rec_1=latents[:mb][order])
        np.random.shuffle(order)                                                                      # :461, the same array again
        out.update(reals_rec_2=reals[mb:][order], labels_rec_2=labels[mb:][order], latents_rec_2=latents[mb:][order],
                   order_1=order_1, order_2=order.copy())
        return out
