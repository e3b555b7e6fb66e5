"""CPU: the host side of the IMLE term (which reals / labels / latents feed each iteration, in which order) is
bit-identical to the reference.

  golden   tests/golden/imle_host_golden.npz = the reference's OWN statements of training/training_loop.py:325-482,
           cut out of the reference file and executed verbatim on the inputs of tests/imle_cases.py
           (tests/golden/make_imle_golden.py; TensorFlow-side objects replaced by recording stand-ins)
  oracle   oracle/training_loop.py, the NumPy restatement of the same lines
  product  inclusivegan_amd.training.imle.ImleSampler + the loop skeleton of inclusivegan_amd.training.training_loop

All three must agree exactly: data-set indices of the fed reals, labels, perturbed latents (bit for bit), refresh
iterations, nearest-neighbour tables, final cursor / staleness, and the position of the data-set iterator.
"""
import os

import numpy as np
import pytest

from tests.imle_cases import CASES, FakeDataset, fake_generator, exact_knn

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'imle_host_golden.npz'))


class _Index:
    def __init__(self, data):
        self.data = data

    def query(self, q, k):
        return exact_knn(self.data, q, k)


def _run_oracle(case):
    from oracle.training_loop import imle_host_loop
    np.random.seed(case['seed'])
    ts = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    ts_rec = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    feeds = []
    log = imle_host_loop(ts, ts_rec, [case['latent_dim']], lambda z, l: fake_generator(case, z), _Index, feeds.append,
                         data_size=case['data_size'], num_samples_factor=case['num_samples_factor'], init_staleness=case['init_staleness'],
                         candidate_batch_size=case['candidate_batch_size'], minibatch_size=case['mb'], minibatch_repeats=case['minibatch_repeats'],
                         total_img=case['total_img'], knn_perturb_factor=case['knn_perturb_factor'],
                         dist_thres_percentile=case['dist_thres_percentile'], attr_interesting=case['attr_interesting'], attr_names=case['attr_names'])
    log['rec_cursor'] = ts_rec.cursor
    return feeds, log


def _run_product(case):
    """The product's sampler inside the skeleton of its training loop (configure calls, refresh test, counters);
    `search` consumes training_set_rec exactly like training_loop.imle_refresh (2 * minibatch per pull)."""
    from inclusivegan_amd.training.imle import ImleSampler
    from inclusivegan_amd.training import misc
    np.random.seed(case['seed'])
    ts = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    ts_rec = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    ds, mb, nsf = case['data_size'], case['mb'], case['num_samples_factor']
    tables = []

    def search(latents, label_candidates, minibatch_size):
        cands = np.concatenate([fake_generator(case, latents[i:i + case['candidate_batch_size']]).reshape(-1, case['dim'])
                                for i in range(0, latents.shape[0], case['candidate_batch_size'])]).astype(np.float64)
        idx, dist = [], []
        for _ in range(ds // (2 * minibatch_size)):
            r, _ = ts_rec.get_minibatch_np(2 * minibatch_size)
            q = misc.adjust_dynamic_range(r.astype(np.float32), ts_rec.dynamic_range, [-1, 1]).reshape(r.shape[0], -1)
            i, d = exact_knn(cands, q, 1)
            idx.append(i[:, 0]); dist.append(d[:, 0])
        tables.append((np.concatenate(idx), np.concatenate(dist)))
        return tables[-1]

    latent_candidates = np.random.randn(ds * nsf, case['latent_dim']).astype(np.float32)
    sampler = ImleSampler(ts_rec, latent_candidates, ds, nsf, case['init_staleness'], case['knn_perturb_factor'],
                          dist_thres_percentile=case['dist_thres_percentile'], attr_interesting=case['attr_interesting'],
                          attr_names=case['attr_names'], search=search)
    feeds, refresh_at = [], []
    cur_nimg = 0
    while cur_nimg < case['total_img']:
        ts.configure(mb * 2, 0)
        ts_rec.configure(mb * 2, 0)
        for _ in range(case['minibatch_repeats']):
            if sampler.refresh_due(cur_nimg, mb):
                sampler.refresh(mb)
                refresh_at.append(cur_nimg)
            feeds.append(sampler.next_batch(mb))
            cur_nimg += mb * 2
    log = dict(refresh_at=refresh_at, nearest_indices=[t[0] for t in tables], nearest_dists=[t[1] for t in tables],
               final_cursor=sampler.cursor, final_staleness=sampler.staleness, rec_cursor=ts_rec.cursor)
    return feeds, log


def _check_against_golden(name, feeds, log):
    g = lambda k: GOLD['%s/%s' % (name, k)]
    assert len(feeds) == int(g('num_iterations'))
    for h in ('1', '2'):
        idx = np.stack([FakeDataset.decode_indices(f['reals_rec_' + h]) for f in feeds])
        assert np.array_equal(idx, g('reals_rec_%s_idx' % h)), 'reals of half ' + h
        assert np.array_equal(np.stack([f['labels_rec_' + h] for f in feeds]), g('labels_rec_' + h))
        lat = np.stack([f['latents_rec_' + h] for f in feeds])
        assert lat.dtype == np.float32 and np.array_equal(lat, g('latents_rec_' + h)), 'latents of half %s are not bit-identical' % h
    assert np.array_equal(np.stack(log['nearest_indices']), g('nearest_indices'))
    assert np.array_equal(np.stack(log['nearest_dists']), g('nearest_dists'))
    assert log['final_cursor'] == int(g('final_cursor')) and log['final_staleness'] == int(g('final_staleness'))
    assert log['rec_cursor'] == int(g('final_rec_cursor'))


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_restatement_reproduces_the_reference_statements(name):
    feeds, log = _run_oracle(CASES[name])
    _check_against_golden(name, feeds, log)


@pytest.mark.parametrize('name', sorted(CASES))
def test_product_sampler_reproduces_the_reference_statements(name):
    feeds, log = _run_product(CASES[name])
    _check_against_golden(name, feeds, log)


def test_second_half_is_permuted_by_the_reshuffled_order():
    """training_loop.py:456-464: `order` is shuffled once for the first half and the SAME array is shuffled again for
    the second half (not a fresh arange)."""
    case = CASES['default']
    feeds, _ = _run_product(case)
    # replay the stream of one iteration by hand: after the slerp noise draw, shuffle(order) twice on ONE array
    np.random.seed(case['seed'])
    np.random.randn(case['data_size'] * case['num_samples_factor'], case['latent_dim'])              # latent candidates (:325)
    np.random.randint(case['data_size'], size=[case['data_size'] * case['num_samples_factor']])      # candidate labels (:357)
    np.random.randn(2 * case['mb'], case['latent_dim'])                                              # slerp noise (:447)
    order = np.arange(case['mb']); np.random.shuffle(order); first = order.copy(); np.random.shuffle(order)
    assert np.array_equal(feeds[0]['order_1'], first) and np.array_equal(feeds[0]['order_2'], order)
    ofeeds, _ = _run_oracle(case)
    assert all(np.array_equal(p['order_1'], o['order_1']) and np.array_equal(p['order_2'], o['order_2']) for p, o in zip(feeds, ofeeds))


def test_selection_exercises_threshold_carry_over_and_mask():
    """The cases are not vacuous: the distance threshold drops rows, surplus rows are carried over, the attribute mask
    keeps only rows whose listed attributes are all 1, and a refresh happens more than once with doubling staleness."""
    feeds, log = _run_product(CASES['thres60'])
    assert len(log['refresh_at']) >= 3 and log['refresh_at'][0] == 0
    assert log['final_staleness'] == CASES['thres60']['init_staleness'] * 2 ** (len(log['refresh_at']) - 1)
    kept = np.mean(log['nearest_dists'][0] <= np.percentile(log['nearest_dists'][0], 60.0))
    assert 0.5 < kept < 0.7
    case = CASES['attr_and']
    feeds, _ = _run_product(case)
    cols = [case['attr_names'].index(a) for a in case['attr_interesting'].split(',')]
    for f in feeds:
        for h in ('1', '2'):
            assert (f['labels_rec_' + h][:, cols] == 1).all()
    ds = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    assert not (ds.labels[:, cols] == 1).all()


def test_refresh_cadence_of_the_product():
    """refresh_due() is the rule of training_loop.py:354 with the doubling of :355-356: checked by calling the product."""
    from inclusivegan_amd.training.imle import ImleSampler
    ds, mb = 96, 4
    s = ImleSampler(None, np.zeros((ds, 2), np.float32), ds, 1, 2, 0.05, search=None)
    assert s.refresh_due(0, mb)                      # first iteration
    s.selected_latents = np.zeros((ds, 2)); s.staleness = 2
    due = [n for n in range(0, 2000, 2 * mb) if s.refresh_due(n, mb)]
    assert due == [0] + [n for n in range(2 * mb, 2000, 2 * mb) if n // 192 != (n - 8) // 192]
    assert due[1:4] == [192, 384, 576]


def test_unknown_attribute_is_an_error():
    from inclusivegan_amd.training.imle import ImleSampler
    with pytest.raises(ValueError):
        ImleSampler(None, np.zeros((4, 2), np.float32), 4, 1, 1, 0.05, attr_interesting='Bald', attr_names=None)
    with pytest.raises(ValueError):
        ImleSampler(None, np.zeros((4, 2), np.float32), 4, 1, 1, 0.05, attr_interesting='Nope', attr_names=['Bald'])


def test_exclusive_assignment_matches_reference_statements():
    """The exclusive variant (training_loop.py:382-396): the oracle restatement and the product's greedy pick on the same k-NN
    table give the same assignment; every candidate is used at most once while unused ones remain among a real's k nearest."""
    from oracle.training_loop import imle_host_loop
    from inclusivegan_amd.training.training_loop import exclusive_assignment
    case = dict(CASES['default'], seed=1010, total_img=8)
    np.random.seed(case['seed'])
    ts = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    ts_rec = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    tables = []

    class Rec(_Index):
        def query(self, q, k):
            out = exact_knn(self.data, q, k)
            tables.append(out)
            return out

    log = imle_host_loop(ts, ts_rec, [case['latent_dim']], lambda z, l: fake_generator(case, z), Rec, lambda feed: None,
                         data_size=case['data_size'], num_samples_factor=case['num_samples_factor'], init_staleness=case['init_staleness'],
                         candidate_batch_size=case['candidate_batch_size'], minibatch_size=case['mb'], minibatch_repeats=1,
                         total_img=case['total_img'], knn_perturb_factor=0.05, exclusive_retrieved_code=1)
    knn_idx = np.concatenate([t[0] for t in tables]); knn_dist = np.concatenate([t[1] for t in tables])
    assert knn_idx.shape == (case['data_size'], case['num_samples_factor'])
    idx, dist = exclusive_assignment(knn_idx, knn_dist)
    assert np.array_equal(idx, log['nearest_indices'][0]) and np.array_equal(dist, log['nearest_dists'][0])
    assert len(set(idx.tolist())) > len(set(knn_idx[:, 0].tolist()))          # exclusivity spreads the reals over more candidates
