"""Golden vectors for the HOST side of the IMLE term, produced by EXECUTING the reference's own statements.

`training/training_loop.py` cannot be imported here (it imports TensorFlow at module level), but the part of its
main loop that decides which (real, label, latent) triples feed each iteration -- lines 325-482: candidate latents,
refresh cadence, label draw, nearest-neighbour bookkeeping, distance threshold / attribute mask, carry-over, slerp
perturbation, the two shuffles -- is plain NumPy.  This script parses the reference file where it lies
(/root/reference, read at generation time only; nothing of it is stored in the repo), cuts exactly those statements
out of `training_loop()`'s syntax tree (plus the module-level `training_schedule`), and executes them unchanged in a
namespace whose TensorFlow-side objects are small recording stand-ins:

    real (reference code)   the statements of :325-482 themselves, `training_schedule` (:65-118), `training.misc`
                            (slerp, adjust_dynamic_range), `dnnlib.EasyDict`, NumPy's global random stream
    stand-ins               G.run (a fixed random linear map + tanh of the latents), the DCI index (exact fp64 search,
                            same return structure as dci.py:316-330), the two data sets (in-memory arrays with the
                            iterator semantics of dataset.py:139-166), tflib.run (records its feed_dict), RunContext

Output: tests/golden/imle_host_golden.npz -- per case and iteration the fed arrays (as data-set indices for the
reals, full arrays for the latents), the refresh iterations, the nearest indices / distances of every refresh.
tests/test_imle_host.py replays the same cases through inclusivegan_amd.training.imle.ImleSampler and through the
restatement oracle/training_loop.py and requires identical results.

Run from the repo root:  python tests/golden/make_imle_golden.py   (needs /root/reference)
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from tests.imle_cases import CASES, FakeDataset, fake_generator, exact_knn   # noqa: E402  (shared with the test: inputs only)


def reference_loop_code():
    """(code object of :325-482 with the per-tick maintenance cut off, code object of training_schedule)."""
    path = os.path.join(REF, 'training', 'training_loop.py')
    mod = ast.parse(open(path).read(), filename=path)
    sched = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name == 'training_schedule'][0]
    tl = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name == 'training_loop'][0]
    loop = [n for n in tl.body if isinstance(n, ast.While)][0]
    first = [i for i, n in enumerate(tl.body) if isinstance(n, ast.Assign) and getattr(n.targets[0], 'id', '') == 'latent_candidates'][0]
    pre = tl.body[first:tl.body.index(loop)]
    # while body: everything up to and including the `for _repeat` loop; the tick maintenance (`done = ...; if ...`) goes
    k = [i for i, n in enumerate(loop.body) if isinstance(n, ast.For)][0]
    cut = ast.While(test=loop.test, body=loop.body[:k + 1], orelse=[])
    block = ast.Module(body=pre + [cut], type_ignores=[])
    ast.fix_missing_locations(block)
    return compile(block, path, 'exec'), compile(ast.Module(body=[sched], type_ignores=[]), path, 'exec')


class Recorder:
    def __init__(self):
        self.iterations = []
        self.refresh = []

    def run(self, ops, feed):       # tflib.run(...)
        if isinstance(ops, list) and ops and ops[0] == 'G_train_op':
            self.iterations.append({k: np.array(v) for k, v in feed.items() if isinstance(k, str) and '_rec_' in k})


def run_case(case):
    sys.path.insert(0, REF)
    import dnnlib
    from training import misc as ref_misc
    loop_code, sched_code = reference_loop_code()
    np.random.seed(case['seed'])
    ts = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    ts_rec = FakeDataset(case, np.random.RandomState(case['seed'] + 1))
    rec = Recorder()

    class FakeG:
        input_shapes = [[None, case['latent_dim']], [None, case['label_size']]]

        @staticmethod
        def run(latents, labels, is_validation, minibatch_size, num_gpus):
            assert is_validation
            return fake_generator(case, latents)

    class FakeDCI:
        def reset(self):
            self.data = None

        def add(self, data, num_levels, field_of_view, prop_to_retrieve):
            assert data.dtype == np.float64
            self.data = np.array(data)

        def query(self, q, num_neighbours, field_of_view, prop_to_retrieve):
            idx, dist = exact_knn(self.data, q, num_neighbours)
            rec.refresh_rows.append((idx[:, 0].copy(), dist[:, 0].copy()))
            return [i for i in idx], [d for d in dist]     # list of int32 arrays, list of float64 arrays (dci.py:316-330)

    rec.refresh_rows = []

    class Ctx:
        @staticmethod
        def get():
            return Ctx

        @staticmethod
        def should_stop():
            return False

    class FakeOpt:
        def reset_optimizer_state(self):
            pass

    ns = dict(np=np, misc=ref_misc, print=lambda *a, **k: None, tflib=rec,
              dnnlib=type('D', (), dict(RunContext=Ctx, EasyDict=dnnlib.EasyDict)),
              G=FakeG, dci_db=FakeDCI(), training_set=ts, training_set_rec=ts_rec,
              data_size=case['data_size'], num_samples_factor=case['num_samples_factor'], init_staleness=case['init_staleness'],
              candidate_batch_size=case['candidate_batch_size'], init_proj_dim=None, proj_dim=case['dim'],
              exclusive_retrieved_code=case.get('exclusive_retrieved_code', 0), dist_thres_percentile=case['dist_thres_percentile'],
              attr_interesting=case['attr_interesting'], attr_names=case['attr_names'], knn_perturb_factor=case['knn_perturb_factor'],
              minibatch_repeats=case['minibatch_repeats'], lazy_regularization=True, G_reg_interval=4, D_reg_interval=16,
              num_gpus=1, sched_args=dict(minibatch_size_base=case['mb'], minibatch_gpu_base=case['mb']),
              reset_opt_for_new_lod=True, prev_lod=-1.0, G_opt=FakeOpt(), D_opt=FakeOpt(), drange_net=[-1, 1],
              cur_nimg=0, total_kimg=case['total_img'] / 1000.0, running_mb_counter=0, cursor=0,
              lod_in='lod_in', lrate_in='lrate_in', minibatch_size_in='minibatch_size_in', minibatch_gpu_in='minibatch_gpu_in',
              reals_rec_1='reals_rec_1', labels_rec_1='labels_rec_1', latents_rec_1='latents_rec_1',
              reals_rec_2='reals_rec_2', labels_rec_2='labels_rec_2', latents_rec_2='latents_rec_2',
              G_train_op='G_train_op', G_loss='G_loss', G_reg_op='G_reg_op', D_train_op='D_train_op', D_loss='D_loss',
              Gs_update_op='Gs_update_op', D_reg_op='D_reg_op')
    exec(sched_code, ns)
    exec(loop_code, ns)
    out = {}
    out['num_iterations'] = np.int64(len(rec.iterations))
    for key in ('reals_rec_1', 'reals_rec_2'):
        out[key + '_idx'] = np.stack([FakeDataset.decode_indices(it[key]) for it in rec.iterations])
    for key in ('labels_rec_1', 'labels_rec_2', 'latents_rec_1', 'latents_rec_2'):
        out[key] = np.stack([it[key] for it in rec.iterations])
    # one refresh = data_size / (2 mb) query calls
    per = case['data_size'] // (2 * case['mb'])
    rows = rec.refresh_rows
    assert len(rows) % per == 0
    out['nearest_indices'] = np.stack([np.concatenate([r[0] for r in rows[i:i + per]]) for i in range(0, len(rows), per)])
    out['nearest_dists'] = np.stack([np.concatenate([r[1] for r in rows[i:i + per]]) for i in range(0, len(rows), per)])
    out['final_cursor'] = np.int64(ns['cursor'])
    out['final_staleness'] = np.int64(ns['init_staleness'])
    out['final_rec_cursor'] = np.int64(ts_rec.cursor)
    return out


def main():
    res = {}
    for name, case in CASES.items():
        for k, v in run_case(case).items():
            res['%s/%s' % (name, k)] = v
        print(name, 'iterations', int(res[name + '/num_iterations']), 'refreshes', res[name + '/nearest_indices'].shape[0])
    np.savez_compressed(os.path.join(HERE, 'imle_host_golden.npz'), **res)


if __name__ == '__main__':
    os.chdir(REPO)
    main()
