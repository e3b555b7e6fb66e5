"""Golden values of the metric statistics, produced by EXECUTING the reference's own statements.

metrics/frechet_inception_distance.py, mode_counts.py and KL.py import TensorFlow at module level and cannot be imported
here, but the statistics they compute on top of the network outputs are plain NumPy / SciPy statements inside
`_evaluate`.  This script parses the reference files where they lie (/root/reference; nothing is copied), cuts out
  FID   the mean / covariance statements (:44-45, :60-61) and the final four (:64-71: m, sqrtm, dist),
  modes the statement reporting len(np.unique(labels_all)) (:49),
  KL    the three statements building the two densities and the sum (:49-51)
and executes them on seeded inputs, with `self._report_result` recording the value.
Output: tests/golden/metrics_golden.npz (inputs + values).  Run: python tests/golden/make_metrics_golden.py"""
import ast
import os

import numpy as np
import scipy
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/metrics'


def evaluate_body(fname, cls):
    mod = ast.parse(open(os.path.join(REF, fname)).read())
    c = [n for n in mod.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    return [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == '_evaluate'][0].body


def run(stmts, ns):
    m = ast.Module(body=stmts, type_ignores=[])
    ast.fix_missing_locations(m)
    exec(compile(m, '<reference statements>', 'exec'), ns)


class Recorder:
    def __init__(self):
        self.values = []

    def _report_result(self, value, suffix='', fmt=''):
        self.values.append(value)


def targets(stmt):
    out = []
    for t in getattr(stmt, 'targets', []):
        out += [e.id for e in ast.walk(t) if isinstance(e, ast.Name)]
    return out


def main():
    rng = np.random.RandomState(42)
    out = {}
    # ---- FID
    body = evaluate_body('frechet_inception_distance.py', 'FID')
    flat = []
    for st in body:
        flat += [st] + [s for s in ast.walk(st) if isinstance(s, ast.stmt) and s is not st]
    stat_real = [s for s in flat if isinstance(s, ast.Assign) and targets(s) in (['mu_real'], ['sigma_real']) and 'activations' in ast.dump(s)]
    stat_fake = [s for s in body if isinstance(s, ast.Assign) and targets(s) in (['mu_fake'], ['sigma_fake'])]
    tail = body[-4:]        # m, (s, _), dist, report
    assert len(stat_real) == 2 and len(stat_fake) == 2 and targets(tail[0]) == ['m']
    for name, (n, f, shift) in dict(small=(300, 12, 0.3), wide=(400, 64, 0.05), same=(256, 16, 0.0)).items():
        real = (rng.randn(n, f) @ rng.randn(f, f) * 0.3).astype(np.float32)
        fake = real.copy() if name == 'same' else (rng.randn(n, f) @ rng.randn(f, f) * 0.3 + shift).astype(np.float32)
        rec = Recorder()
        ns = dict(np=np, scipy=scipy, self=rec, activations=real)
        run(stat_real, ns)
        ns['activations'] = fake
        run(stat_fake + tail, ns)
        out['fid/%s/real' % name] = real; out['fid/%s/fake' % name] = fake
        out['fid/%s/value' % name] = np.float64(rec.values[0])
        print('FID', name, rec.values[0])
    # ---- mode counts / KL
    mc = evaluate_body('mode_counts.py', 'mode_counts')[-1:]
    kl = evaluate_body('KL.py', 'KL')[-4:]
    assert targets(kl[0]) == ['density_fake']
    for name, labels in dict(spread=rng.randint(0, 1000, size=4000), collapsed=rng.choice([3, 17, 512], size=2000, p=[0.7, 0.2, 0.1]),
                             all=np.arange(1000).repeat(3)).items():
        labels_all = labels.astype(np.float32)
        rec = Recorder()
        run(mc, dict(np=np, self=rec, labels_all=labels_all))
        classifier = type('C', (), dict(output_shape=[None, 1000]))
        run(kl, dict(np=np, self=rec, labels_all=labels_all, classifier=classifier))
        out['cls/%s/labels' % name] = labels_all
        out['cls/%s/modes' % name] = np.int64(rec.values[0]); out['cls/%s/kl' % name] = np.float64(rec.values[1])
        print(name, rec.values)
    np.savez_compressed(os.path.join(HERE, 'metrics_golden.npz'), **out)


if __name__ == '__main__':
    main()
