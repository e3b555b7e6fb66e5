"""GPU: the two second-order steps at the BENCH configuration (CelebA-shaped 128x128, config-e-Gskip-Dresnet, fmap_base 8192, minibatch_gpu 6: path-length
batch 3, R1 on 12 reals) under every arithmetic form of the large 3x3 convolutions, against the fp64 oracle, from ONE state (VERDICT r05 next #1a;
reference: training/loss.py:55-89,107-111).

Forms (child processes, tests/reg_forms.py): IGAN_CONV_PLANES=0 every convolution on the exact fp32 matrix instruction, =1 three bf16 pieces (exact 24-bit
split, six products), =2 two fp16 pieces (the default: operands to <= 1 ulp of fp32, scales per pixel / per channel, three products).

The path-length step is evaluated at two moving averages: pl_mean = 0 (initialisation) and pl_mean = 0.9 x the batch's mean path length (a trained network's
regime: the penalty is a difference of nearly equal numbers and every error of the path-length VALUE reaches the gradients amplified ~10x).

Bar: per op, the default form's worst per-variable gradient deviation (relative L2 against fp64) <= 2 x the exact-fp32 form's, the same for the bf16 form,
and every form inside the suite's stated tolerance (5e-3 per variable, regulariser value 1e-3).  The table goes to stdout (`pytest -s` / `-rP`; committed
under profiles/)."""
import os

import pytest
import torch

from tests import reg_forms as RF

pytestmark = pytest.mark.gpu

FACTOR = 2.0
FORMS = [('fp32 (0)', '0'), ('bf16x3 (1)', '1'), ('fp16x2 (2)', '2')]


def test_regularisers_at_the_bench_configuration_per_form(cuda_device, tmp_path):
    state, names = RF.init_state(cuda_device, 128, 8192, 6, pl_fracs=(0.0, 0.9))
    torch.cuda.empty_cache()
    spath = str(tmp_path / 'state.npz')
    RF.save_state_dict(spath, state)
    hip = {}
    for label, form in FORMS:
        opath = str(tmp_path / ('out_%s.npz' % form))
        info = RF.run_child(spath, opath, dict(IGAN_CONV_PLANES=form))
        assert info['form'] == int(form), info
        hip[label] = RF.load_result(opath)
        os.remove(opath)
    ora = RF.oracle_ops_of_state(state, trainables=names)
    labels = [l for l, _ in FORMS]
    devs = {l: {op: RF.deviations(hip[l][op], ora[op]) for op in ora} for l in labels}
    print('128x128 config-e, minibatch_gpu 6 (path-length batch 3, R1 on 12 reals); mean path length %.5g; G_reg@0: pl_mean 0, G_reg@1: pl_mean = 0.9 x mean path length'
          % state['mean_path_length'])
    print('oracle seconds: %s' % {op: round(r['seconds'], 1) for op, r in ora.items()})
    print(RF.table(devs, labels))
    bad = []
    for op in ora:
        worst = {l: max(devs[l][op]['errs'].values()) for l in labels}
        for l in labels:
            if not worst[l] < 5e-3:
                bad.append('%s %s: worst per-variable deviation %.2e exceeds the suite tolerance 5e-3' % (op, l, worst[l]))
            if not devs[l][op]['value'] < 1e-3:
                bad.append('%s %s: value deviation %.2e exceeds 1e-3' % (op, l, devs[l][op]['value']))
        for l in labels[1:]:
            if not worst[l] <= FACTOR * worst[labels[0]]:
                bad.append('%s: %s worst %.2e > %g x %s worst %.2e' % (op, l, worst[l], FACTOR, labels[0], worst[labels[0]]))
    assert not bad, '\n'.join(bad)
