"""Second-order steps (path-length regulariser, R1) of ONE fixed state under each arithmetic form of the large 3x3 convolutions, against the fp64 oracle
(reference: training/loss.py:55-89 path length, :107-111 R1; the ops training_loop.py:288-289 registers).

The convolution form is a per-process decision of the library (IGAN_CONV_PLANES, IGAN_PLANES_MIN_ROWS, ... are read once), so every form runs in a
child process (`python -m tests.reg_forms child <state.npz> <out.npz>`) on a state file the parent wrote:

    weights of G and D by variable name, dlatent_avg, pl_mean, the random draws of the two ops as tapes (product order), the reals of the R1 step

The child replays the draws (tflib.tfutil.RandomTape), so every form -- and the oracle -- sees the same latents, noise inputs and image-space noise.
What is compared is what tests/test_gpu_networks.py compares: relative L2 per trainable variable (scalar parameters jointly), the op's value and pl_mean.

`pl_frac`: pl_mean of the state as a fraction of the batch's mean path length.  At initialisation pl_mean is 0; in a trained network it tracks the lengths,
and the penalty (pl_lengths - pl_mean)^2 becomes a difference of nearly equal numbers whose gradient amplifies every error of the path-length VALUE by
pl_length / (pl_length - pl_mean) -- the regime in which round 5 saw the step react to the convolution arithmetic (profiles/r05_small_layers.txt section 5-6).

Used by tests/test_gpu_reg_forms.py (the bench configuration, forms 0 / 1 / 2) and tools/reg_forms.py (thresholds, loop states, config 2)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_nets(dev, res, fmap, seed=11):
    """G / D at random initialisation, biases and noise strengths made non-zero (as tests/test_gpu_networks.py does)."""
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=res, label_size=0, fmap_base=fmap, device=dev)
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=seed, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=seed + 1, **kw)
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for net in (G, D):
            for n, v in net.vars.items():
                if n.endswith('bias') or n.endswith('noise_strength'):
                    v.copy_(torch.from_numpy(np.asarray(rng.randn(*v.shape) * 0.1, dtype=np.float32)).to(v.device).reshape(v.shape))
    return G, D


def _tape_arrays(prefix, entries):
    out = {prefix + '_kinds': np.array([k for k, _ in entries])}
    for i, (_, v) in enumerate(entries):
        out['%s_%03d' % (prefix, i)] = np.asarray(v)
    return out


def _tape_from(z, prefix):
    kinds = [str(k) for k in z[prefix + '_kinds']]
    return [(k, z['%s_%03d' % (prefix, i)]) for i, k in enumerate(kinds)]


def save_state(path, cfg, G_vars, D_vars, pl_means, tape_G, tape_D, reals):
    """cfg: dict(res, fmap, B).  G_vars / D_vars: name -> ndarray (every variable, trainable or not).  pl_means: the path-length step is evaluated once per
    entry (ops 'G_reg@0', 'G_reg@1', ...: same weights and draws, another moving average).  reals: [2B, 3, res, res] float32 in [-1, 1]."""
    arrays = dict(cfg=np.array(json.dumps(cfg)), pl_means=np.asarray(pl_means, np.float64), reals=np.asarray(reals, np.float32))
    arrays.update({'G/' + n: np.asarray(v) for n, v in G_vars.items()})
    arrays.update({'D/' + n: np.asarray(v) for n, v in D_vars.items()})
    arrays.update(_tape_arrays('tapeG', tape_G))
    arrays.update(_tape_arrays('tapeD', tape_D))
    np.savez(path, **arrays)


def save_state_dict(path, state):
    save_state(path, state['cfg'], state['G'], state['D'], state['pl_means'], state['tape_G'], state['tape_D'], state['reals'])
    if state.get('tape_Dloss'):          # the first-order D step (op 'D_loss'): its own draws; it reads the same `reals`
        z = dict(np.load(path, allow_pickle=False))
        z.update(_tape_arrays('tapeDloss', state['tape_Dloss']))
        np.savez(path, **z)


def load_state(path):
    z = np.load(path, allow_pickle=False)
    cfg = json.loads(str(z['cfg']))
    G_vars = {k[2:]: z[k] for k in z.files if k.startswith('G/')}
    D_vars = {k[2:]: z[k] for k in z.files if k.startswith('D/')}
    st = dict(cfg=cfg, G=G_vars, D=D_vars, pl_means=[float(v) for v in z['pl_means']], tape_G=_tape_from(z, 'tapeG'), tape_D=_tape_from(z, 'tapeD'), reals=z['reals'])
    if 'tapeDloss_kinds' in z.files:
        st['tape_Dloss'] = _tape_from(z, 'tapeDloss')
    return st


def _assign(net, vars_):
    with torch.no_grad():
        for n, v in net.vars.items():
            v.copy_(torch.from_numpy(np.ascontiguousarray(vars_[n])).to(v.device).reshape(v.shape))


def hip_ops_of_state(state, dev, ops=('G_reg', 'D_reg'), record=False):
    """The two regulariser ops on the HIP path from `state` (draws replayed, or recorded when record=True: then state['tape_*'] are filled in).
    -> dict op -> dict(value=[per-sample reg], grads={name: ndarray}, pl_mean=float (G_reg))."""
    from inclusivegan_amd.dnnlib.tflib import tfutil
    from inclusivegan_amd.training import loss as PL
    from inclusivegan_amd.training.dataset import SyntheticDataset
    cfg = state['cfg']
    res, fmap, B = cfg['res'], cfg['fmap'], cfg['B']
    G, D = make_nets(dev, res, fmap)
    _assign(G, state['G']); _assign(D, state['D'])
    ts = SyntheticDataset(resolution=res, label_size=0, data_size=24, device=dev)
    lab = torch.zeros(B, 0, device=dev)
    zdev = torch.zeros(B, 512, device=dev)
    out = {}
    for i, pl_mean in enumerate(state['pl_means'] if 'G_reg' in ops else []):
        G.zero_grad(); D.requires_grad_(False)
        G.pl_mean_var = torch.tensor(pl_mean, device=dev, dtype=torch.float32)
        src = tfutil.RecordingRandom() if (record and i == 0) else tfutil.RandomTape(state['tape_G'])
        with tfutil.use_random(src):
            _, reg = PL.G_logistic_ns_rec_interp_arb_pathreg(G, D, None, ts, B, None, lab, zdev, None, lab, zdev, NN_rec_lpips_weight=2.5, phase='reg')
        torch.autograd.backward(reg.mean(), inputs=list(G.trainables.values()))
        D.requires_grad_(True)
        torch.cuda.synchronize()
        if record and i == 0:
            state['tape_G'] = list(src.entries)
        out['G_reg@%d' % i] = dict(value=reg.detach().double().cpu().numpy(), pl_mean=float(G.pl_mean_var),
                                   grads={n: v.grad.detach().cpu().numpy().copy() for n, v in G.trainables.items() if v.grad is not None})
        _assign(G, state['G'])      # dlatent_avg moved (networks_stylegan2.py:203-209)
    if 'D_reg' in ops:
        G.zero_grad(); D.zero_grad()
        reals = torch.from_numpy(state['reals']).to(dev).contiguous(memory_format=torch.channels_last)
        lab2 = torch.zeros(reals.shape[0], 0, device=dev)
        src = tfutil.RecordingRandom() if record else tfutil.RandomTape(state['tape_D'])
        with tfutil.use_random(src):
            _, reg = PL.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase='reg')
        torch.autograd.backward(reg.mean(), inputs=list(D.trainables.values()))
        torch.cuda.synchronize()
        if record:
            state['tape_D'] = list(src.entries)
        out['D_reg'] = dict(value=reg.detach().double().cpu().numpy(),
                            grads={n: v.grad.detach().cpu().numpy().copy() for n, v in D.trainables.items() if v.grad is not None})
    if 'D_loss' in ops:          # the first-order D step (training/loss.py:93-105): G forward without gradients, D(fakes) and D(reals), softplus
        _assign(G, state['G'])
        G.zero_grad(); D.zero_grad()
        reals = torch.from_numpy(state['reals']).to(dev).contiguous(memory_format=torch.channels_last)
        lab2 = torch.zeros(reals.shape[0], 0, device=dev)
        G.requires_grad_(False)
        with tfutil.use_random(tfutil.RandomTape(state['tape_Dloss'])):
            loss, _ = PL.D_logistic_r1(G, D, ts, B, reals, lab2, gamma=100, phase='loss')
        G.requires_grad_(True)
        torch.autograd.backward(loss.mean(), inputs=list(D.trainables.values()))
        torch.cuda.synchronize()
        out['D_loss'] = dict(value=loss.detach().double().cpu().numpy(),
                             grads={n: v.grad.detach().cpu().numpy().copy() for n, v in D.trainables.items() if v.grad is not None})
        _assign(G, state['G'])
    names = dict(G=list(G.trainables), D=list(D.trainables))
    return out, names


def oracle_ops_of_state(state, ops=('G_reg', 'D_reg'), trainables=None, dtype=torch.float64):
    """The same ops on the fp64 oracle (oracle/loss.py).  `trainables`: dict(G=[names], D=[names]) (which variables take a gradient).
    dtype=torch.float32: the SAME restatement evaluated in fp32 by PyTorch's CPU kernels -- not an oracle, a second fp32 implementation (how far
    does fp32 arithmetic as such sit from fp64 on this state?)."""
    from oracle import loss as OL
    from oracle.misc import Tape
    cfg = state['cfg']
    res, fmap, B = cfg['res'], cfg['fmap'], cfg['B']
    ocfg = dict(resolution=res, num_channels=3, fmap_base=fmap, G_arch='skip', D_arch='resnet')
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def params(vars_, names):
        p = {n: torch.from_numpy(np.asarray(v, np.float64)).to(dtype) for n, v in vars_.items()}
        for n in names:
            p[n].requires_grad_(True)
        return p

    out = {}
    z = torch.zeros(B, 512, dtype=dtype)
    for i, pl_mean in enumerate(state['pl_means'] if 'G_reg' in ops else []):
        gp, dp = params(state['G'], trainables['G']), params(state['D'], [])
        st = dict(pl_mean=torch.tensor(pl_mean, dtype=dtype), dlatent_avg=gp['dlatent_avg'])
        t0 = time.time()
        _, reg, _ = OL.G_loss(gp, dp, {}, ocfg, Tape(state['tape_G'], dtype), B, None, z, None, z, 2.5, phase='reg', state=st)
        reg.mean().backward()
        out['G_reg@%d' % i] = dict(value=reg.detach().double().numpy(), pl_mean=float(st['pl_mean']), seconds=time.time() - t0,
                                   grads={n: gp[n].grad.double().numpy() for n in trainables['G'] if gp[n].grad is not None})
    if 'D_reg' in ops:
        gp, dp = params(state['G'], []), params(state['D'], trainables['D'])
        t0 = time.time()
        _, reg, _ = OL.D_loss(gp, dp, ocfg, Tape(state['tape_D'], dtype), B, torch.from_numpy(state['reals']).to(dtype), gamma=100, phase='reg', state={})
        reg.mean().backward()
        out['D_reg'] = dict(value=reg.detach().double().numpy(), seconds=time.time() - t0,
                            grads={n: dp[n].grad.double().numpy() for n in trainables['D'] if dp[n].grad is not None})
    if 'D_loss' in ops:
        gp, dp = params(state['G'], []), params(state['D'], trainables['D'])
        t0 = time.time()
        loss, _, _ = OL.D_loss(gp, dp, ocfg, Tape(state['tape_Dloss'], dtype), B, torch.from_numpy(state['reals']).to(dtype), gamma=100, phase='loss',
                               state=dict(dlatent_avg=gp['dlatent_avg']))
        loss.mean().backward()
        out['D_loss'] = dict(value=loss.detach().double().numpy(), seconds=time.time() - t0,
                             grads={n: dp[n].grad.double().numpy() for n in trainables['D'] if dp[n].grad is not None})
    torch.set_num_threads(threads_before)
    return out


def deviations(hip, ora):
    """Relative L2 per variable (scalar parameters jointly, as tests/test_gpu_networks.py:_grad_errs) + the op's value deviations."""
    errs, sh, so = {}, [], []
    for n, go in ora['grads'].items():
        gh = hip['grads'][n].astype(np.float64).reshape(go.shape)
        if go.size == 1:
            sh.append(gh.reshape(1)); so.append(go.reshape(1))
            continue
        if not np.any(go):
            assert not np.any(gh), (n, 'gradient where the oracle has none')
            continue
        errs[n] = float(np.linalg.norm(gh - go) / np.linalg.norm(go))
    if so:
        errs['<all scalar parameters>'] = float(np.linalg.norm(np.concatenate(sh) - np.concatenate(so)) / (np.linalg.norm(np.concatenate(so)) + 1e-30))
    val = float(np.abs(hip['value'] - ora['value']).max() / (np.abs(ora['value']).max() + 1e-30))
    res = dict(errs=errs, value=val)
    if 'pl_mean' in ora:
        res['pl_mean'] = abs(hip['pl_mean'] - ora['pl_mean']) / (abs(ora['pl_mean']) + 1e-30)
    return res


def save_result(path, result):
    arrays = {}
    for op, r in result.items():
        arrays[op + '/value'] = r['value']
        if 'pl_mean' in r:
            arrays[op + '/pl_mean'] = np.float64(r['pl_mean'])
        for n, g in r['grads'].items():
            arrays[op + '/g/' + n] = g
    np.savez(path, **arrays)


def load_result(path):
    z = np.load(path, allow_pickle=False)
    out = {}
    for k in z.files:
        op, rest = k.split('/', 1)
        r = out.setdefault(op, dict(grads={}))
        if rest == 'value':
            r['value'] = z[k]
        elif rest == 'pl_mean':
            r['pl_mean'] = float(z[k])
        else:
            r['grads'][rest[2:]] = z[k]
    return out


def run_child(state_path, out_path, env, ops=('G_reg', 'D_reg'), timeout=1800):
    """One form = one process: `env` (IGAN_CONV_PLANES etc.) on top of the caller's.  Returns the child's info line (kernel form, kernel names)."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-m', 'tests.reg_forms', 'child', state_path, out_path, ','.join(ops)], env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('INFO ')][-1][5:])


def init_state(dev, res, fmap, B, pl_fracs, seed=5):
    """State at random initialisation: draws recorded from one HIP evaluation of the two ops (under the calling process's form; the draws do not depend
    on it), pl_means = pl_fracs x the batch's mean path length as that evaluation measured it."""
    G, D = make_nets(dev, res, fmap)
    g = torch.Generator().manual_seed(seed)
    reals = (torch.rand(2 * B, 3, res, res, generator=g) * 2 - 1).numpy()
    state = dict(cfg=dict(res=res, fmap=fmap, B=B), G={n: v.detach().cpu().numpy() for n, v in G.vars.items()}, D={n: v.detach().cpu().numpy() for n, v in D.vars.items()},
                 pl_means=[0.0], tape_G=[], tape_D=[], reals=reals)
    del G, D
    torch.manual_seed(seed)
    out, names = hip_ops_of_state(state, dev, record=True)
    mean_len = out['G_reg@0']['pl_mean'] / 0.01          # pl_mean = 0 + pl_decay * mean(pl_lengths), loss.py:71
    state['pl_means'] = [float(f * mean_len) for f in pl_fracs]
    state['mean_path_length'] = mean_len
    return state, names


def table(devs, labels, top=6):
    """devs: label -> op -> deviations().  Text table: per op the value deviation, worst variables of every label, and every label's figure on the
    union of those variables."""
    lines = []
    ops = list(next(iter(devs.values())).keys())
    for op in ops:
        lines.append('%s: value %s%s' % (op, '  '.join('%s %.2e' % (l, devs[l][op]['value']) for l in labels),
                                         ('   pl_mean ' + '  '.join('%s %.2e' % (l, devs[l][op]['pl_mean']) for l in labels)) if 'pl_mean' in devs[labels[0]][op] else ''))
        names = []
        for l in labels:
            e = devs[l][op]['errs']
            for n in sorted(e, key=e.get, reverse=True)[:top]:
                if n not in names:
                    names.append(n)
        lines.append('  %-44s %s' % ('variable (relative L2 of its gradient vs fp64)', ' '.join('%10s' % l for l in labels)))
        for n in names:
            lines.append('  %-44s %s' % (n, ' '.join('%10.2e' % devs[l][op]['errs'].get(n, float('nan')) for l in labels)))
        lines.append('  %-44s %s' % ('WORST', ' '.join('%10.2e' % max(devs[l][op]['errs'].values()) for l in labels)))
        lines.append('  %-44s %s' % ('median', ' '.join('%10.2e' % float(np.median(list(devs[l][op]['errs'].values()))) for l in labels)))
    return '\n'.join(lines)


def _child_main(argv):
    state_path, out_path = argv[0], argv[1]
    ops = tuple(argv[2].split(',')) if len(argv) > 2 else ('G_reg', 'D_reg')
    import inclusivegan_amd  # noqa: F401  (sets the runtime flags before HIP initialises)
    from inclusivegan_amd import _abi
    lib = _abi.get_plugin()
    dev = torch.device('cuda', 0)
    state = load_state(state_path)
    out, _ = hip_ops_of_state(state, dev, ops=ops)
    save_result(out_path, out)
    print('INFO ' + json.dumps(dict(form=int(lib.igan_conv_piece_form()), min_rows=os.environ.get('IGAN_PLANES_MIN_ROWS', 'default'),
                                    wgrad_min_rows=os.environ.get('IGAN_WGRAD_PLANES_MIN_ROWS', 'default'))))


if __name__ == '__main__':
    if len(sys.argv) >= 4 and sys.argv[1] == 'child':
        _child_main(sys.argv[2:])
    else:
        raise SystemExit('usage: python -m tests.reg_forms child <state.npz> <out.npz> [ops]')
