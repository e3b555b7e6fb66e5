"""RCCL process group next to hipGraph capture (world size 1: the boxes have one GPU; the N>1 arithmetic is covered by the
gloo world-2 test in test_host_logic.py): init nccl, capture a GraphedStep while the watchdog thread is alive,
interleave replays with all-reduces."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_group_and_graph_capture_coexist(cuda_device):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dist_smoke.py')], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'ok ' in r.stdout and 'True' in r.stdout


def test_rccl_allreduce_inside_captured_graph(cuda_device):
    """The form the gradient exchange takes under RCCL: asynchronous all-reduces of bucket chunks issued from inside a captured
    region, waited at its end, replayed.  One rank (the collective degenerates to a copy but takes the same capture path); the
    communicator must stay usable afterwards."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dist_capture_probe.py')], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'replays correct = True' in r.stdout and 'RESULT capturable' in r.stdout and 'eager all-reduce after the capture: ok 8.0' in r.stdout


def _run_world2(mode, tmp_path, timeout=900, env=None):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), mode, str(r), '2', str(port), str(tmp_path)],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    import torch
    return [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(2)]


def test_non_finite_gradient_on_one_rank_stops_every_rank(cuda_device, tmp_path):
    """world 2 over gloo at BASELINE config 4's size (128x128, fmap_base 8192), rank 1's gradient non-finite.  The gate sits AFTER the
    exchange (/root/reference/dnnlib/tflib/optimizer.py:237 tests the all-reduced gradients), so nobody may update -- and both ranks take the next,
    finite step identically."""
    import torch
    r0, r1 = _run_world2('nonfinite', tmp_path, timeout=1800, env=dict(os.environ, IGAN_TEST_RES='128', IGAN_TEST_FMAP='8192'))
    for r in (r0, r1):
        assert r['poisoned_bucket_finite'] is False           # the poison reached BOTH ranks' averaged bucket
        assert r['poisoned_moved'] is False and r['poisoned_state_moved'] is False and r['poisoned_overflows'] == 1
        assert r['clean_bucket_finite'] is True and r['clean_moved'] is True and r['clean_overflows'] == 1
        assert r['clean_state_moved'] is True
    assert torch.equal(r0['poisoned_params'], r1['poisoned_params']) and torch.equal(r0['clean_params'], r1['clean_params'])


def test_two_rank_step_equals_averaged_single_process(cuda_device, tmp_path):
    """Two ranks (B = 3 each) take one G step and one D step through Optimizer.differentiate -- gradients placed, pre-scaled
    and all-reduced chunk by chunk from autograd hooks while backward runs -- and apply_updates.  Both ranks must end
    bit-identical, and equal to THIS process computing the two ranks' gradient buckets one after the other, averaging them
    (g0 / 2 + g1 / 2, dnnlib/tflib/optimizer.py:186,199) and applying the same Adam step."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dist_worker as W
    from inclusivegan_amd.training.dataset import SyntheticDataset
    r0, r1 = _run_world2('exchange', tmp_path)
    assert len(r0['chunks']) >= 2 and sum(r0['chunks']) == r0['g_avg'].numel()      # the bucket really went out in several pieces
    for k in ('G', 'D', 'g_avg', 'd_avg'):
        assert torch.equal(r0[k], r1[k]), k
    dev = cuda_device
    G, D, lp, G_opt, D_opt = W.build(dev)
    ts = SyntheticDataset(resolution=W.RES, label_size=0, data_size=24, device=dev)
    inps = [W.rank_inputs(r, dev) for r in range(2)]
    grads = []
    for r in range(2):
        W.g_backward(G, D, lp, G_opt, inps[r], r, ts, overlap=False)
        grads.append(G.flat_grads.clone())
    avg = grads[0] * 0.5 + grads[1] * 0.5
    assert torch.equal(avg.cpu(), r0['g_avg'])
    G.flat_grads.copy_(avg); G_opt.mark_registered(G); G_opt.apply_updates()
    assert torch.equal(G.flat_params.cpu(), r0['G'])
    grads = []
    for r in range(2):
        W.d_backward(G, D, D_opt, inps[r], r, ts, overlap=False)
        grads.append(D.flat_grads.clone())
    avg = grads[0] * 0.5 + grads[1] * 0.5
    assert torch.equal(avg.cpu(), r0['d_avg'])
    D.flat_grads.copy_(avg); D_opt.mark_registered(D); D_opt.apply_updates()
    assert torch.equal(D.flat_params.cpu(), r0['D'])
    assert float(r0['G'].abs().max()) > 0 and bool(torch.isfinite(r0['D']).all())


def test_two_rank_training_loop_keeps_replicas_identical(cuda_device, tmp_path):
    """The real training loop at world size 2 (BASELINE config 4's 2-rank leg as far as one GPU allows): sharded IMLE
    refresh + min-exchange, rank slices of the global minibatch, captured graphs with the gradient exchange after each
    replay.  After two iterations G, D and Gs are bit-identical on both ranks and finite, and both ranks fed the same host
    permutations."""
    import torch
    r0, r1 = _run_world2('loop', tmp_path)
    for k in ('G', 'D', 'Gs'):
        assert torch.equal(r0[k], r1[k]), k
        assert bool(torch.isfinite(r0[k]).all())
    assert r0['fed'] == r1['fed'] and len(r0['fed']) == 2


def test_bench_launches_its_own_ranks(cuda_device):
    """`python bench.py --gpus 2` with no launcher around it (the way the driver calls it): the parent spawns the ranks through
    torch.distributed.run before touching the GPU, the ranks run the real loop on their slices and rank 0 prints ONE JSON line with
    the whole-job rate.  Both ranks share GPU 0 and talk over gloo here (--one-gpu --backend gloo): everything but RCCL itself."""
    import json
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--one-gpu', '--backend', 'gloo', '--steps', '3', '--warmup', '1', '--no-roofline', '--no-cpu-baseline',
                        '--resolution', '32', '--minibatch-gpu', '3', '--data-size', '48', '--num-samples-factor', '2'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 6 and d['config']['images_per_step'] == 12 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and abs(d['value'] - 12 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']


def test_bench_eight_ranks_at_the_bench_size(cuda_device):
    """The driver's multi-GPU command at the bench configuration itself (128x128 config-e, minibatch_gpu 6), eight ranks on this one
    GPU over gloo, 2 timed steps: the self-launch path, the data_size rounding to a multiple of 2 * minibatch_gpu * ranks, the
    barrier-bracketed max-over-ranks timing and the whole-job rate -- everything of an 8-GPU run but RCCL and the other seven devices."""
    import json
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--one-gpu', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--no-roofline',
                        '--no-cpu-baseline', '--data-size', '1000', '--num-samples-factor', '1'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    print(lines[0])
    assert d['n_gpus'] == 8 and d['steps'] == 2 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp8', d
    assert d['config']['data_size'] == 960 and d['config']['data_size'] % (2 * 6 * 8) == 0          # 1000 rounded down
    assert d['config']['global_batch'] == 48 and d['config']['images_per_step'] == 96
    assert abs(d['value'] - 96 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']
    assert d['rccl'] == {'ranks': 8, 'backend': 'gloo', 'in_graph': False, 'version': None}, d['rccl']
    assert d['hip_graphs']['captured'] and d['hip_graphs']['faithful'], d['hip_graphs']


def _eight_rank_replay_stress(lib=None, steps=17, timeout=3000):
    """bench.py with eight ranks on this one GPU and `--revalidate-every 1`: after every iteration each of the four captured training ops is replayed
    against its eager execution again (bit for bit), under the load of the seven other processes.  Returns the bench line's hip_graphs record."""
    import json
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    if lib is not None:
        env['IGAN_LIB'] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--one-gpu', '--backend', 'gloo', '--steps', str(steps), '--warmup', '1', '--no-roofline',
                        '--no-cpu-baseline', '--data-size', '1000', '--num-samples-factor', '1', '--revalidate-every', '1'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])['hip_graphs']


def test_eight_rank_replay_stress(cuda_device):
    """the multi-process replay mismatch of round 4 showed only in whole training ops replayed beside seven other processes (the piece
    kernels alone, eager or replayed from eight processes, never differed: profiles/r04_f16_pairs_variant.txt section 7), so the stress runs at that level:
    eight ranks on one GPU, 17 iterations, every captured op re-validated against its eager execution after every iteration -- 4 ops x 18 checks on
    each of 8 ranks, each with the weight gradients of all large layers on conv_wgrad_planes_kernel<2>.  (tools/r5_replay_stress.sh runs the same
    command against a build with the weight-gradient kernel's LDS-DMA issued through the compiler builtin -- the build that failed in round 4 -- and
    profiles/r05_replay_stress.txt has both outcomes.)"""
    g = _eight_rank_replay_stress()
    assert g['captured'] and g['faithful'], g
    assert len(g['checks']) >= 17 and all(c['faithful'] for c in g['checks']), g['checks']


@pytest.mark.parametrize('size', ['32x32_fmap256', '128x128_fmap8192'])
def test_eight_rank_loop_at_config5_shape(cuda_device, tmp_path, size):
    """BASELINE config 5 (8 GPUs, minibatch_gpu 3, attribute-masked selection) as far as one GPU allows: EIGHT ranks on GPU 0 over
    gloo run two iterations of the real loop -- at a reduced size and at config 5's own (128x128, fmap_base 8192: eight replicas of the
    full networks and their captured graphs in one device's memory; the oracle comparison at that size is tests/test_gpu_loop_parity.py's).  Replicas end bit-identical; the ranks' slices tile the global minibatch (reals picked by
    the attribute mask, latents, labels); the refresh sharded over eight ranks + the two min-exchanges assign every real the same
    candidate as ONE process searching all candidates (same seeds; noise strengths are zero at initialisation, so the candidate
    images do not depend on the ranks' device generators)."""
    import numpy as np
    import torch
    world = 8
    full = size.startswith('128')
    env8 = dict(os.environ, IGAN_TEST_RES='128' if full else '32', IGAN_TEST_FMAP='8192' if full else '256')
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), 'config5', str(r), str(world), str(port), str(tmp_path)],
                              cwd=ROOT, env=env8, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=2400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    recs = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(world)]
    for r in recs[1:]:
        for k in ('G', 'D', 'Gs'):
            assert torch.equal(recs[0][k], r[k]), k
        assert np.array_equal(recs[0]['assign'][0][0], r['assign'][0][0])
    assert bool(torch.isfinite(recs[0]['G']).all()) and len(recs[0]['fed']) == 2
    col = 31        # 'Smiling' in the CelebA attribute order
    for it in range(2):
        fed = recs[0]['fed'][it]
        assert fed['latents_rec_1'].shape == (24, 512) and fed['labels_rec_1'].shape == (24, 40)
        assert bool((fed['labels_rec_1'][:, col] == 1).all()) and bool((fed['labels_rec_2'][:, col] == 1).all())      # the AND-mask selection (:416-424)
        lat = np.concatenate([recs[r]['slices'][it]['lat'] for r in range(world)])
        lab = np.concatenate([recs[r]['slices'][it]['lab'] for r in range(world)])
        assert np.array_equal(lat, fed['latents_rec_1'].astype(np.float32)) and np.array_equal(lab, fed['labels_rec_1'].astype(np.float32))
    # one process, the whole minibatch of 24: the same host stream, hence the same candidates -- and the same assignment
    single = tmp_path / 'single'
    single.mkdir()
    env = dict(env8, IGAN_TEST_MB_GPU='24')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), 'config5', '0', '1', str(port), str(single)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    one = torch.load(os.path.join(str(single), 'rank0.pt'), weights_only=False)
    assert np.array_equal(one['assign'][0][0], recs[0]['assign'][0][0])
    assert np.allclose(one['assign'][0][1], recs[0]['assign'][0][1], rtol=1e-6, atol=0)
    assert np.array_equal(one['fed'][0]['latents_rec_1'], recs[0]['fed'][0]['latents_rec_1'])
