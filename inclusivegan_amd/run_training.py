#!/usr/bin/env python3
"""Command line + configuration assembly for the HIP training engine: same flags and the same
nested `*_args` dictionaries as the reference's `run_training.py` (flags :191-216, assembly :36-165),
so that a config built here is interchangeable with one built there (checked against the
reference's own output in tests/test_golden.py).

`build_kwargs(...)` returns exactly what the reference hands to `dnnlib.submit_run`; `run(...)`
calls `inclusivegan_amd.training.training_loop.training_loop` with it (one process per GPU; for
N > 1 launch under `python -m torch.distributed.run --nproc-per-node N`).  Run-directory creation,
log teeing and metric scheduling (dnnlib/submission) are out of scope.
"""
import argparse
import copy
import os
import sys

from .dnnlib import EasyDict

_CONFIGS_E = ['config-e-G%s-D%s' % (g, d) for g in ('orig', 'resnet', 'skip') for d in ('orig', 'resnet', 'skip')]
_valid_configs = ['config-a', 'config-b', 'config-c', 'config-d', 'config-e', 'config-f'] + _CONFIGS_E

from .metrics.metric_defaults import metric_defaults      # the metrics built here: mode_counts_24k, KL24k, fid30k (run_training.py:21 of the reference)


def build_kwargs(dataset, data_dir, result_dir, config_id, num_gpus, gamma, mirror_augment, metrics, resume_pkl,
                 minibatch_gpu, data_size, num_epochs, init_proj_dim, init_staleness, num_samples_factor, knn_perturb_factor,
                 candidate_batch_size, exclusive_retrieved_code, NN_rec_lpips_weight, dist_thres_percentile, attr_interesting,
                 init_mul):
    assert config_id in _valid_configs
    if config_id in ('config-a', 'config-b', 'config-c', 'config-d'):
        raise NotImplementedError('configs a-d (progressive growing / StyleGAN-1 nets) are outside the hot path; '
                                  'their network functions do not exist in the reference either (SURVEY.md section 2.1)')

    # Network / loss / optimizer options (run_training.py:51-61).
    G = EasyDict(func_name='training.networks_stylegan2.G_main', init_mul=init_mul)
    D = EasyDict(func_name='training.networks_stylegan2.D_stylegan2_feature')
    G_opt = EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8)
    D_opt = EasyDict(beta1=0.0, beta2=0.99, epsilon=1e-8)
    G_loss = EasyDict(func_name='training.loss.G_logistic_ns_rec_interp_arb_pathreg', NN_rec_lpips_weight=NN_rec_lpips_weight)
    D_loss = EasyDict(func_name='training.loss.D_logistic_r1', gamma=10.0)
    sched = EasyDict(G_lrate_base=0.002, D_lrate_base=0.002, minibatch_gpu_base=minibatch_gpu, minibatch_size_base=minibatch_gpu * num_gpus)
    grid = EasyDict(size='1080p', layout='random')
    tf_config = {'rnd.np_random_seed': 1000, 'gpu_options.allow_growth': False, 'graph_options.place_pruned_graph': True}

    # Architecture switches (:116-127).
    if config_id != 'config-f':
        G.fmap_base = D.fmap_base = 8 << 10
    if config_id.startswith('config-e'):
        D_loss.gamma = 100
        for tag, target in (('G', G), ('D', D)):
            for arch in ('orig', 'skip', 'resnet'):
                if tag + arch in config_id:
                    target.architecture = arch
    if gamma is not None:
        D_loss.gamma = gamma

    # Run description (:90-113).
    desc = 'stylegan2-%s-%dgpu-%s' % (dataset, num_gpus, config_id)
    desc += '_noProj' if init_proj_dim is None else '_%dProj' % init_proj_dim
    desc += '_init_staleness_%d_num_samples_factor_%d_knn_perturb_factor_%f_NN_rec_lpips_weight_%f' % (
        init_staleness, num_samples_factor, knn_perturb_factor, NN_rec_lpips_weight)
    if attr_interesting is not None:
        desc += '_%s' % attr_interesting.replace(',', '_and_')
    desc += '_scratch' if (resume_pkl is None or '_scratch' in resume_pkl) else '_finetune'

    train = EasyDict(
        data_dir=data_dir, total_kimg=(data_size * num_epochs) // 1000, mirror_augment=mirror_augment, resume_pkl=resume_pkl,
        data_size=data_size, num_epochs=num_epochs, init_proj_dim=init_proj_dim, init_staleness=init_staleness,
        num_samples_factor=num_samples_factor, knn_perturb_factor=knn_perturb_factor, candidate_batch_size=candidate_batch_size,
        exclusive_retrieved_code=exclusive_retrieved_code, dist_thres_percentile=dist_thres_percentile, attr_interesting=attr_interesting)

    out = EasyDict(run_func_name='training.training_loop.training_loop', num_gpus=num_gpus, run_desc=desc)
    out.update(train)
    out.update(G_args=G, D_args=D, G_opt_args=G_opt, D_opt_args=D_opt, G_loss_args=G_loss, D_loss_args=D_loss)
    out.update(dataset_args=EasyDict(tfrecord_dir=dataset, max_label_size='full'), sched_args=sched, grid_args=grid,
               metric_arg_list=[metric_defaults[m] for m in metrics], tf_config=tf_config)
    return out


def run(attr_file=None, **args):
    """`attr_file`: CelebA's Anno/list_attr_celeba.txt for --attr-interesting (the reference reads
    'celeba/Anno/list_attr_celeba.txt' relative to the working directory, training_loop.py:174-180; that path is tried
    too).  With the synthetic CelebA-shaped source the 40 label columns carry CelebA's attribute order."""
    from .training import training_loop as TL
    from .training import imle
    kw = build_kwargs(**args)
    run_desc = kw['run_desc']
    for k in ('run_func_name', 'num_gpus', 'run_desc'):
        kw.pop(k)
    kw = copy.deepcopy(kw)
    # the run directory dnnlib.submit_run would create: <result_dir>/<next 5-digit id>-<run_desc> (dnnlib/submission/submit.py:_create_run_dir_local)
    result_dir = args.get('result_dir')
    if result_dir is not None and int(os.environ.get('RANK', '0')) == 0:
        os.makedirs(result_dir, exist_ok=True)
        ids = [int(d.split('-')[0]) for d in os.listdir(result_dir) if d.split('-')[0].isdigit() and os.path.isdir(os.path.join(result_dir, d))]
        kw['run_dir'] = os.path.join(result_dir, '%05d-%s' % (max(ids) + 1 if ids else 0, run_desc))
    # synthetic data source (no tfrecords reader in this round): shape follows the dataset name
    ds = kw['dataset_args']
    if 'mnist' in ds['tfrecord_dir']:
        ds.update(resolution=32, num_channels=3, label_size=1000, label_kind='onehot')
    else:
        ds.update(resolution=128, num_channels=3, label_size=40, label_kind='attributes')
    if kw.get('attr_interesting') is not None:
        candidates = [f for f in (attr_file, 'celeba/Anno/list_attr_celeba.txt') if f and os.path.isfile(f)]
        if attr_file is not None and not candidates:
            raise FileNotFoundError('--attr-file %s does not exist' % attr_file)
        kw['attr_names'] = imle.attribute_names(candidates[0]) if candidates else list(imle.CELEBA_ATTRIBUTES)
        unknown = [a for a in kw['attr_interesting'].split(',') if a not in kw['attr_names']]
        if unknown:
            raise ValueError('--attr-interesting: unknown attribute(s) %s; known: %s' % (unknown, ', '.join(kw['attr_names'])))
    return TL.training_loop(**kw)


def _str_to_bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def _parse_comma_sep(s):
    if s is None or s.lower() == 'none' or s == '':
        return []
    return s.split(',')


def main():
    p = argparse.ArgumentParser(description='Train StyleGAN2 + IMLE on MI355X.', formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--result-dir', default='results', metavar='DIR')
    p.add_argument('--data-dir', required=True)
    p.add_argument('--dataset', required=True)
    p.add_argument('--config', default='config-e', dest='config_id', metavar='CONFIG')
    p.add_argument('--init-mul', default=1.0, type=float)
    p.add_argument('--num-gpus', default=1, type=int, metavar='N')
    p.add_argument('--gamma', default=None, type=float)
    p.add_argument('--mirror-augment', default=False, metavar='BOOL', type=_str_to_bool)
    p.add_argument('--metrics', default='none', type=_parse_comma_sep)
    p.add_argument('--minibatch-gpu', metavar='N', default=6, type=int)
    p.add_argument('--data-size', metavar='N', default=30000, type=int)
    p.add_argument('--num-epochs', metavar='N', default=10000, type=int)
    p.add_argument('--init-proj-dim', metavar='N', default=None, type=int)
    p.add_argument('--init-staleness', metavar='N', default=10, type=int)
    p.add_argument('--num-samples-factor', metavar='N', default=10, type=int)
    p.add_argument('--knn-perturb-factor', default=0.05, type=float)
    p.add_argument('--candidate-batch-size', metavar='N', default=256, type=int)
    p.add_argument('--exclusive-retrieved-code', metavar='N', default=0, type=int)
    p.add_argument('--NN-rec-lpips-weight', default=2.5, type=float)
    p.add_argument('--dist-thres-percentile', default=100.0, type=float)
    p.add_argument('--attr-interesting', default=None, type=str)
    p.add_argument('--resume-pkl', default=None, type=str)
    p.add_argument('--attr-file', default=None, type=str, help='CelebA Anno/list_attr_celeba.txt (attribute vocabulary for --attr-interesting)')
    args = p.parse_args()
    if args.config_id not in _valid_configs:
        print('Error: --config value must be one of: ', ', '.join(_valid_configs))
        sys.exit(1)
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    from . import hostaffinity
    hostaffinity.limit_host_threads()           # one process per GPU with a small CPU pool (hostaffinity.py: a 128-thread OpenMP pool starves the graph submissions)
    hostaffinity.pin_to_device_node(int(os.environ.get('LOCAL_RANK', '0')))     # IGAN_PIN_NUMA=1 only
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        # generous timeout: rank 0 evaluates the snapshot metrics alone while the other ranks wait at the next collective
        torch.distributed.init_process_group('nccl', timeout=datetime.timedelta(hours=4))
    assert world == args.num_gpus, 'one process per GPU: launch %d ranks' % args.num_gpus
    run(**vars(args))


if __name__ == '__main__':
    main()
