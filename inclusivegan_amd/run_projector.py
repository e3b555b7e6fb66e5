#!/usr/bin/env python3
"""Projection of real images into the generator's latent space and the IvOM figure (mean LPIPS distance of the best
reconstructions), reference: run_projector.py:22-66.  `project_real_images` takes a snapshot pickle (this engine's or the
reference's) and a data set directory in the reference's record format."""
import argparse
import os

import numpy as np

from . import projector_lpips as projector
from .training import dataset
from .training import misc


def project_image(proj, targets, init_latents, png_prefix, num_snapshots):
    snapshot_steps = set(proj.num_steps - np.linspace(0, proj.num_steps, num_snapshots, endpoint=False, dtype=int))
    if png_prefix is not None:
        misc.save_image_grid(targets[:36], png_prefix + 'target.png', drange=[-1, 1])
    proj.start(targets, init_latents)
    while proj.get_cur_step() < proj.num_steps:
        proj.step()
        if png_prefix is not None and proj.get_cur_step() in snapshot_steps:
            misc.save_image_grid(proj.get_images()[:36], png_prefix + 'step%04d.png' % proj.get_cur_step(), drange=[-1, 1])
    return proj.get_dist()


def project_real_images(network_pkl, dataset_name, data_dir, num_images, minibatch_size, num_steps, num_snapshots, result_dir, device=None):
    print('Loading networks from "%s"...' % network_pkl)
    Gs = misc.as_networks(misc.load_pkl(network_pkl), device=device)[-1]
    proj = projector.Projector()
    print('Loading images from "%s"...' % dataset_name)
    dataset_obj = dataset.load_dataset(data_dir=data_dir, tfrecord_dir=dataset_name, max_label_size=0, repeat=True, shuffle_mb=0)
    assert dataset_obj.shape == Gs.output_shape[1:]
    os.makedirs(result_dir, exist_ok=True)
    proj.set_network(Gs, minibatch_size=minibatch_size, num_steps=num_steps)
    dists = None
    for image_idx in range(0, num_images, minibatch_size):
        print('Projecting image %d/%d ...' % (image_idx, num_images))
        images, _labels = dataset_obj.get_minibatch_np(minibatch_size)
        images = misc.adjust_dynamic_range(images.astype(np.float32), [0, 255], [-1, 1])
        dist = project_image(proj, targets=images, init_latents=None, png_prefix=os.path.join(result_dir, 'image%04d-' % image_idx), num_snapshots=num_snapshots)
        dists = np.array(dist) if dists is None else np.concatenate((dists, dist), axis=0)
    dist_mean, dist_std = np.mean(dists), np.std(dists)
    print('%s: IvOM = %.4f, std = %.4f' % (os.path.basename(network_pkl), dist_mean, dist_std))
    return dist_mean, dist_std


def main():
    p = argparse.ArgumentParser(description='StyleGAN2 + IMLE projector (MI355X).')
    sub = p.add_subparsers(dest='command')
    q = sub.add_parser('project-real-images')
    q.add_argument('--data-dir', required=True)
    q.add_argument('--dataset', dest='dataset_name', required=True)
    q.add_argument('--network', dest='network_pkl', required=True)
    q.add_argument('--result-dir', default='results', metavar='DIR')
    q.add_argument('--num-images', type=int, default=3000)
    q.add_argument('--num-snapshots', type=int, default=1)
    q.add_argument('--minibatch-size', type=int, default=50)
    q.add_argument('--num-steps', type=int, default=400)
    args = p.parse_args()
    if args.command is None:
        p.error('missing subcommand')
    kw = vars(args)
    kw.pop('command')
    project_real_images(**kw)


if __name__ == '__main__':
    main()
