"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Host-side IMLE bookkeeping of the reference's main loop, `training/training_loop.py:325-482`, restated statement by
statement in NumPy (three parallel arrays for reals / labels / latents, the same order of draws from the global NumPy
stream, the same cursor arithmetic), with the TensorFlow-side objects passed in as callables:

    generate(latents, labels)          G.run(..., is_validation=True)                      :361
    build_index() -> index             dci_db.reset(); dci_db.add(candidates, ...)          :367-368
    index.query(q, k) -> (idx, dist)   dci_db.query(q, num_neighbours=k, ...)               :386,398  ([nq,k] arrays)
    step(feed)                         the tflib.run(...) calls of one iteration            :466-479

PINNED: tests/golden/imle_host_golden.npz holds the results of executing the reference's own statements of these lines
(tests/golden/make_imle_golden.py cuts them out of /root/reference/training/training_loop.py and runs them verbatim);
tests/test_imle_host.py requires this restatement to reproduce them exactly.
"""
import numpy as np

from oracle.misc import slerp_np, adjust_dynamic_range


def imle_host_loop(training_set, training_set_rec, latent_shape, generate, make_index, step, *, data_size, num_samples_factor,
                   init_staleness, candidate_batch_size, minibatch_size, minibatch_repeats, total_img, knn_perturb_factor,
                   dist_thres_percentile=100.0, attr_interesting=None, attr_names=None, exclusive_retrieved_code=0,
                   projector=None, drange_net=(-1, 1)):
    drange_net = list(drange_net)
    mb = minibatch_size
    N = data_size * num_samples_factor
    cur_nimg = 0
    cursor = 0
    Z = np.random.randn(N, *latent_shape).astype(np.float32)                                  # :325
    sel_Z = None
    rem_R = rem_L = rem_Z = None                                                              # :328-330
    log = dict(refresh_at=[], nearest_indices=[], nearest_dists=[])
    beginning = False
    while cur_nimg < total_img:                                                               # :332
        training_set.configure(mb * 2, 0)                                                     # :339
        training_set_rec.configure(mb * 2, 0)                                                 # :340
        for _ in range(minibatch_repeats):                                                    # :348
            period = data_size * init_staleness
            if sel_Z is None or cur_nimg // period != (cur_nimg - mb * 2) // period:          # :354
                if sel_Z is not None:
                    init_staleness *= 2                                                       # :355-356
                cand_labels = training_set_rec.get_random_labels_np(N)                        # :357
                cands = None
                for i in range(N // candidate_batch_size + 1):                                # :359
                    lo, hi = i * candidate_batch_size, (i + 1) * candidate_batch_size
                    img = generate(Z[lo:hi, :], cand_labels[lo:hi, :])                        # :361
                    flat = np.reshape(img, (-1, np.prod(img.shape[1:]))).astype(np.float64)   # :363
                    if projector is not None:
                        flat = np.matmul(flat, projector)                                     # :365
                    if cands is None:
                        cands = np.zeros((N, flat.shape[1])).astype(np.float64)               # :358
                    cands[lo:hi, :] = flat
                index = make_index(cands)                                                     # :367-368
                nn_idx, nn_dist = [], []
                while len(nn_idx) != data_size:                                               # :374
                    R, _ = training_set_rec.get_minibatch_np(mb * 2)                          # :376
                    R = R.astype(np.float32)
                    q = np.reshape(adjust_dynamic_range(R, training_set.dynamic_range, drange_net), (-1, np.prod(R.shape[1:]))).astype(np.float64)
                    if projector is not None:
                        q = np.matmul(q, projector)                                           # :380
                    if exclusive_retrieved_code:                                              # :382-396
                        ii, dd = index.query(q, num_samples_factor)
                        for r in range(mb * 2):
                            added = False
                            for j in range(num_samples_factor):
                                if ii[r, j] not in nn_idx:
                                    nn_idx.append(ii[r, j]); nn_dist.append(dd[r, j])
                                    added = True
                                    break
                            if not added:
                                nn_idx.append(ii[r, 0]); nn_dist.append(dd[r, 0])
                    else:
                        ii, dd = index.query(q, 1)                                            # :398
                        nn_idx += list(ii[:, 0]); nn_dist += list(dd[:, 0])                   # :401-402
                    cursor += mb * 2                                                          # :403
                sel_Z = Z[np.array(nn_idx)]                                                   # :404
                sel_d = np.array(nn_dist)                                                     # :405
                thres = np.percentile(sel_d, dist_thres_percentile)                           # :406
                log['refresh_at'].append(cur_nimg)
                log['nearest_indices'].append(np.array(nn_idx)); log['nearest_dists'].append(sel_d)

            fresh = rem_R is None or cursor % data_size == 0                                  # :409-411
            cR = None if fresh else np.array(rem_R)
            cL = None if fresh else np.array(rem_L)
            cZ = None if fresh else np.array(rem_Z)
            while cR is None or cR.shape[0] < mb * 2:                                         # :412
                tR, tL = training_set_rec.get_minibatch_np(mb * 2)
                tR = tR.astype(np.float32)
                pos = cursor % data_size
                tZ = sel_Z[pos:pos + mb * 2]                                                  # :415
                if attr_interesting is None:
                    keep = sel_d[pos:pos + mb * 2] <= thres                                   # :417
                else:
                    active = np.ones(tL.shape[0])
                    for attr in attr_interesting.split(','):
                        active *= tL[:, attr_names.index(attr)]
                    keep = active == 1                                                        # :419-424
                sR, sL, sZ = tR[keep], tL[keep], tZ[keep]
                restart = cR is None or cursor % data_size == 0
                cR = np.array(sR) if restart else np.concatenate((cR, sR), axis=0)            # :428
                restart = cL is None or cursor % data_size == 0
                cL = np.array(sL) if restart else np.concatenate((cL, sL), axis=0)            # :429
                restart = cZ is None or cursor % data_size == 0
                cZ = np.array(sZ) if restart else np.concatenate((cZ, sZ), axis=0)            # :430
                if cR.shape[0] > mb * 2:                                                      # :431-438
                    rem_R = np.array(cR[mb * 2:]); cR = np.array(cR[:mb * 2])
                    rem_L = np.array(cL[mb * 2:]); cL = np.array(cL[:mb * 2])
                    rem_Z = np.array(cZ[mb * 2:]); cZ = np.array(cZ[:mb * 2])
                else:
                    rem_R = rem_L = rem_Z = None
                if cursor % data_size == 0:
                    beginning = True                                                          # :439-440
                cursor += mb * 2                                                              # :441

            R1, L1, R2, L2 = cR[:mb], cL[:mb], cR[mb:], cL[mb:]                               # :443-446
            cZ = slerp_np(cZ, np.random.randn(*cZ.shape).astype(np.float32), knn_perturb_factor)   # :447
            Z1, Z2 = cZ[:mb], cZ[mb:]
            if beginning:
                beginning = False                                                             # :450-454 (snapshot copies)
            order = np.arange(mb)
            np.random.shuffle(order)                                                          # :456-457
            o1 = order.copy()
            R1, L1, Z1 = R1[order], L1[order], Z1[order]
            np.random.shuffle(order)                                                          # :461 (the same array)
            R2, L2, Z2 = R2[order], L2[order], Z2[order]
            step(dict(reals_rec_1=R1, labels_rec_1=L1, latents_rec_1=Z1, reals_rec_2=R2, labels_rec_2=L2, latents_rec_2=Z2,
                      order_1=o1, order_2=order.copy(), cur_nimg=cur_nimg))                    # :466-479
            cur_nimg += mb * 2                                                                # :481
    log.update(final_cursor=cursor, final_staleness=init_staleness)
    return log


def process_reals(x, labels, lod, mirror_augment, drange_data, drange_net, coin=None):
    """training_loop.py:40-60 in NumPy: cast to float32 (:42), dynamic range (:43-44), random mirror (:45-49: a per-image uniform
    draw `coin` < 0.5 keeps the image, otherwise it is reversed along W -- tf.where(coin < 0.5, x, reverse(x, [3]))), then the
    level-of-detail fade (:50-57: lerp towards the 2x2 box-filtered image by frac(lod)) and upscale (:58-59: every pixel repeated
    2^floor(lod) times along H and W), which are the identity at lod = 0 -- the only value configs e/f produce.
    PINNED: tests/golden/ref_train_golden.npz `preals_*` (the reference's own function executed under np_tf)."""
    x = np.asarray(x).astype(np.float32)
    x = adjust_dynamic_range(x, list(drange_data), list(drange_net))
    if mirror_augment:
        coin = np.asarray(coin, np.float32).reshape(-1, 1, 1, 1)
        x = np.where(coin < 0.5, x, x[:, :, :, ::-1])
    n, c, h, w = x.shape
    y = x.reshape(n, c, h // 2, 2, w // 2, 2).mean(axis=(3, 5), keepdims=True)                     # :52-53
    y = np.tile(y, [1, 1, 1, 2, 1, 2]).reshape(n, c, h, w)                                        # :54-55
    x = x + (y - x) * np.float32(lod - np.floor(lod))                                             # :56
    factor = int(2 ** np.floor(lod))                                                              # :59
    x = np.tile(x.reshape(n, c, h, 1, w, 1), [1, 1, 1, factor, 1, factor]).reshape(n, c, h * factor, w * factor)   # :60-62
    return x, labels


def lazy_regularization_args(lrate, beta1, beta2, reg_interval, lazy_regularization):
    """:244-251 -> (learning rate, beta1, beta2) the main AND the regularisation optimizer of a network run with.
    PINNED: `setup_*` of ref_train_golden.npz."""
    if not lazy_regularization:
        return lrate, beta1, beta2
    mb_ratio = reg_interval / (reg_interval + 1)
    return lrate * mb_ratio, beta1 ** mb_ratio, beta2 ** mb_ratio


def smoothing_beta(minibatch_size, G_smoothing_kimg):
    """:222.  PINNED: `gs_beta` of ref_train_golden.npz."""
    return 0.5 ** (minibatch_size / (G_smoothing_kimg * 1000.0)) if G_smoothing_kimg > 0.0 else 0.0


def registered_objectives(G_loss, G_reg, D_loss, D_reg, G_reg_interval, D_reg_interval, lazy_regularization):
    """:283-291 -> the scalars each optimizer differentiates: {'TrainG': [...], 'RegG': [...], 'TrainD': [...], 'RegD': [...]}."""
    out = dict(TrainG=[], RegG=[], TrainD=[], RegD=[])
    if not lazy_regularization:
        G_loss = G_loss + G_reg
        D_loss = D_loss + D_reg
    else:
        out['RegG'].append(np.mean(G_reg * G_reg_interval))
        out['RegD'].append(np.mean(D_reg * D_reg_interval))
    out['TrainG'].append(np.mean(G_loss))
    out['TrainD'].append(np.mean(D_loss))
    return out
