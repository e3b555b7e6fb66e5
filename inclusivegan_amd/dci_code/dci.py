"""`DCI`: the nearest-neighbour index surface the training loop uses
(`from dci import DCI`, training/training_loop.py:21-23,197,367-368,398), backed by the exact
on-GPU streaming 1-NN instead of the reference's CPU Prioritized-DCI index.

Kept from the reference's `dci_code/src/dci.py:61-340`:
  * `DCI(dim, num_comp_indices, num_simp_indices)`; `add(data, ...)` (one array only, :261-262),
    `query(query, num_neighbours, ...)` returning `(list of int32 index arrays, list of float64
    distance arrays)` per query (:316-330), `reset()`, `clear()`, `num_points`, `dim`;
  * distances are Euclidean (sqrt) like `compute_dist` (dci_code/src/util.c:62-69);
  * NumPy input must be float64, C-contiguous, of the declared dimension (:113-121).
The indexing hyper-parameters (num_levels, field_of_view, prop_to_retrieve, blind, ...) are accepted
and ignored: the search is exact, so every reference setting maps to the same (best possible)
answer.  Extension: `add` / `query` also accept device tensors, so candidates generated on the GPU
never visit the host (the reference materialises a 118 GB fp64 array, training_loop.py:358).

`num_neighbours == 1` (the default, non-exclusive IMLE assignment, training_loop.py:398) runs entirely in the streaming
HIP kernel; `num_neighbours > 1` (the exclusive variant, :382-396) screens with the same MFMA products and re-ranks a
short list exactly with torch ops.
"""
import numpy as np
import torch

from .. import hip_ops


class DCI(object):
    def __init__(self, dim, num_comp_indices=2, num_simp_indices=7, device=None):
        self._dim = int(dim)
        self._num_comp_indices = num_comp_indices
        self._num_simp_indices = num_simp_indices
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError('inclusivegan_amd DCI needs a ROCm device; there is no CPU path')
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self._data = None
        self._norms = None
        # The kernels address an operand through 32-bit byte offsets (include/igan_hip.h: below 2 GiB each; igan_conv2d / igan_nn1_update reject more),
        # so at large unprojected dimensions (256x256x3 = 196 608) a pass takes fewer rows instead of failing.
        fit = max(1, 0x7FFFFFF0 // (4 * max(self._dim, 1)) - 1)
        self.cand_chunk = min(8192, fit)      # candidates folded per kernel pass
        self.query_chunk = min(4096, fit)     # queries per kernel pass
        self.rerank_bytes = 1 << 30 # largest fp64 block the exact re-rank of query_device_k gathers at once

    @property
    def dim(self):
        return self._dim

    @property
    def num_comp_indices(self):
        return self._num_comp_indices

    @property
    def num_simp_indices(self):
        return self._num_simp_indices

    @property
    def num_points(self):
        return 0 if self._data is None else int(self._data.shape[0])

    @property
    def num_levels(self):
        return 0 if self._data is None else 1

    @property
    def proj_vec(self):
        """The reference exposes its random projection directions (dci.py:93-105); the exact search keeps none."""
        raise AttributeError('this DCI is an exact on-GPU 1-NN search: it has no projection vectors')

    def _check_numpy(self, arr):
        if arr.ndim != 2 or arr.shape[1] != self.dim:
            raise ValueError('mismatch between array dimension (%s) and the declared dimension of this DCI instance (%d)' % (arr.shape[1:] , self.dim))
        if arr.dtype != np.float64:
            raise TypeError('array must consist of double-precision floats')
        if not arr.flags.c_contiguous:
            raise ValueError('the memory layout of array must be in row-major (C-order)')

    def _to_device(self, arr, check):
        if torch.is_tensor(arr):
            if arr.dim() != 2 or arr.shape[1] != self.dim:
                raise ValueError('mismatch between array dimension (%s) and the declared dimension of this DCI instance (%d)' % (tuple(arr.shape[1:]), self.dim))
            return arr.to(self.device, torch.float32).contiguous()
        arr = np.asarray(arr)
        if check:
            self._check_numpy(arr)
        elif arr.ndim != 2 or arr.shape[1] != self.dim:
            raise ValueError('mismatch between array dimension (%s) and the declared dimension of this DCI instance (%d)' % (arr.shape[1:], self.dim))
        out = torch.empty((arr.shape[0], self.dim), device=self.device, dtype=torch.float32)
        step = max(1, (64 << 20) // (8 * self.dim))
        for i in range(0, arr.shape[0], step):
            out[i:i + step] = torch.from_numpy(np.ascontiguousarray(arr[i:i + step], dtype=np.float32)).to(self.device)
        return out

    def add(self, data, indices=None, num_levels=2, field_of_view=10, blind=False, num_to_visit=-1, num_to_retrieve=-1,
            prop_to_visit=-1.0, prop_to_retrieve=-1.0):
        if self.num_points > 0:
            raise RuntimeError('DCI class does not support insertion of more than one array. Must combine all arrays into one array before inserting')
        if indices is not None:
            raise NotImplementedError('DCI.add(indices=...) is not built on the hip path')
        self._data = self._to_device(data, check=True)
        self._norms = hip_ops.row_sqnorm_raw(self._data)

    def query(self, query, num_neighbours=-1, field_of_view=100, blind=False, num_to_visit=-1, num_to_retrieve=-1,
              prop_to_visit=-1.0, prop_to_retrieve=-1.0):
        if self._data is None:
            raise RuntimeError('DCI.query() on an empty database')
        if num_neighbours < 0:
            num_neighbours = self.num_points
        num_neighbours = min(num_neighbours, self.num_points)
        q = self._to_device(query, check=False)
        if num_neighbours == 1:
            idx, dist = self.query_device(q)
            idx = idx.cpu().numpy().astype(np.int32)
            dist = dist.cpu().numpy().astype(np.float64)
            return [idx[i:i + 1] for i in range(idx.shape[0])], [dist[i:i + 1] for i in range(dist.shape[0])]
        idx, dist = self.query_device_k(q, num_neighbours)
        idx = idx.cpu().numpy().astype(np.int32)
        dist = dist.cpu().numpy().astype(np.float64)
        return [idx[i] for i in range(idx.shape[0])], [dist[i] for i in range(dist.shape[0])]

    def query_device(self, q):
        """q: device fp32 [nq, dim] -> (int64 idx [nq], fp64 Euclidean dist [nq]) on the device."""
        nq = int(q.shape[0])
        best_d2, best_idx = hip_ops.nn1_state(nq, self.device)
        n = self.num_points
        for q0 in range(0, nq, self.query_chunk):
            qs = q[q0:q0 + self.query_chunk]
            qn = hip_ops.row_sqnorm_raw(qs)
            for c0 in range(0, n, self.cand_chunk):
                hip_ops.nn1_update_raw(qs, qn, self._data[c0:c0 + self.cand_chunk], self._norms[c0:c0 + self.cand_chunk],
                                       best_d2[q0:q0 + self.query_chunk], best_idx[q0:q0 + self.query_chunk], c0)
        return unpack_best(best_d2, best_idx)

    def _screen(self, qs, qn, keep):
        """Top-`keep` candidates per query by the fp32 MFMA screening distance over all candidate batches -> (d2 [nq, keep] ascending, idx)."""
        n = self.num_points
        best_v = best_i = None
        for c0 in range(0, n, self.cand_chunk):
            cs = self._data[c0:c0 + self.cand_chunk]
            dots = hip_ops.conv2d_raw(qs.reshape(qs.shape[0], self.dim, 1, 1), cs.reshape(1, 1, cs.shape[0], self.dim),
                                      hip_ops.ConvGeom(1, 1, 1, 1, 0, 0), (1, 1), cs.shape[0], w_transposed=True).reshape(qs.shape[0], -1)
            d2 = qn[:, None] + self._norms[c0:c0 + self.cand_chunk].double()[None, :] - 2.0 * dots.double()
            d2 = torch.where(torch.isfinite(d2), d2, torch.full_like(d2, float('inf')))
            ci = torch.arange(c0, c0 + cs.shape[0], device=self.device)[None, :].expand_as(d2)
            if best_v is not None:
                d2 = torch.cat([best_v, d2], dim=1)
                ci = torch.cat([best_i, ci], dim=1)
            v, j = torch.topk(d2, min(keep, d2.shape[1]), dim=1, largest=False)
            best_v, best_i = v, torch.gather(ci, 1, j)
        return best_v, best_i

    def _rerank(self, qs, cand_i, k, rerank_chunk):
        """Exact fp64 squared distances (direct differences, compute_dist of dci_code/src/util.c:62-69) of the listed candidates,
        ordered by (distance, index) -> (e [nq, k], idx [nq, k]).  The query rows per pass are sized so that the gathered
        [rows, keep, dim] fp64 block stays inside `self.rerank_bytes`."""
        keep = int(cand_i.shape[1])
        rows = max(1, min(rerank_chunk, self.rerank_bytes // max(1, keep * self.dim * 8)))
        es, iis = [], []
        for r0 in range(0, qs.shape[0], rows):
            qq = qs[r0:r0 + rows].double()
            ii = cand_i[r0:r0 + rows]
            if keep * self.dim * 8 > self.rerank_bytes:          # one row does not fit either: walk the candidates in slabs
                step = max(1, self.rerank_bytes // (self.dim * 8))
                e = torch.cat([((self._data[ii[0, c0:c0 + step]].double() - qq[0][None, :]) ** 2).sum(dim=1) for c0 in range(0, keep, step)])[None, :]
            else:
                diff = self._data[ii.reshape(-1)].double().reshape(ii.shape[0], ii.shape[1], self.dim) - qq[:, None, :]
                e = (diff * diff).sum(dim=2)
            e = torch.where(torch.isfinite(e), e, torch.full_like(e, float('inf')))
            order = torch.argsort(ii, dim=1, stable=True)                       # lower index first among equal distances
            e, ii = torch.gather(e, 1, order), torch.gather(ii, 1, order)
            order = torch.argsort(e, dim=1, stable=True)
            es.append(torch.gather(e, 1, order)[:, :k]); iis.append(torch.gather(ii, 1, order)[:, :k])
        return torch.cat(es), torch.cat(iis)

    def query_device_k(self, q, k, margin=8, rerank_chunk=64, max_keep=1024):
        """k > 1 neighbours (the exclusive IMLE assignment asks for num_samples_factor of them, training_loop.py:386):
        (int64 idx [nq, k], fp64 Euclidean dist [nq, k]), ascending, ties to the lower index.  Screening on the fp32 MFMA
        products keeps the k + margin best candidates per query over all candidate batches; those are then measured exactly
        (direct differences in fp64) and re-ranked by torch ops (a non-default path: the reference's default is k = 1).

        The short list is CHECKED before it is trusted: every candidate that was screened out has a screening distance of at
        least the worst kept one, hence an exact one of at least that minus the screening error tol * (|q|^2 + max |c|^2).  `tol`
        is the 1-NN kernel's nn1_tol = 2^-22 sqrt(dim): a STATISTICAL bound on the rounding error of a dim-long fp32 dot product
        (random-walk growth, ~4 standard deviations), not the worst case dim * 2^-24 -- the same assumption the 1-NN screening
        makes.  Queries whose check fails are searched again with four times the margin (only those queries; the margin always
        grows).  Past `max_keep` kept candidates -- near-tied candidates, e.g. a collapsed generator -- the list is no longer
        widened: the exact k-th distance already found bounds the true one from above, so only candidates whose screening
        distance lies within the slack of it can matter; they are selected by threshold and measured exactly in slabs, which
        terminates for any input and keeps the fp64 blocks inside `rerank_bytes`."""
        nq = int(q.shape[0])
        n = self.num_points
        out_i = torch.empty((nq, k), device=self.device, dtype=torch.int64)
        out_d = torch.empty((nq, k), device=self.device, dtype=torch.float64)
        tol = max(2.0 ** -22 * float(np.sqrt(self.dim)), 1e-6)
        cmax = float(self._norms.max()) if n else 0.0
        self.last_margins = []
        self.last_threshold_queries = 0
        for q0 in range(0, nq, self.query_chunk):
            qs_all = q[q0:q0 + self.query_chunk]
            todo = torch.arange(qs_all.shape[0], device=self.device)
            m = max(int(margin), 0)
            while todo.numel():
                qs = qs_all[todo]
                qn = hip_ops.row_sqnorm_raw(qs).double()
                keep = min(n, k + m)
                best_v, best_i = self._screen(qs, qn, keep)
                e_k, i_k = self._rerank(qs, best_i, k, rerank_chunk)
                # sufficiency: no screened-out candidate can be closer than the k-th exact distance
                slack = tol * (qn + cmax)
                ok = (best_v[:, -1] - slack > e_k[:, -1]) if keep < n else torch.ones_like(qn, dtype=torch.bool)
                done = todo[ok]
                out_i[q0 + done] = i_k[ok]
                out_d[q0 + done] = torch.sqrt(e_k[ok])
                self.last_margins.append(m)
                bad = ~ok
                if not bool(bad.any()):
                    break
                todo = todo[bad]
                m = max(4 * m, 8)
                if k + m > max_keep:         # stop widening: threshold selection against the exact k-th distance found so far
                    self._threshold_rerank(qs_all, todo, q0, e_k[bad][:, -1] + slack[bad], k, out_i, out_d, i_k[bad])
                    self.last_threshold_queries += int(todo.numel())
                    break
        return out_i, out_d

    def _threshold_rerank(self, qs_all, todo, q0, limit, k, out_i, out_d, short):
        """For each listed query: every candidate whose screening distance is <= limit (the exact k-th distance of a valid short
        list plus the screening slack -- the true k nearest are among them), measured exactly slab by slab, best k kept.  `short`
        [len(todo), k]: the short list the limit came from; its members are always measured too, so that at least k candidates are (the
        one-row screening product below may round differently from the batched one that chose them: ADVICE r04)."""
        n = self.num_points
        step = max(1, self.rerank_bytes // (self.dim * 8))
        for row, (t, lim) in enumerate(zip(todo.tolist(), limit.tolist())):
            qrow = qs_all[t:t + 1]
            qn = hip_ops.row_sqnorm_raw(qrow).double()
            sel = []
            for c0 in range(0, n, self.cand_chunk):
                cs = self._data[c0:c0 + self.cand_chunk]
                dots = hip_ops.conv2d_raw(qrow.reshape(1, self.dim, 1, 1), cs.reshape(1, 1, cs.shape[0], self.dim),
                                          hip_ops.ConvGeom(1, 1, 1, 1, 0, 0), (1, 1), cs.shape[0], w_transposed=True).reshape(-1)
                d2 = qn[0] + self._norms[c0:c0 + self.cand_chunk].double() - 2.0 * dots.double()
                sel.append(torch.nonzero(~(d2 > lim)).reshape(-1) + c0)        # a NaN screening value stays in (measured exactly below); +inf is out -- unless the short list holds it
            ii = torch.unique(torch.cat(sel + [short[row].to(torch.int64).reshape(-1)]))      # sorted ascending: ties go to the lower index
            qq = qrow[0].double()
            e = torch.cat([((self._data[ii[c0:c0 + step]].double() - qq[None, :]) ** 2).sum(dim=1) for c0 in range(0, ii.numel(), step)])
            e = torch.where(torch.isfinite(e), e, torch.full_like(e, float('inf')))
            order = torch.argsort(e, stable=True)[:k]                          # ii ascending already: ties go to the lower index
            out_i[q0 + t] = ii[order]
            out_d[q0 + t] = torch.sqrt(e[order])

    def clear(self):
        self._data = None
        self._norms = None

    def reset(self):
        # The reference also re-draws its random projection directions here (dci.c:859-863); an exact
        # search has none.
        self.clear()


def unpack_best(best_d2, best_idx):
    """running minimum (fp64 squared distance, int32 index) -> (idx int64, Euclidean distance fp64)."""
    return best_idx.to(torch.int64), torch.sqrt(best_d2)
