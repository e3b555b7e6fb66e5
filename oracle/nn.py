"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Exact 1-nearest-neighbour by direct differences in float64 -- what `dci_query(..., num_neighbours=1)`
approximates (dci_code/src/dci.c:788-828) with `compute_dist` (dci_code/src/util.c:62-69: Euclidean,
sqrt of the summed squared differences).  PINNED against the reference's own DCI library built into
oracle/_ref/ (oracle/dci_ref.py) on the golden fixture tests/golden/dci_golden.npz.
"""
import numpy as np


def nearest_neighbour(data, query, chunk=256):
    """data [N,dim], query [nq,dim] (any float) -> (idx int32 [nq], dist float64 [nq]); ties -> lowest index."""
    data = np.asarray(data, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    idx = np.empty(query.shape[0], dtype=np.int32)
    dist = np.empty(query.shape[0], dtype=np.float64)
    for i in range(0, query.shape[0], chunk):
        q = query[i:i + chunk]
        d2 = np.empty((q.shape[0], data.shape[0]), dtype=np.float64)
        for j in range(q.shape[0]):
            diff = data - q[j]
            d2[j] = np.einsum('ij,ij->i', diff, diff)
        a = np.argmin(d2, axis=1)
        idx[i:i + chunk] = a
        dist[i:i + chunk] = np.sqrt(d2[np.arange(q.shape[0]), a])
    return idx, dist
