"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ctypes driver for the REFERENCE's DCI library (oracle/_ref/libdci_ref.so, built by oracle/Makefile
from /root/reference/dci_code/src/{dci.c,util.c}).  Mirrors the call sequence of the reference's
Python wrapper (dci_code/src/dci.py:230-330 -> py_dci.c -> dci.h:76-89) with the training-time
parameters (training/training_loop.py:197,368,398).  Used to pin oracle/nn.py and as the CPU
baseline of the nearest-neighbour stage.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_ref', 'libdci_ref.so')


class _Dci(ctypes.Structure):            # dci.h:51-63
    _fields_ = [('dim', ctypes.c_int), ('num_comp_indices', ctypes.c_int), ('num_simp_indices', ctypes.c_int),
                ('num_points', ctypes.c_int), ('num_levels', ctypes.c_int), ('num_coarse_points', ctypes.c_int),
                ('indices', ctypes.c_void_p), ('proj_vec', ctypes.c_void_p), ('data', ctypes.c_void_p),
                ('next_level_ranges', ctypes.c_void_p), ('num_finest_level_points', ctypes.c_void_p)]


class _QueryConfig(ctypes.Structure):    # dci.h:68-78
    _fields_ = [('blind', ctypes.c_bool), ('num_to_visit', ctypes.c_int), ('num_to_retrieve', ctypes.c_int),
                ('prop_to_visit', ctypes.c_double), ('prop_to_retrieve', ctypes.c_double),
                ('field_of_view', ctypes.c_int), ('min_num_finest_level_points', ctypes.c_int)]


def available():
    return os.path.isfile(LIB_PATH)


class DCIRef:
    def __init__(self, dim, num_comp_indices=3, num_simp_indices=15):
        self.lib = ctypes.CDLL(LIB_PATH)
        self.lib.dci_init.argtypes = [ctypes.POINTER(_Dci), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.lib.dci_add.argtypes = [ctypes.POINTER(_Dci), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, _QueryConfig]
        self.lib.dci_query.argtypes = [ctypes.POINTER(_Dci), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, _QueryConfig,
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_int)), ctypes.POINTER(ctypes.POINTER(ctypes.c_double)),
                                       ctypes.POINTER(ctypes.c_int)]
        self.lib.dci_free.argtypes = [ctypes.POINTER(_Dci)]
        self.inst = _Dci()
        self.dim = dim
        self.lib.dci_init(ctypes.byref(self.inst), dim, num_comp_indices, num_simp_indices)
        self._data = None
        self._libc = ctypes.CDLL(None)
        self._libc.free.argtypes = [ctypes.c_void_p]

    def add(self, data, num_levels=3, field_of_view=10, prop_to_retrieve=0.002):
        """dci.py:230-271 defaults: blind False, num_to_visit -1, prop_to_visit 1.0."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert data.shape[1] == self.dim
        self._data = data  # the index borrows the buffer (dci.h:78)
        cfg = _QueryConfig(False, -1, -1, 1.0, prop_to_retrieve, field_of_view if num_levels >= 3 else -1, 0)
        self.lib.dci_add(ctypes.byref(self.inst), self.dim, data.shape[0], data.ctypes.data, num_levels, cfg)

    def query(self, query, num_neighbours=1, field_of_view=200, prop_to_retrieve=1.0):
        """dci.py:273-330 -> (idx int32 [nq, k], dist float64 [nq, k])."""
        query = np.ascontiguousarray(query, dtype=np.float64)
        nq = query.shape[0]
        cfg = _QueryConfig(False, -1, -1, 1.0, prop_to_retrieve, field_of_view if self.inst.num_levels >= 2 else -1, 0)
        nn = (ctypes.POINTER(ctypes.c_int) * nq)()
        nd = (ctypes.POINTER(ctypes.c_double) * nq)()
        num = (ctypes.c_int * nq)()
        self.lib.dci_query(ctypes.byref(self.inst), self.dim, nq, query.ctypes.data, num_neighbours, cfg, nn, nd, num)
        idx = np.empty((nq, num_neighbours), dtype=np.int32)
        dist = np.empty((nq, num_neighbours), dtype=np.float64)
        for i in range(nq):
            assert num[i] >= num_neighbours
            for j in range(num_neighbours):
                idx[i, j] = nn[i][j]
                dist[i, j] = nd[i][j]
            self._libc.free(ctypes.cast(nn[i], ctypes.c_void_p))
            self._libc.free(ctypes.cast(nd[i], ctypes.c_void_p))
        return idx, dist

    def close(self):
        self.lib.dci_free(ctypes.byref(self.inst))
