"""Second-order FM prediction on the GPU (reference: forward.py:14-22).

``FmModel.fma(X)`` is the reference's ``fma(x)``: X is a scipy.sparse CSR design matrix
(fm.py:61-93), the result ``mu + X.W + 0.5 * (||X V||^2 - (X*X).(V*V).1)`` per row.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class FmModel(object):
    def __init__(self, n_features, dim, device=0):
        self._lib = L.load()
        self._h = L._p()
        self.n_features, self.dim = int(n_features), int(dim)
        self._check(self._lib.tfr_fm_create(C.byref(self._h), self.n_features, self.dim, int(device)))

    def _check(self, rc):
        if rc != L.OK:
            text = self._lib.tfr_fm_last_error().decode("utf-8", "replace")
            raise (L.OutOfRangeError if rc == L.ERR_OOB else L.TfrError)(rc, text)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tfr_fm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set(self, mu, W, V):
        """the bundle fm_mangaki.py:39-45 pickles: {'mu', 'W', 'V'}"""
        W, V = L.as_f32(W).reshape(-1), L.as_f32(V)
        if W.size != self.n_features or V.shape != (self.n_features, self.dim):
            raise ValueError("W must be [%d], V [%d, %d]" % (self.n_features, self.n_features, self.dim))
        self._check(self._lib.tfr_fm_set(self._h, float(mu), L.ptr_f32(W), L.ptr_f32(V.reshape(-1))))

    def init(self, seed=0, stddev=0.1):
        self._check(self._lib.tfr_fm_init(self._h, int(seed), stddev))

    def forward_csr(self, indptr, indices, data):
        indptr = np.ascontiguousarray(indptr, np.int64)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float32)
        n = indptr.size - 1
        out = np.empty(n, np.float32)
        self._check(self._lib.tfr_fm_forward(self._h, L.ptr_i64(indptr), L.ptr_i32(indices), L.ptr_f32(data), n,
                                             L.ptr_f32(out)))
        return out

    def fma(self, x):
        """forward.py:21-22 on a scipy.sparse matrix (any format; converted to CSR)."""
        x = x.tocsr()
        if x.shape[1] != self.n_features:
            raise ValueError("X has %d columns, the model %d features" % (x.shape[1], self.n_features))
        return self.forward_csr(x.indptr, x.indices, x.data)

    def forward_dev(self, d_indptr, d_indices, d_data, n_rows, d_out):
        self._check(self._lib.tfr_fm_forward_dev(self._h, d_indptr, d_indices, d_data, n_rows, d_out))

    def sync(self):
        ms = C.c_float()
        self._check(self._lib.tfr_fm_sync(self._h, C.byref(ms)))
        return ms.value
