"""``MangakiALS3`` with the reference's interface (als3.py:9-143), fitted on the GPU in float64.

    als = MangakiALS3(nb_components=20, nb_iterations=20, lambda_=0.1)
    als.nb_users, als.nb_works = U, W          # set by the caller, as forward.py:32-33 does
    als.fit(X_train, y_train, y_test, X_test)  # prints 'Step k rmse' per iteration (als3.py:31)
    als.predict(X)

``init_vars`` draws U, V, W_user, W_work from NumPy's global legacy RNG in the reference's order
(als3.py:57-65), so a seeded run reproduces the reference's trajectory.  pickle save/load of the
reference (als3.py:122-137) is replaced by ``state()`` / ``load_state()`` (plain arrays).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class MangakiALS3(object):
    def __init__(self, nb_components=20, nb_iterations=20, lambda_=0.1, device=0, verbose=True):
        self.nb_components, self.nb_iterations, self.lambda_ = nb_components, nb_iterations, lambda_
        self.device, self.verbose = device, verbose
        self._h = None
        self._lib = L.load()

    @property
    def is_serializable(self):
        return True

    def _check(self, rc):
        if rc != L.OK:
            text = self._lib.tfr_als_last_error().decode("utf-8", "replace")
            raise (L.OutOfRangeError if rc == L.ERR_OOB else L.TfrError)(rc, text)

    def close(self):
        if self._h:
            self._lib.tfr_als_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _f64(a):
        return np.ascontiguousarray(a, np.float64)

    @staticmethod
    def _p64(a):
        return a.ctypes.data_as(L._f64p)

    def init_vars(self):
        """als3.py:57-65 - same draw order from the global legacy RNG; bias = 0."""
        U = np.random.rand(self.nb_users, self.nb_components)
        V = np.random.rand(self.nb_works, self.nb_components)
        W_user = np.random.rand(self.nb_users)
        W_work = np.random.rand(self.nb_works)
        self.close()
        self._h = L._p()
        self._check(self._lib.tfr_als_create(C.byref(self._h), self.nb_users, self.nb_works, self.nb_components,
                                             float(self.lambda_), int(self.device)))
        self.load_state(dict(U=U, V=V, W_user=W_user, W_work=W_work, bias=0.0))

    def load_state(self, st):
        U, V, Wu, Ww = (self._f64(st[k]) for k in ("U", "V", "W_user", "W_work"))
        self._check(self._lib.tfr_als_set(self._h, self._p64(U), self._p64(V), self._p64(Wu), self._p64(Ww)))
        self._check(self._lib.tfr_als_set_bias(self._h, float(st["bias"])))
        self.bias = float(st["bias"])

    def state(self):
        U = np.empty((self.nb_users, self.nb_components))
        V = np.empty((self.nb_works, self.nb_components))
        Wu, Ww, b = np.empty(self.nb_users), np.empty(self.nb_works), C.c_double()
        self._check(self._lib.tfr_als_get(self._h, self._p64(U), self._p64(V), self._p64(Wu), self._p64(Ww), C.byref(b)))
        return dict(U=U, V=V, W_user=Wu, W_work=Ww, bias=b.value)

    U = property(lambda self: self.state()["U"])
    V = property(lambda self: self.state()["V"])
    W_user = property(lambda self: self.state()["W_user"])
    W_work = property(lambda self: self.state()["W_work"])

    def fit(self, X, y, y_test, X_test):
        """als3.py:20-35."""
        self.X_test, self.y_test = X_test, y_test
        self.init_vars()
        X = np.asarray(X, np.int64)
        y = self._f64(y)
        u, w = np.ascontiguousarray(X[:, 0]), np.ascontiguousarray(X[:, 1])
        self._check(self._lib.tfr_als_load(self._h, L.ptr_i64(u), L.ptr_i64(w), self._p64(y), y.size))
        self.bias = float(y.mean())                                   # als3.py:24 (NumPy's own mean: bit-identical)
        self._check(self._lib.tfr_als_set_bias(self._h, self.bias))
        self.sweep_ms = 0.0
        for nb_iter in range(self.nb_iterations):
            if self.verbose:
                print('Step', nb_iter, self.compute_rmse(self.y_test, self.predict(self.X_test)))
            ms = C.c_float()
            self._check(self._lib.tfr_als_sweep(self._h, 1, C.byref(ms)))
            self.sweep_ms += ms.value

    def predict(self, X):
        """als3.py:110-113 at the given (user, work) pairs."""
        X = np.asarray(X, np.int64)
        u, w = np.ascontiguousarray(X[:, 0]), np.ascontiguousarray(X[:, 1])
        out = np.empty(u.size, np.float64)
        self._check(self._lib.tfr_als_predict(self._h, L.ptr_i64(u), L.ptr_i64(w), u.size, self._p64(out)))
        return out

    def get_shortname(self):
        return 'als3-%d' % self.nb_components

    @staticmethod
    def compute_rmse(y_pred, y_true):
        """als3.py:118-120 (argument order as in the reference; the metric is symmetric)."""
        return float(np.sqrt(np.mean((np.asarray(y_true, np.float64) - np.asarray(y_pred, np.float64)) ** 2)))

    def compute_all_errors(self, X_train, y_train, X_test, y_test):
        print('Train RMSE=%f' % self.compute_rmse(y_train, self.predict(X_train)))
        print('Test RMSE=%f' % self.compute_rmse(y_test, self.predict(X_test)))
