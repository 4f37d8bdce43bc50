"""Second-order FM prediction on the GPU (reference: forward.py:14-22).

``FmModel.fma(X)`` is the reference's ``fma(x)``: X is a scipy.sparse CSR design matrix
(fm.py:61-93), the result ``mu + X.W + 0.5 * (||X V||^2 - (X*X).(V*V).1)`` per row.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class FmModel(object):
    def __init__(self, n_features, dim, device=0, loss="nll", optimizer="sgd", lr=0.01, reg=0.0):
        self._lib = L.load()
        self._h = L._p()
        self.n_features, self.dim = int(n_features), int(dim)
        o = L.TfrOpts()
        self._lib.tfr_default_opts(C.byref(o))
        o.loss, o.optimizer, o.adam_mode = L.LOSS[loss], L.OPTIMIZER[optimizer], L.ADAM_MODE["lazy"]
        o.device, o.lr, o.reg = int(device), lr, reg
        self._check(self._lib.tfr_fm_create(C.byref(self._h), self.n_features, self.dim, C.byref(o)))

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

    def get(self):
        mu = np.empty(1, np.float32)
        W = np.empty(self.n_features, np.float32)
        V = np.empty(self.n_features * self.dim, np.float32)
        self._check(self._lib.tfr_fm_get(self._h, L.ptr_f32(mu), L.ptr_f32(W), L.ptr_f32(V)))
        return float(mu[0]), W, V.reshape(self.n_features, self.dim)

    def train_step(self, x, y):
        """One minibatch of SGD / lazy-Adam training on CSR rows ``x`` with targets ``y``.
        Returns (predictions before the update, data loss)."""
        x = x.tocsr()
        indptr = np.ascontiguousarray(x.indptr, np.int64)
        indices = np.ascontiguousarray(x.indices, np.int32)
        data = np.ascontiguousarray(x.data, np.float32)
        y = L.as_f32(y)
        n = indptr.size - 1
        if y.shape != (n,) or x.shape[1] != self.n_features:
            raise ValueError("x must be [n, %d] and y [n]" % self.n_features)
        pred, loss = np.empty(n, np.float32), C.c_float()
        self._check(self._lib.tfr_fm_train_step(self._h, L.ptr_i64(indptr), L.ptr_i32(indices), L.ptr_f32(data),
                                                L.ptr_f32(y), n, L.ptr_f32(pred), C.byref(loss)))
        return pred, loss.value

    def train_step_dev(self, d_indptr, d_indices, d_data, d_y, n_rows, nnz, d_pred=None):
        self._check(self._lib.tfr_fm_train_step_dev(self._h, d_indptr, d_indices, d_data, d_y, n_rows, nnz, d_pred))

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


AGENTS = ["users", "items", "skills", "attempts", "wins", "fails", "item_wins", "item_fails"]   # dataio.py:67 order


def df_to_sparse(df, user_num, item_num, active_agents, qmatrix=None, skill_wins=None, skill_fails=None):
    """The FM design matrix of fm.py:61-93 as one CSR ``[n_events, sum of block widths]``:

    ``users`` / ``items`` one-hot blocks (fm.py:74-75); ``skills`` = the q-matrix rows of the events'
    items (fm.py:76; identity q-matrix when none is given, fm.py:44-47); ``item_wins`` /
    ``item_fails`` = the item one-hot pattern carrying the event's win / fail counts
    (fm.py:78-81); ``attempts`` / ``wins`` / ``fails`` = per-skill counters supplied as matrices
    (fm.py:83-89).  Blocks are concatenated in the order of ``AGENTS`` restricted to
    ``active_agents`` (dataio.py:63-72, fm.py:91).  Host code (scipy), like the reference."""
    import scipy.sparse as sp
    n = len(df["user"])
    rows = np.arange(n)
    user = np.asarray(df["user"], np.int64)
    item = np.asarray(df["item"], np.int64)
    ones = np.ones(n, np.float32)
    blocks = {"users": sp.coo_matrix((ones, (rows, user)), shape=(n, user_num)),
              "items": sp.coo_matrix((ones, (rows, item)), shape=(n, item_num))}
    q = qmatrix if qmatrix is not None else sp.identity(item_num, dtype=np.float32, format="csr")
    blocks["skills"] = q.tocsr()[item]
    if "wins" in df:
        blocks["item_wins"] = sp.coo_matrix((np.asarray(df["wins"], np.float32), (rows, item)), shape=(n, item_num))
        blocks["item_fails"] = sp.coo_matrix((np.asarray(df["fails"], np.float32), (rows, item)), shape=(n, item_num))
    if skill_wins is not None:
        blocks["attempts"] = skill_wins + skill_fails
        blocks["wins"] = skill_wins
        blocks["fails"] = skill_fails
    chosen = [a for a in AGENTS if a in active_agents]
    missing = [a for a in chosen if a not in blocks]
    if missing:
        raise ValueError("no data for blocks %s" % missing)
    x = sp.hstack([blocks[a] for a in chosen]).tocsr().astype(np.float32)
    x.data = np.nan_to_num(x.data)                     # fm.py:126,130
    return x
