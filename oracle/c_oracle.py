"""ctypes wrapper of oracle/libsvd_oracle.so (svd_oracle.c).  TEST INFRASTRUCTURE ONLY:
tests/ and bench.py's cpu_baseline leg.  Same caveat as svd_oracle.py: parity unpinned by
reference fixtures."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libsvd_oracle.so")
_LOSS = {"mse": 0, "nll": 1}
_OPT = {"adam": 0, "sgd": 1}
_MODE = {"tf1": 0, "lazy": 1}
_i32p, _f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("%s not built: run `make -C oracle` (or __graft_entry__.build())" % _PATH)
        lib = C.CDLL(_PATH)
        lib.svdo_create.restype = C.c_void_p
        lib.svdo_create.argtypes = [C.c_int64, C.c_int64] + [C.c_int] * 6 + [C.c_float] * 5
        lib.svdo_destroy.argtypes = [C.c_void_p]
        lib.svdo_table.restype = _f32p
        lib.svdo_table.argtypes = [C.c_void_p, C.c_int]
        lib.svdo_set_frozen.argtypes = [C.c_void_p, C.c_uint32]
        lib.svdo_step.restype = C.c_int64
        lib.svdo_step.argtypes = [C.c_void_p]
        lib.svdo_threads.restype = C.c_int
        lib.svdo_set_threads.argtypes = [C.c_int]
        lib.svdo_forward.argtypes = [C.c_void_p, _i32p, _i32p, C.c_int64, _f32p]
        lib.svdo_train_step.argtypes = [C.c_void_p, _i32p, _i32p, _f32p, C.c_int64, _f32p, _f32p, _f32p]
        lib.fmo_forward.restype = C.c_int
        lib.fmo_forward.argtypes = [C.c_double, _f32p, _f32p, C.c_int64, C.c_int, C.POINTER(C.c_int64), _i32p, _f32p, C.c_int64,
                                    C.POINTER(C.c_double)]
        _lib = lib
    return _lib


def threads():
    """Threads the OpenMP loops will use (all host cores unless set_threads / OMP_NUM_THREADS say otherwise)."""
    return load().svdo_threads()


def set_threads(n):
    load().svdo_set_threads(int(n))


class COracle:
    def __init__(self, U, I, D, *, loss="mse", item_abs=False, reg_bias=False, optimizer="adam",
                 adam_mode="tf1", lr=1e-3, reg=0.05, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lib = load()
        self.U, self.I, self.D = U, I, D
        self.h = self.lib.svdo_create(U, I, D, _LOSS[loss], int(item_abs), int(reg_bias), _OPT[optimizer],
                                      _MODE[adam_mode], lr, reg, beta1, beta2, eps)
        if not self.h:
            raise MemoryError("svdo_create failed")

    def close(self):
        if self.h:
            self.lib.svdo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _view(self, which):
        t = which & 7
        shape = {0: (1,), 1: (self.U,), 2: (self.I,), 3: (self.U, self.D), 4: (self.I, self.D)}[t]
        p = self.lib.svdo_table(self.h, which)
        return np.ctypeslib.as_array(p, shape=(int(np.prod(shape)),)).reshape(shape)

    def table(self, which):
        return self._view(which).copy()

    def set_tables(self, mu, bu, bi, P, Q):
        for which, val in ((0, mu), (1, bu), (2, bi), (3, P), (4, Q)):
            self._view(which)[...] = np.asarray(val, np.float32).reshape(self._view(which).shape)

    def set_frozen(self, mask):
        self.lib.svdo_set_frozen(self.h, mask)

    @property
    def step(self):
        return self.lib.svdo_step(self.h)

    def forward(self, u, i):
        u, i = np.ascontiguousarray(u, np.int32), np.ascontiguousarray(i, np.int32)
        out = np.empty(u.size, np.float32)
        rc = self.lib.svdo_forward(self.h, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p), u.size, out.ctypes.data_as(_f32p))
        if rc:
            raise IndexError("id out of range")
        return out

    def train_step(self, u, i, r, want_logits=True):
        u, i = np.ascontiguousarray(u, np.int32), np.ascontiguousarray(i, np.int32)
        r = np.ascontiguousarray(r, np.float32)
        logits = np.empty(u.size, np.float32) if want_logits else None
        loss, reg = C.c_float(), C.c_float()
        rc = self.lib.svdo_train_step(self.h, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p), r.ctypes.data_as(_f32p),
                                      u.size, logits.ctypes.data_as(_f32p) if want_logits else None,
                                      C.byref(loss), C.byref(reg))
        if rc == -2:
            raise IndexError("id out of range")
        if rc:
            raise MemoryError("svdo_train_step rc=%d" % rc)
        return logits, loss.value, reg.value


def fm_forward(mu, W, V, indptr, indices, data):
    """forward.py:21-22 (general x^2 form) on CSR rows, float64 accumulation, OpenMP over rows: svd_oracle.c fmo_forward"""
    lib = load()
    W, V = np.ascontiguousarray(W, np.float32), np.ascontiguousarray(V, np.float32)
    indptr = np.ascontiguousarray(indptr, np.int64)
    indices, data = np.ascontiguousarray(indices, np.int32), np.ascontiguousarray(data, np.float32)
    n = indptr.size - 1
    out = np.empty(n, np.float64)
    rc = lib.fmo_forward(float(mu), W.ctypes.data_as(_f32p), V.ctypes.data_as(_f32p), V.shape[0], V.shape[1],
                         indptr.ctypes.data_as(C.POINTER(C.c_int64)), indices.ctypes.data_as(_i32p), data.ctypes.data_as(_f32p), n,
                         out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise IndexError("fmo_forward: feature id out of range or dim > 256")
    return out
