"""ctypes binding of libtfrecomm_hip.so (include/tfrecomm.h).

There is no fallback: if the shared library is missing or no HIP device is usable the
import of the library / creation of a model raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C tf-recomm_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtfrecomm_hip.so")

ABI_VERSION = 3       # include/tfrecomm.h TFR_ABI_VERSION this binding was written against (checked at load)
OK, ERR_ARG, ERR_OOB, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4, -5
MU, BU, BI, P, Q = 0, 1, 2, 3, 4
SLOT_M, SLOT_V = 8, 16
LOSS = {"mse": 0, "nll": 1}
OPTIMIZER = {"adam": 0, "sgd": 1}
ADAM_MODE = {"tf1": 0, "lazy": 1}
K_FORWARD, K_SORT, K_REDUCE_ITEM, K_REDUCE_USER, K_APPLY, K_FINALIZE, K_GATHER, K_DRAW = range(8)
KERNEL_NAMES = ["forward", "sort", "reduce_item", "reduce_user", "apply", "finalize", "gather", "draw"]


class TfrOpts(C.Structure):
    _fields_ = [("loss", C.c_int32), ("item_abs", C.c_int32), ("reg_bias", C.c_int32),
                ("optimizer", C.c_int32), ("adam_mode", C.c_int32), ("device", C.c_int32),
                ("lr", C.c_float), ("reg", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("reserved", C.c_int32 * 5)]


class TfrError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("tfrecomm error %d: %s" % (code, text))
        self.code = code


class OutOfRangeError(TfrError, IndexError):
    """The C-ABI's TFR_ERR_OOB: what TensorFlow's CPU gather raises as InvalidArgumentError."""


_p = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes).  Must list every symbol include/tfrecomm.h declares
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "tfr_create": (C.c_int, [C.POINTER(_p), C.c_int64, C.c_int64, C.c_int32, C.POINTER(TfrOpts)]),
    "tfr_destroy": (C.c_int, [_p]),
    "tfr_default_opts": (None, [C.POINTER(TfrOpts)]),
    "tfr_init_tables": (C.c_int, [_p, C.c_uint64, C.c_float, C.c_float]),
    "tfr_set_triples_dev": (C.c_int, [_p, _p, _p, _p, C.c_int64]),
    "tfr_set_table": (C.c_int, [_p, C.c_int32, _f32p, C.c_int64]),
    "tfr_get_table": (C.c_int, [_p, C.c_int32, _f32p, C.c_int64]),
    "tfr_set_frozen": (C.c_int, [_p, C.c_uint32]),
    "tfr_get_step": (C.c_int, [_p, _i64p, _f32p, _f32p]),
    "tfr_set_step": (C.c_int, [_p, C.c_int64, C.c_float, C.c_float]),
    "tfr_set_hyper": (C.c_int, [_p, C.c_float, C.c_float]),
    "tfr_forward": (C.c_int, [_p, _i32p, _i32p, C.c_int64, _f32p]),
    "tfr_eval": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64, C.POINTER(C.c_double), _i64p]),
    "tfr_upload_eval_triples": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64]),
    "tfr_eval_resident": (C.c_int, [_p, C.POINTER(C.c_double), _i64p, _i64p]),
    "tfr_eval_binary": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64, _i64p, _f64p, _f64p]),
    "tfr_eval_binary_resident": (C.c_int, [_p, _i64p, _f64p, _f64p, _i64p]),
    "tfr_auc_dev": (C.c_int, [_p, _p, _p, C.c_int64, _f64p]),
    "tfr_last_batch_auc": (C.c_int, [_p, _f64p]),
    "tfr_train_step": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64, _f32p, _f32p, _f32p]),
    "tfr_upload_triples": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64]),
    "tfr_train_steps_resident": (C.c_int, [_p, _i64p, C.c_int64, C.c_int32, _f32p]),
    "tfr_stage_ids": (C.c_int, [_p, _i64p, C.c_int64]),
    "tfr_train_steps_staged": (C.c_int, [_p, C.c_int64, C.c_int64, C.c_int32, _f32p]),
    "tfr_train_steps_repeat": (C.c_int, [_p, _i32p, _i32p, _f32p, C.c_int64, C.c_int32, _f32p, _f32p]),
    "tfr_forward_resident": (C.c_int, [_p, C.c_int64, C.c_int64, _f32p]),
    "tfr_rng_seed": (C.c_int, [_p, C.c_uint32]),
    "tfr_rng_set_state": (C.c_int, [_p, C.POINTER(C.c_uint32), C.c_int32]),
    "tfr_rng_get_state": (C.c_int, [_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "tfr_draw_ids": (C.c_int, [_p, C.c_int64, C.c_int64, _i64p]),
    "tfr_draw_ids_dev": (C.c_int, [_p, C.c_int64, C.c_int64, _p]),
    "tfr_join_draws": (C.c_int, [_p]),
    "tfr_join_draw": (C.c_int, [_p, C.c_int64]),
    "tfr_train_steps_drawn": (C.c_int, [_p, C.c_int64, C.c_int32, _f32p]),
    "tfr_train_step_ids": (C.c_int, [_p, _i64p, C.c_int64]),
    "tfr_forward_dev": (C.c_int, [_p, _p, _p, C.c_int64, _p]),
    "tfr_train_step_dev": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p]),
    "tfr_table_devptr": (C.c_int, [_p, C.c_int32, C.POINTER(_p), _i64p]),
    "tfr_set_stream": (C.c_int, [_p, _p]),
    "tfr_switch_stream": (C.c_int, [_p, _p]),
    "tfr_get_stream": (C.c_int, [_p, C.POINTER(_p)]),
    "tfr_scalars_devptr": (C.c_int, [_p, C.POINTER(_p)]),
    "tfr_shard_row_stride": (C.c_int32, [_p]),
    "tfr_shard_route": (C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _p]),
    "tfr_shard_route_ids": (C.c_int, [_p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _p]),
    "tfr_shard_bucket_ids": (C.c_int, [_p, _p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, _p]),
    "tfr_shard_route_recs": (C.c_int, [_p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _p]),
    "tfr_shard_routed_devptrs": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "tfr_shard_gather": (C.c_int, [_p, _p, C.c_int64, _p]),
    "tfr_shard_forward_reduce": (C.c_int, [_p, _p, _p, _p, _p]),
    "tfr_shard_forward_items": (C.c_int, [_p, _p, _p, _p, _p]),
    "tfr_shard_reduce_users": (C.c_int, [_p, _p]),
    "tfr_shard_apply_items": (C.c_int, [_p, _p, _p, C.c_int64]),
    "tfr_shard_select": (C.c_int, [_p, C.c_int32]),
    "tfr_shard_presort": (C.c_int, [_p, _p, C.c_int64]),
    "tfr_shard_finish_step": (C.c_int, [_p, _p]),
    "tfr_dp_flat_size": (C.c_int64, [_p]),
    "tfr_dp_local_grads": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p, _p]),
    "tfr_dp_hint_next": (C.c_int, [_p, _p]),
    "tfr_dp_apply": (C.c_int, [_p, _p]),
    "tfr_staged_ids_devptr": (C.c_int, [_p, C.POINTER(_p), _i64p]),
    "tfr_fm_create": (C.c_int, [C.POINTER(_p), C.c_int64, C.c_int32, C.POINTER(TfrOpts)]),
    "tfr_fm_destroy": (C.c_int, [_p]),
    "tfr_fm_set": (C.c_int, [_p, C.c_float, _f32p, _f32p]),
    "tfr_fm_get": (C.c_int, [_p, _f32p, _f32p, _f32p]),
    "tfr_fm_train_step": (C.c_int, [_p, _i64p, _i32p, _f32p, _f32p, C.c_int64, _f32p, _f32p]),
    "tfr_fm_train_step_dev": (C.c_int, [_p, _p, _p, _p, _p, C.c_int64, C.c_int64, _p]),
    "tfr_fm_init": (C.c_int, [_p, C.c_uint64, C.c_float]),
    "tfr_fm_forward": (C.c_int, [_p, _i64p, _i32p, _f32p, C.c_int64, _f32p]),
    "tfr_fm_forward_dev": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p]),
    "tfr_fm_sync": (C.c_int, [_p, _f32p]),
    "tfr_fm_last_error": (C.c_char_p, []),
    "tfr_als_create": (C.c_int, [C.POINTER(_p), C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int32]),
    "tfr_als_destroy": (C.c_int, [_p]),
    "tfr_als_set": (C.c_int, [_p, _f64p, _f64p, _f64p, _f64p]),
    "tfr_als_get": (C.c_int, [_p, _f64p, _f64p, _f64p, _f64p, _f64p]),
    "tfr_als_set_bias": (C.c_int, [_p, C.c_double]),
    "tfr_als_load": (C.c_int, [_p, _i64p, _i64p, _f64p, C.c_int64]),
    "tfr_als_sweep": (C.c_int, [_p, C.c_int32, _f32p]),
    "tfr_als_predict": (C.c_int, [_p, _i64p, _i64p, C.c_int64, _f64p]),
    "tfr_als_last_error": (C.c_char_p, []),
    "tfr_sort_segments": (C.c_int, [_p, C.c_int32, _i32p, C.c_int64, _i32p, _i32p]),
    "tfr_kernel_plan": (C.c_int, [_p, C.c_int64, C.c_char_p, C.c_int64]),
    "tfr_profile": (C.c_int, [_p, C.c_int32]),
    "tfr_profile_read": (C.c_int, [_p, C.c_int32, C.POINTER(C.c_double), _i64p]),
    "tfr_lds_bytes": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p]),
    "tfr_sync": (C.c_int, [_p]),
    "tfr_last_error": (C.c_char_p, []),
    "tfr_version": (C.c_int, []),
    "tfr_device_count": (C.c_int, []),
    "tfr_device_copy_rate": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _f64p, _f64p]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: the HIP extension is not built (run __graft_entry__.build()). "
                "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.tfr_version.restype = C.c_int
        lib.tfr_version.argtypes = []
        got = lib.tfr_version()
        if got != ABI_VERSION:      # a stale .so under a newer binding (or the reverse) would be called with the wrong argument layout
            raise ImportError("%s reports ABI version %d, this binding needs %d: rebuild it (__graft_entry__.build())"
                              % (LIB_PATH, got, ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().tfr_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        text = last_error()
        raise (OutOfRangeError if rc == ERR_OOB else TfrError)(rc, text)


def as_i32(x, what="ids"):
    """Feed columns arrive as any numeric dtype (the reference's iterator yields float64,
    dataio.py:103,117; TF casts to the int32 placeholder, svd_train_val.py:40-41).
    The cast must be exact: a non-integral or out-of-int32 id is an error, not a truncation."""
    a = np.asarray(x)
    if a.dtype == np.int32:
        return np.ascontiguousarray(a)
    if a.dtype.kind in "iu":
        if a.size and (a.min() < -2**31 or a.max() >= 2**31):
            raise OutOfRangeError(ERR_OOB, "%s do not fit int32" % what)
        return np.ascontiguousarray(a.astype(np.int32))
    if a.dtype.kind == "f":
        r = np.rint(a)
        if a.size and (not np.all(r == a) or np.abs(a).max() >= 2**31):
            raise ValueError("%s must be integral values that fit int32" % what)
        return np.ascontiguousarray(r.astype(np.int32))
    raise TypeError("%s: unsupported dtype %s" % (what, a.dtype))


def as_f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def ptr_i32(a):
    return a.ctypes.data_as(_i32p)


def ptr_i64(a):
    return a.ctypes.data_as(_i64p)


def ptr_f32(a):
    return a.ctypes.data_as(_f32p)
