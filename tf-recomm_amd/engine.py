"""SvdModel: thin object wrapper over the C-ABI handle (include/tfrecomm.h).

Holds no arithmetic: every number comes from the HIP kernels.  The reference-shaped
surface (``ops.inference_svd`` / ``ops.optimization`` / ``Session.run``) in ``ops.py``
and ``graph.py`` is built on this class.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def device_copy_rate(device=0, nbytes=1 << 30, reps=10):
    """(best, mean) GB/s (read + write) of the library's own float4 copy kernel - bench.py's same-device yardstick."""
    lib = L.load()
    best, mean = C.c_double(), C.c_double()
    L.check(lib.tfr_device_copy_rate(int(device), int(nbytes), int(reps), C.byref(best), C.byref(mean)))
    return best.value, mean.value


class SvdModel:
    """The five trainables of ops.py:8-12,29-32 (+ optimiser slots) resident in HBM."""

    def __init__(self, user_num, item_num, dim, *, loss="mse", item_abs=False, reg_bias=False,
                 optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05, beta1=0.9, beta2=0.999,
                 eps=1e-8, device=0):
        lib = L.load()
        o = L.TfrOpts()
        lib.tfr_default_opts(C.byref(o))
        o.loss = L.LOSS[loss]
        o.item_abs = int(bool(item_abs))
        o.reg_bias = int(bool(reg_bias))
        o.optimizer = L.OPTIMIZER[optimizer]
        o.adam_mode = L.ADAM_MODE[adam_mode]
        o.device = int(device)
        o.lr, o.reg, o.beta1, o.beta2, o.eps = lr, reg, beta1, beta2, eps
        self._h = L._p()
        self._lib = lib
        self.user_num, self.item_num, self.dim = int(user_num), int(item_num), int(dim)
        self.loss, self.optimizer, self.adam_mode = loss, optimizer, adam_mode
        self.item_abs, self.reg_bias = bool(item_abs), bool(reg_bias)
        self.device = int(device)
        L.check(lib.tfr_create(C.byref(self._h), self.user_num, self.item_num, self.dim, C.byref(o)))

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.tfr_destroy(self._h)
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

    # -- variables ----------------------------------------------------------------
    def _count(self, which):
        t = which & 7
        return {L.MU: 1, L.BU: self.user_num, L.BI: self.item_num,
                L.P: self.user_num * self.dim, L.Q: self.item_num * self.dim}[t]

    def _shape(self, which):
        t = which & 7
        return {L.MU: (), L.BU: (self.user_num,), L.BI: (self.item_num,),
                L.P: (self.user_num, self.dim), L.Q: (self.item_num, self.dim)}[t]

    def set_table(self, which, values):
        a = L.as_f32(values).reshape(-1)
        L.check(self._lib.tfr_set_table(self._h, which, L.ptr_f32(a), a.size))

    def get_table(self, which):
        out = np.empty(self._count(which), np.float32)
        L.check(self._lib.tfr_get_table(self._h, which, L.ptr_f32(out), out.size))
        return out.reshape(self._shape(which))

    def init_tables(self, seed=0, feature_stddev=0.02, bias_stddev=1.0):
        """tf.global_variables_initializer() (svd_train_val.py:53,56) with the initialisers of
        ops.py:9-12,29-32, drawn on the device."""
        L.check(self._lib.tfr_init_tables(self._h, int(seed), feature_stddev, bias_stddev))

    def set_tables(self, mu, bu, bi, P, Q):
        for which, val in ((L.MU, mu), (L.BU, bu), (L.BI, bi), (L.P, P), (L.Q, Q)):
            self.set_table(which, val)

    def tables(self):
        return {w: self.get_table(w) for w in (L.MU, L.BU, L.BI, L.P, L.Q)}

    def set_frozen(self, mask):
        L.check(self._lib.tfr_set_frozen(self._h, int(mask)))

    def set_hyper(self, lr, reg):
        L.check(self._lib.tfr_set_hyper(self._h, lr, reg))

    @property
    def step(self):
        return self.get_step()[0]

    def get_step(self):
        s, a, b = C.c_int64(), C.c_float(), C.c_float()
        L.check(self._lib.tfr_get_step(self._h, C.byref(s), C.byref(a), C.byref(b)))
        return s.value, a.value, b.value

    def set_step(self, step, beta1_power, beta2_power):
        L.check(self._lib.tfr_set_step(self._h, int(step), beta1_power, beta2_power))

    # -- forward / train ----------------------------------------------------------
    def forward(self, users, items):
        u, i = L.as_i32(users, "user ids"), L.as_i32(items, "item ids")
        if u.shape != i.shape or u.ndim != 1:
            raise ValueError("user and item batches must be 1-D and of equal length")
        out = np.empty(u.size, np.float32)
        L.check(self._lib.tfr_forward(self._h, L.ptr_i32(u), L.ptr_i32(i), u.size, L.ptr_f32(out)))
        return out

    def eval(self, users, items, rates):
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        if not (u.shape == i.shape == r.shape) or u.ndim != 1:
            raise ValueError("batches must be 1-D and of equal length")
        sse, neq = C.c_double(), C.c_int64()
        L.check(self._lib.tfr_eval(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size,
                                   C.byref(sse), C.byref(neq)))
        return sse.value, neq.value

    def upload_eval_triples(self, users, items, rates):
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        L.check(self._lib.tfr_upload_eval_triples(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size))

    def eval_resident(self):
        """(sum of squared errors, number of infer == rate, n) over the resident validation set."""
        sse, neq, n = C.c_double(), C.c_int64(), C.c_int64()
        L.check(self._lib.tfr_eval_resident(self._h, C.byref(sse), C.byref(neq), C.byref(n)))
        return sse.value, neq.value, n.value

    def eval_binary(self, users, items, rates):
        """The fork's epoch metrics on the device (svd_train_val.py:94-98,170-178): dict(acc, mean_nll, auc, n)."""
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        if not (u.shape == i.shape == r.shape) or u.ndim != 1:
            raise ValueError("batches must be 1-D and of equal length")
        neq, nll, auc = C.c_int64(), C.c_double(), C.c_double()
        L.check(self._lib.tfr_eval_binary(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size,
                                          C.byref(neq), C.byref(nll), C.byref(auc)))
        n = max(1, u.size)
        return dict(acc=neq.value / n, mean_nll=nll.value / n, auc=auc.value, n=u.size)

    def eval_binary_resident(self):
        neq, nll, auc, n = C.c_int64(), C.c_double(), C.c_double(), C.c_int64()
        L.check(self._lib.tfr_eval_binary_resident(self._h, C.byref(neq), C.byref(nll), C.byref(auc), C.byref(n)))
        return dict(acc=neq.value / max(1, n.value), mean_nll=nll.value / max(1, n.value), auc=auc.value, n=n.value)

    def last_batch_auc(self):
        """AUC of the batch the last ``train_step`` (with logits) ran on - svd_train_val.py:97 on the device."""
        auc = C.c_double()
        L.check(self._lib.tfr_last_batch_auc(self._h, C.byref(auc)))
        return auc.value

    def auc_dev(self, d_score, d_label, n):
        auc = C.c_double()
        L.check(self._lib.tfr_auc_dev(self._h, d_score, d_label, int(n), C.byref(auc)))
        return auc.value

    def train_step(self, users, items, rates, want_logits=True):
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        if not (u.shape == i.shape == r.shape) or u.ndim != 1:
            raise ValueError("batches must be 1-D and of equal length")
        logits = np.empty(u.size, np.float32) if want_logits else None
        loss, reg = C.c_float(), C.c_float()
        L.check(self._lib.tfr_train_step(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size,
                                         L.ptr_f32(logits) if want_logits else None,
                                         C.byref(loss), C.byref(reg)))
        return logits, loss.value, reg.value

    def train_steps_repeat(self, users, items, rates, nsteps, want_logits=True, want_loss=True):
        """``for _ in range(nsteps): sess.run(train_op, feed_dict)`` on ONE batch (the per-user fine-tuning loops of
        adaptive_test.py:104-116 / non_adaptive_test.py:82-87) without a host round trip between the steps.
        Returns (pre-update logits of the last step, data loss per step)."""
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        if not (u.shape == i.shape == r.shape) or u.ndim != 1:
            raise ValueError("batches must be 1-D and of equal length")
        logits = np.empty(u.size, np.float32) if want_logits else None
        loss = np.empty(int(nsteps), np.float32) if want_loss else None
        L.check(self._lib.tfr_train_steps_repeat(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size, int(nsteps),
                                                 L.ptr_f32(logits) if want_logits else None,
                                                 L.ptr_f32(loss) if want_loss else None))
        return logits, loss

    # -- resident store -----------------------------------------------------------
    def upload_triples(self, users, items, rates):
        u, i, r = L.as_i32(users, "user ids"), L.as_i32(items, "item ids"), L.as_f32(rates)
        L.check(self._lib.tfr_upload_triples(self._h, L.ptr_i32(u), L.ptr_i32(i), L.ptr_f32(r), u.size))

    def set_triples_dev(self, d_user, d_item, d_rate, n):
        L.check(self._lib.tfr_set_triples_dev(self._h, d_user, d_item, d_rate, n))

    def train_steps_resident(self, ids, batch, want_loss=True):
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64)).reshape(-1)
        if ids.size % batch:
            raise ValueError("ids length must be a multiple of batch")
        n = ids.size // batch
        loss = np.empty(n, np.float32) if want_loss else None
        L.check(self._lib.tfr_train_steps_resident(self._h, L.ptr_i64(ids), batch, n,
                                                   L.ptr_f32(loss) if want_loss else None))
        return loss

    def stage_ids(self, ids):
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64)).reshape(-1)
        L.check(self._lib.tfr_stage_ids(self._h, L.ptr_i64(ids), ids.size))

    def train_steps_staged(self, first_step, batch, nsteps, want_loss=False):
        loss = np.empty(nsteps, np.float32) if want_loss else None
        L.check(self._lib.tfr_train_steps_staged(self._h, first_step, batch, nsteps,
                                                 L.ptr_f32(loss) if want_loss else None))
        return loss

    # -- the ShuffleIterator id draw on the device (dataio.py:115; include/tfrecomm.h "id draw") -----
    def rng_seed(self, seed):
        """``np.random.seed(seed)`` for the device generator (svd_train_val.py:15)."""
        L.check(self._lib.tfr_rng_seed(self._h, int(seed) & 0xFFFFFFFF))

    def rng_set_state(self, key, pos):
        k = np.ascontiguousarray(np.asarray(key, dtype=np.uint32)).reshape(-1)
        if k.size != 624:
            raise ValueError("MT19937 key must hold 624 words")
        L.check(self._lib.tfr_rng_set_state(self._h, k.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos)))

    def rng_get_state(self):
        k, pos = np.empty(624, np.uint32), C.c_int32()
        L.check(self._lib.tfr_rng_get_state(self._h, k.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos)))
        return k, pos.value

    def rng_from_numpy(self):
        """Hand NumPy's legacy global generator (the one dataio.ShuffleIterator draws from) to the device."""
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise RuntimeError("NumPy's legacy generator is not MT19937")
        self.rng_set_state(st[1], st[2])

    def rng_to_numpy(self):
        """...and back: host draws after this continue the stream where the device left it."""
        k, pos = self.rng_get_state()
        st = np.random.get_state()
        np.random.set_state((st[0], k, pos, st[3], st[4]))

    def draw_ids(self, high, count):
        """``np.random.randint(0, high, (count,))`` drawn on the device (int64, bit-identical stream)."""
        out = np.empty(int(count), np.int64)
        L.check(self._lib.tfr_draw_ids(self._h, int(high), out.size, L.ptr_i64(out)))
        return out

    def draw_ids_dev(self, high, count, d_out):
        """``np.random.randint(0, high, (count,))`` into device memory (pointer), asynchronously on the draw stream"""
        L.check(self._lib.tfr_draw_ids_dev(self._h, int(high), int(count), d_out))

    def join_draws(self):
        """the model's stream waits for every draw issued by ``draw_ids_dev`` so far"""
        L.check(self._lib.tfr_join_draws(self._h))

    def join_draw(self, ordinal):
        """...for the first ``ordinal`` draws issued (counted from 1) only"""
        L.check(self._lib.tfr_join_draw(self._h, int(ordinal)))

    def train_steps_drawn(self, batch, nsteps, want_loss=False):
        """nsteps x { next(iter_train); sess.run(train_op) } with the id draw, the gather from the resident
        store and the step all on the device."""
        loss = np.empty(nsteps, np.float32) if want_loss else None
        L.check(self._lib.tfr_train_steps_drawn(self._h, int(batch), int(nsteps), L.ptr_f32(loss) if want_loss else None))
        return loss

    def train_step_ids(self, ids):
        """One step on store rows ``ids`` (host int64, e.g. straight from np.random.randint); asynchronous."""
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int64)).reshape(-1)
        L.check(self._lib.tfr_train_step_ids(self._h, L.ptr_i64(a), a.size))

    def forward_resident(self, lo, hi):
        out = np.empty(hi - lo, np.float32)
        L.check(self._lib.tfr_forward_resident(self._h, lo, hi, L.ptr_f32(out)))
        return out

    def sort_segments(self, side, ids):
        a = L.as_i32(ids)
        ks, ps = np.empty(a.size, np.int32), np.empty(a.size, np.int32)
        L.check(self._lib.tfr_sort_segments(self._h, side, L.ptr_i32(a), a.size, L.ptr_i32(ks), L.ptr_i32(ps)))
        return ks, ps

    # -- device-pointer plumbing --------------------------------------------------
    def forward_dev(self, d_user, d_item, batch, d_logits):
        L.check(self._lib.tfr_forward_dev(self._h, d_user, d_item, batch, d_logits))

    def train_step_dev(self, d_user, d_item, d_rate, batch, d_logits=None):
        L.check(self._lib.tfr_train_step_dev(self._h, d_user, d_item, d_rate, batch, d_logits))

    # -- row-sharded building blocks (device pointers; see include/tfrecomm.h) ------
    def shard_row_stride(self):
        return int(self._lib.tfr_shard_row_stride(self._h))

    def shard_route(self, d_user, d_item, d_rate, batch_global, rank, world, user_num_global, item_num_global,
                    sample_cap, slot_cap, d_req):
        L.check(self._lib.tfr_shard_route(self._h, d_user, d_item, d_rate, batch_global, rank, world, user_num_global,
                                          item_num_global, sample_cap, slot_cap, d_req))

    def shard_route_ids(self, d_ids, batch_global, rank, world, user_num_global, item_num_global, sample_cap, slot_cap, d_req):
        L.check(self._lib.tfr_shard_route_ids(self._h, d_ids, batch_global, rank, world, user_num_global, item_num_global,
                                              sample_cap, slot_cap, d_req))

    def shard_bucket_ids(self, d_ids, batch, world, user_num_global, pair_cap, d_send):
        L.check(self._lib.tfr_shard_bucket_ids(self._h, d_ids, batch, world, user_num_global, pair_cap, d_send))

    def shard_route_recs(self, d_recs, n, rank, world, user_num_global, item_num_global, sample_cap, slot_cap, d_req):
        L.check(self._lib.tfr_shard_route_recs(self._h, d_recs, n, rank, world, user_num_global, item_num_global,
                                               sample_cap, slot_cap, d_req))

    def shard_routed_devptrs(self):
        ps = [L._p() for _ in range(4)]
        L.check(self._lib.tfr_shard_routed_devptrs(self._h, *[C.byref(p) for p in ps]))
        return tuple(p.value for p in ps)

    def shard_gather(self, d_req_recv, n, d_rows_out):
        L.check(self._lib.tfr_shard_gather(self._h, d_req_recv, n, d_rows_out))

    def shard_forward_reduce(self, d_item_rows, d_logits, d_item_grad, d_scalars4):
        L.check(self._lib.tfr_shard_forward_reduce(self._h, d_item_rows, d_logits, d_item_grad, d_scalars4))

    def shard_forward_items(self, d_item_rows, d_logits, d_item_grad, d_scalars4):
        L.check(self._lib.tfr_shard_forward_items(self._h, d_item_rows, d_logits, d_item_grad, d_scalars4))

    def shard_reduce_users(self, d_item_rows):
        L.check(self._lib.tfr_shard_reduce_users(self._h, d_item_rows))

    def shard_apply_items(self, d_req_recv, d_grad_recv, n):
        L.check(self._lib.tfr_shard_apply_items(self._h, d_req_recv, d_grad_recv, n))

    def shard_select(self, which):
        L.check(self._lib.tfr_shard_select(self._h, int(which)))

    def shard_presort(self, d_req_recv, n):
        L.check(self._lib.tfr_shard_presort(self._h, d_req_recv, int(n)))

    def shard_finish_step(self, d_scalars4):
        L.check(self._lib.tfr_shard_finish_step(self._h, d_scalars4))

    # -- data-parallel building blocks ------------------------------------------------
    def dp_flat_size(self):
        return int(self._lib.tfr_dp_flat_size(self._h))

    def dp_hint_next(self, d_next_store_ids):
        """Look-ahead: the store ids the next ``dp_local_grads`` call will use (device pointer)."""
        L.check(self._lib.tfr_dp_hint_next(self._h, d_next_store_ids))

    def dp_local_grads(self, d_user, d_item, d_rate, batch, d_store_ids, d_flat):
        L.check(self._lib.tfr_dp_local_grads(self._h, d_user, d_item, d_rate, batch, d_store_ids, d_flat))

    def dp_apply(self, d_flat):
        L.check(self._lib.tfr_dp_apply(self._h, d_flat))

    def staged_ids_devptr(self):
        p, n = L._p(), C.c_int64()
        L.check(self._lib.tfr_staged_ids_devptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def table_devptr(self, which):
        p, n = L._p(), C.c_int64()
        L.check(self._lib.tfr_table_devptr(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_stream(self, stream_ptr):
        L.check(self._lib.tfr_set_stream(self._h, stream_ptr))

    def switch_stream(self, stream_ptr):
        """set_stream without draining the stream in use (the caller orders the streams itself)"""
        L.check(self._lib.tfr_switch_stream(self._h, stream_ptr))

    def get_stream(self):
        p = L._p()
        L.check(self._lib.tfr_get_stream(self._h, C.byref(p)))
        return p.value

    def sync(self):
        L.check(self._lib.tfr_sync(self._h))

    # -- per-kernel HIP-event timing ----------------------------------------------
    def profile(self, enable=True):
        L.check(self._lib.tfr_profile(self._h, int(bool(enable))))

    def kernel_plan(self, batch):
        """{slot: kernel symbol} of one training step at this batch size (rocprofv3's spelling)."""
        buf = C.create_string_buffer(1024)
        L.check(self._lib.tfr_kernel_plan(self._h, int(batch), buf, 1024))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if kv)

    def profile_read(self):
        out = {}
        for k, name in enumerate(L.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_int64()
            L.check(self._lib.tfr_profile_read(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out
