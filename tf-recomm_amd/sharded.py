"""Row-sharded SVD training across the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI) for the exchanges, the HIP kernels of
this package for everything else - including the integer routing.

Partition: block row-sharding - ``owner = id // ceil(rows / world)`` - of ``user_features`` /
``user_bias`` by user id and of ``item_features`` / ``item_bias`` by item id, Adam slots
co-located with their rows, ``bias_global`` replicated.

One step on the *global* batch (every rank sees the same ids; the reference has no distributed
path, the semantics are those of one ``minimize`` on the whole batch, ops.py:143-149):

  0. ``route``: every rank keeps the samples whose USER row it owns and lays the distinct ITEM ids
     among them out as requests ``[world][slot_cap]`` (slot ``w * slot_cap + k`` = the k-th distinct
     item this rank needs from owner ``w``; unused slots hold -1).  Device kernels (csrc/shard.hip);
  1. all-to-all of the request slots (int32), owners ``gather`` the rows: one packed row per slot =
     ``dim`` features, the bias, padding to 16 bytes;  all-to-all of the rows back;
  2. ``forward_reduce`` locally: user rows are updated in place, item-row gradients - already reduced
     over the rank's own samples - come out per slot in the same packed layout;
  3. all-to-all of the gradient rows to the owners, which add them in rank order (deterministic) and
     apply the optimiser (``apply_items``);
  4. all-reduce of 3 scalars (loss, regulariser, sum g) -> ``bias_global`` update everywhere.

Three equal-split all-to-alls and one 16-byte all-reduce per step, and NO host synchronisation: the
capacities are fixed, how many slots / samples are really in use stays on the device.  A step that
overflows a capacity is void and reported like an out-of-range id at the next sync.

Wire volume per rank and step is ``2 * stride * 4`` bytes per request slot each way (fetching user
rows as well would double it).  An all-to-all maps one-to-one onto the 7 direct xGMI links of a
GPU; the path is xGMI-bound, not HBM-bound.

The compute backend is injected so the exchange logic can be exercised on CPU with ``gloo``
(tests/test_sharded_cpu.py uses an oracle-backed stand-in; the product backend is ``HipShard`` below
and has no CPU path).
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from .engine import SvdModel

SETUP_STEPS = 30     # untimed steps ahead of the warm-up in bench_entry (one-off costs of a process's first collectives)
XGMI_EGRESS_GBS = 7 * 153.0          # MI355X: 7 point-to-point links x ~153 GB/s per GPU


def rows_per_rank(rows, world):
    return int(math.ceil(rows / world))


def shard_range(rows, world, rank):
    per = rows_per_rank(rows, world)
    lo = min(rows, rank * per)
    return lo, min(rows, lo + per)


def packed_stride(dim):
    """floats per exchanged row: the features, the bias, the sender's error flag, padding to 16 bytes (csrc/api.hip shard_stride)"""
    return dim + 4 if dim % 4 == 0 else dim + 2


# TFR_SHARD_ALIAS_WORLD1=1: in a one-rank world the equal-split all-to-all returns its input (RCCL would copy the buffer onto
# itself, 150 + 140 us per step at the C3 shape) - what is left is the step's kernel time.  A measurement knob for the one-GPU
# rehearsal, off by default: the rehearsal's point is to run the real exchanges.
_ALIAS_WORLD1 = os.environ.get("TFR_SHARD_ALIAS_WORLD1") == "1"


class Comm(object):
    """equal-split all-to-all / all-reduce over ``torch.distributed``.  With a CPU-only backend (gloo)
    device tensors are staged through host memory - the rehearsal path; RCCL takes them as is."""

    def __init__(self, group=None, side_group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.stage = dist.get_backend(group) == "gloo"
        # a second communicator for the small exchanges of the NEXT batch's routing (RCCL runs a communicator's collectives in
        # issue order on one stream: on their own communicator they do not queue in front of this step's row exchanges)
        self.side_group = side_group

    def side(self):
        """the same exchanges on the side communicator (falls back to the main one)"""
        if self.side_group is None:
            return self
        c = Comm.__new__(Comm)
        c.group, c.world, c.rank, c.stage, c.side_group = self.side_group, self.world, self.rank, self.stage, None
        return c

    def all_to_all(self, inp, out=None):
        """``inp`` = ``world`` equal chunks along dim 0, chunk w for rank w; returns the chunks received, by source."""
        if self.world == 1 and _ALIAS_WORLD1:
            return inp                                   # measurement knob: a one-rank world exchanges nothing (kernel time only)
        if out is None:
            out = torch.empty_like(inp)
        if self.stage and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, group=self.group)
        return out

    def all_to_all_start(self, inp, out=None):
        """the same exchange, started without making the caller's stream wait for it: returns (out, work); ``work.wait()`` (or
        None with the staged CPU backend, which has already finished) orders the current stream behind the exchange"""
        if self.stage or not inp.is_cuda or (self.world == 1 and _ALIAS_WORLD1):
            return self.all_to_all(inp, out), None
        if out is None:
            out = torch.empty_like(inp)
        return out, dist.all_to_all_single(out, inp, group=self.group, async_op=True)

    def all_reduce_sum(self, t):
        if self.stage and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def barrier(self):
        dist.barrier(group=self.group)


class HipShard(object):
    """The product backend: this rank's shard in HBM behind the C-ABI, torch tensors as the
    exchange buffers (device memory and streams are plumbing; the arithmetic is the kernels')."""

    def __init__(self, u_rows, i_rows, dim, device, **opts):
        self.device = torch.device("cuda", device)
        self.D = dim
        self.model = SvdModel(max(1, u_rows), max(1, i_rows), dim, device=device, **opts)
        self.stream = torch.cuda.Stream(device=self.device)
        self._mstream = self.stream                      # the torch stream the model's kernels are launched on right now
        self.model.set_stream(self.stream.cuda_stream)
        self.stride = self.model.shard_row_stride()
        self._buf = {}
        self._routed = None
        self._set = 0

    # The model's kernels run on self.stream.  When the caller has made that torch's current stream (bench_entry
    # does) the collectives order themselves against it and no cross-stream events are needed; otherwise fence.
    def _foreign(self):
        cur = torch.cuda.current_stream(self.device)
        return None if cur == self._mstream else cur

    def _sync_in(self):
        cur = self._foreign()
        if cur is not None:
            self._mstream.wait_stream(cur)

    def _sync_out(self):
        cur = self._foreign()
        if cur is not None:
            cur.wait_stream(self._mstream)

    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def route(self, u, i, r, rank, world, U, I, sample_cap, slot_cap):
        req = self._get("req", (world * slot_cap,), torch.int32)
        self._sync_in()
        self.model.shard_route(u.data_ptr(), i.data_ptr(), r.data_ptr(), u.numel(), rank, world, U, I, sample_cap, slot_cap,
                               req.data_ptr())
        self._sync_out()
        self._routed = (sample_cap, world)
        return req

    def set_store(self, u, i, r):
        """this rank's copy of the rating store (GLOBAL user / item ids; int32, int32, float32 tensors on the device)"""
        self._sync_in()
        self.model.set_triples_dev(u.data_ptr(), i.data_ptr(), r.data_ptr(), u.numel())
        self._sync_out()

    def route_ids(self, ids, rank, world, U, I, sample_cap, slot_cap):
        """`route` with the global batch given as rows of the store (int64 tensor on the device): the gather of dataio.py:115-117
        happens inside the routing kernels, only this rank's samples leave the store"""
        req = self._get("req", (world * slot_cap,), torch.int32)
        self._sync_in()
        self.model.shard_route_ids(ids.data_ptr(), ids.numel(), rank, world, U, I, sample_cap, slot_cap, req.data_ptr())
        self._sync_out()
        self._routed = (sample_cap, world)
        return req

    def select(self, which):
        """which of the model's two routed-batch sets the following calls fill / consume (and which exchange buffers)"""
        self._set = int(which)
        self.model.shard_select(which)

    def presort(self, req_recv):
        self._sync_in()
        self.model.shard_presort(req_recv.data_ptr(), req_recv.numel())
        self._sync_out()

    def on_stream(self, stream):
        """run the following model calls on `stream` (a torch stream), or back on the model's own with None"""
        self._mstream = stream or self.stream
        self.model.switch_stream(self._mstream.cuda_stream)     # no drain: the caller orders the streams with wait_stream

    def bucket_ids(self, ids, world, U, pair_cap):
        """this rank's own batch rows -> [world * pair_cap, 4] int32 records grouped by the owner of the user row"""
        send = self._get("send%d" % self._set, (world * pair_cap, 4), torch.int32)
        self._sync_in()
        self.model.shard_bucket_ids(ids.data_ptr(), ids.numel(), world, U, pair_cap, send.data_ptr())
        self._sync_out()
        return send

    def route_recs(self, recv, rank, world, U, I, sample_cap, slot_cap):
        """`route` on the records received from the peers (all owned by this rank; user = -1 marks an unused slot)"""
        req = self._get("req%d" % self._set, (world * slot_cap,), torch.int32)
        self._sync_in()
        self.model.shard_route_recs(recv.data_ptr(), recv.shape[0], rank, world, U, I, sample_cap, slot_cap, req.data_ptr())
        self._sync_out()
        self._routed = (sample_cap, world)
        return req

    def routed(self):
        """views of the routed batch (test / bookkeeping): mine, u_local, slot [sample_cap]; counts [2 + world]"""
        cap, world = self._routed
        ptrs = self.model.shard_routed_devptrs()

        def view(ptr, n):
            iface = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3}
            holder = type("_V", (), {"__cuda_array_interface__": iface})()
            return torch.as_tensor(holder, device=self.device)
        return dict(mine=view(ptrs[0], cap), u_local=view(ptrs[1], cap), slot=view(ptrs[2], cap), counts=view(ptrs[3], 2 + world))

    def gather(self, req_recv):
        n = req_recv.numel()
        rows = self._get("rows_out", (n, self.stride), torch.float32)
        self._sync_in()
        self.model.shard_gather(req_recv.data_ptr(), n, rows.data_ptr())
        self._sync_out()
        return rows

    def forward_reduce(self, item_rows):
        cap, _ = self._routed
        grad = self._get("grad", tuple(item_rows.shape), torch.float32)
        logits = self._get("logits", (cap,), torch.float32)
        scal = self._get("scal", (4,), torch.float32)
        self._sync_in()
        self.model.shard_forward_reduce(item_rows.data_ptr(), logits.data_ptr(), grad.data_ptr(), scal.data_ptr())
        self._sync_out()
        return grad, scal, logits

    def forward_items(self, item_rows):
        """the item half of forward_reduce: sort, forward + item-side reduce into the exchange buffer, local scalars"""
        cap, _ = self._routed
        grad = self._get("grad", tuple(item_rows.shape), torch.float32)
        logits = self._get("logits", (cap,), torch.float32)
        scal = self._get("scal", (4,), torch.float32)
        self._sync_in()
        self.model.shard_forward_items(item_rows.data_ptr(), logits.data_ptr(), grad.data_ptr(), scal.data_ptr())
        self._sync_out()
        return grad, scal, logits

    def reduce_users(self, item_rows):
        """the user half: touches the fetched rows and the local user rows only - it may run beside the gradient exchange"""
        self._sync_in()
        self.model.shard_reduce_users(item_rows.data_ptr())
        self._sync_out()

    def apply_items(self, req_recv, grad_recv):
        self._sync_in()
        self.model.shard_apply_items(req_recv.data_ptr(), grad_recv.data_ptr(), req_recv.numel())
        self._sync_out()

    def finish_step(self, scal):
        self._sync_in()
        self.model.shard_finish_step(scal.data_ptr())
        self._sync_out()

    def set_tables(self, mu, bu, bi, P, Q):
        self.model.set_tables(mu, bu, bi, P, Q)

    def tables(self):
        return self.model.tables()

    def sync(self):
        self.model.sync()


class ShardedSvd(object):
    """One rank of the row-sharded model.  ``backend_factory(u_rows, i_rows, dim)`` builds the
    compute backend for the local shard.  ``slack``: head-room of the fixed capacities over their expected
    fill at uniform ids (batches up to 65536 use exact upper bounds instead and can never overflow)."""

    def __init__(self, user_num, item_num, dim, comm, backend_factory, device="cpu", slack=1.25):
        self.U, self.I, self.D = int(user_num), int(item_num), int(dim)
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.per_u, self.per_i = rows_per_rank(self.U, self.world), rows_per_rank(self.I, self.world)
        self.u_lo, self.u_hi = shard_range(self.U, self.world, self.rank)
        self.i_lo, self.i_hi = shard_range(self.I, self.world, self.rank)
        self.device = torch.device(device)
        self.backend = backend_factory(self.u_hi - self.u_lo, self.i_hi - self.i_lo, self.D)
        self.slack = float(slack)
        self.timers = None                                # optional: list collecting (phase, start event, end event)

    # -- tables -------------------------------------------------------------------------
    def set_tables_from_global(self, mu, bu, bi, P, Q):
        """Every rank passes the full tables and keeps its slice (set_table stays a contiguous
        slice because the partition is by blocks)."""
        P, Q = np.asarray(P, np.float32), np.asarray(Q, np.float32)
        bu, bi = np.asarray(bu, np.float32), np.asarray(bi, np.float32)

        def sl(x, lo, hi):
            part = x[lo:hi]
            if part.shape[0] == 0:                      # an empty shard keeps one dummy row
                part = np.zeros((1,) + x.shape[1:], np.float32)
            return part
        self.backend.set_tables(np.float32(mu), sl(bu, self.u_lo, self.u_hi), sl(bi, self.i_lo, self.i_hi),
                                sl(P, self.u_lo, self.u_hi), sl(Q, self.i_lo, self.i_hi))

    def local_tables(self):
        return self.backend.tables()

    # -- capacities -----------------------------------------------------------------------
    def capacities(self, batch_global):
        """(sample_cap, slot_cap): fixed sizes of the routed batch and of one owner's request slots"""
        Bg, W = int(batch_global), self.world
        if Bg <= 65536:                                   # small: exact upper bounds
            sample_cap = max(1, Bg)
            return sample_cap, max(1, min(self.per_i, sample_cap))
        sample_cap = min(Bg, int(self.slack * Bg / W) + 4096)
        return sample_cap, max(1, min(self.per_i, sample_cap, int(self.slack * Bg / (W * W)) + 1024))

    def pair_capacity(self, batch_local):
        """records one rank may send to one owner per step (pre-split batches): the whole batch while that is small (an exact
        bound), else ``slack`` x the expected share at uniform user ids"""
        B, W = int(batch_local), self.world
        if B * W <= 65536:
            return max(1, B)
        return min(B, int(self.slack * B / W) + 1024)

    def _phase(self, name):
        if self.timers is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0)

    def _end(self, tok):
        if tok is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.timers.append((tok[0], tok[1], e1))

    # -- one step ------------------------------------------------------------------------
    def train_step(self, u, i, r):
        """``u, i, r``: the GLOBAL batch (identical on every rank; int32, int32, float32 tensors on this rank's
        device).  Returns (logits [sample_cap], mine [sample_cap], global {loss, reg, sum g, -}): the pre-update
        logits of this rank's samples and their positions in the global batch; the first ``counts[0]`` entries
        are in use (``backend.routed()["counts"]``, on the device - nothing here waits for the host)."""
        sample_cap, slot_cap = self.capacities(u.numel())
        t = self._phase("route")
        req = self.backend.route(u, i, r, self.rank, self.world, self.U, self.I, sample_cap, slot_cap)
        self._end(t)
        return self._exchange_and_update(req)

    def train_step_ids(self, ids):
        """The same step with the GLOBAL batch given as rows ``ids`` (int64 tensor, identical on every rank) of the rating
        store every rank holds a copy of (``backend.set_store``): nothing of the batch is materialised outside the routing
        kernels but this rank's own samples."""
        sample_cap, slot_cap = self.capacities(ids.numel())
        t = self._phase("route")
        req = self.backend.route_ids(ids, self.rank, self.world, self.U, self.I, sample_cap, slot_cap)
        self._end(t)
        return self._exchange_and_update(req)

    def train_step_local_ids(self, ids, next_ids=None):
        """Pre-split batches (SURVEY 8e's other variant): ``ids`` are THIS rank's own B rows of the rating store (int64 tensor;
        every rank brings a different batch - e.g. its own id stream).  The records are grouped by the owner of their user row,
        one more equal-split all-to-all (16 bytes per sample) takes them there, and the step goes on as for a global batch of
        ``world * pair_capacity`` slots.  No rank draws, reads or tests a sample of another rank's batch that it does not own.

        ``next_ids`` (the batch of the NEXT call, same shape): its whole integer front end - bucket, sample exchange, routing,
        request exchange and the sorts of the step - is started on a side stream once this step's rows are on their way, into
        the model's second routed-batch set; the next call finds it done and starts at the row gather.  Nothing in that front end
        reads a table, so the trajectory is bit for bit the one of calls without ``next_ids``."""
        c, be = self.comm, self.backend
        pipelined = hasattr(be, "select")
        pre = getattr(self, "_pre", None)
        self._pre = None
        if pipelined and pre is not None and pre["ids"] == (ids.data_ptr(), ids.numel()):
            be.select(pre["set"])
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).wait_stream(self._side)     # the front end of this batch ran there
            be._routed = pre["routed"]
            req_recv, cur = pre["req_recv"], pre["set"]
        else:
            cur = 0
            if pipelined:
                be.select(cur)
            # (with a second routed-batch set every step pre-sorts: the step phases then never touch the sort scratch, which a
            # front end running beside them on the side stream uses)
            req_recv = self._front_end(ids, c, timed=True, presort=pipelined)
        hook_a = hook_b = None
        if pipelined and next_ids is not None:
            # The next batch's front end in two parts, so that on ONE communicator (collectives run in issue order) neither of this
            # step's row-sized exchanges queues behind a small one that still waits for side-stream kernels: bucket + sample
            # exchange + routing once this step's rows are on their way; request exchange + sorts once its gradient rows are.
            state = {}

            def on_side(fn):
                if self.device.type == "cuda":
                    side = self._side_stream()
                    side.wait_stream(torch.cuda.current_stream(self.device))  # ids drawn / everything queued so far
                    ctx = torch.cuda.stream(side)
                else:                                    # CPU stand-in backend (tests): the same call order, no streams
                    import contextlib
                    side, ctx = None, contextlib.nullcontext()
                with ctx:
                    be.on_stream(side)
                    be.select(cur ^ 1)
                    try:
                        fn()
                    finally:
                        be.select(cur)
                        be.on_stream(None)
                be._routed = self._routed_now

            def hook_a():
                on_side(lambda: state.update(req=self._front_end_a(next_ids, c.side(), timed=False), routed=getattr(be, "_routed", None)))

            def hook_b():
                def second():
                    rr = self._front_end_b(state["req"], c.side(), timed=False, presort=True)
                    self._pre = dict(ids=(next_ids.data_ptr(), next_ids.numel()), set=cur ^ 1, req_recv=rr, routed=state["routed"])
                on_side(second)
        self._routed_now = getattr(be, "_routed", None)
        return self._exchange_and_update(None, req_recv=req_recv, after_rows=hook_a, after_grads_start=hook_b)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _front_end(self, ids, c, timed, presort=False):
        """bucket -> sample exchange -> routing -> request exchange (-> the step's sorts): the integer part of a step"""
        return self._front_end_b(self._front_end_a(ids, c, timed), c, timed, presort)

    def _front_end_a(self, ids, c, timed):
        """first half: bucket -> sample exchange -> routing; returns the request slots"""
        be = self.backend
        pair_cap = self.pair_capacity(ids.numel())
        sample_cap, slot_cap = self.capacities(ids.numel() * self.world)
        sample_cap = min(sample_cap, self.world * pair_cap)
        t = self._phase("bucket") if timed else None
        send = be.bucket_ids(ids, self.world, self.U, pair_cap)
        self._end(t)
        t = self._phase("all_to_all samples") if timed else None
        recv = c.all_to_all(send, self._named_buf("recv", send))
        self._end(t)
        t = self._phase("route") if timed else None
        req = be.route_recs(recv, self.rank, self.world, self.U, self.I, sample_cap, slot_cap)
        self._end(t)
        return req

    def _front_end_b(self, req, c, timed, presort):
        """second half: request exchange (-> the step's sorts); returns the requests received as an owner"""
        be = self.backend
        t = self._phase("all_to_all ids") if timed else None
        req_recv = c.all_to_all(req, self._named_buf("req_recv", req))         # slots asked of me, by requester
        self._end(t)
        if presort:
            t = self._phase("presort") if timed else None
            be.presort(req_recv)
            self._end(t)
        return req_recv

    def _named_buf(self, name, like):
        """persistent receive buffers, one per routed-batch set (an exchange of the next batch must not land in a buffer this
        step still reads)"""
        key = "%s%d" % (name, getattr(self.backend, "_set", 0))
        bufs = self.__dict__.setdefault("_bufs", {})
        b = bufs.get(key)
        if b is None or b.shape != like.shape or b.device != like.device or b.dtype != like.dtype:
            b = bufs[key] = torch.empty_like(like)
        return b

    def _exchange_and_update(self, req, req_recv=None, after_rows=None, after_grads_start=None):
        c, be = self.comm, self.backend
        if req_recv is None:
            t = self._phase("all_to_all ids")
            req_recv = c.all_to_all(req)                                   # slots asked of me, by requester
            self._end(t)
        t = self._phase("gather")
        rows_out = be.gather(req_recv)
        self._end(t)
        t = self._phase("all_to_all rows")
        item_rows = c.all_to_all(rows_out, self._named_buf("item_rows", rows_out))   # chunk w = the rows owner w holds for my slots
        self._end(t)
        if after_rows is not None:
            after_rows()                                                   # the next batch's front end starts on the side stream
        if hasattr(be, "forward_items"):
            # the user half needs nothing the gradient exchange touches: it runs on the model's stream while RCCL moves the
            # gradient rows on its own; the 16-byte all-reduce of the scalars goes out early as well
            t = self._phase("forward_items")
            grad, scal, logits = be.forward_items(item_rows)
            self._end(t)
            t = self._phase("all_to_all grads (start) + reduce_users")
            grad_recv, work = c.all_to_all_start(grad, self._recv_buf(grad))   # by requester: rank order = fixed add order
            if after_grads_start is not None:
                after_grads_start()                                            # second part of the next batch's front end
            be.reduce_users(item_rows)
            self._end(t)
            t = self._phase("all_to_all grads (wait)")
            if work is not None:
                work.wait()
            self._end(t)
        else:
            t = self._phase("forward_reduce")
            grad, scal, logits = be.forward_reduce(item_rows)
            self._end(t)
            t = self._phase("all_to_all grads")
            grad_recv = c.all_to_all(grad)                                 # by requester: rank order = fixed add order
            self._end(t)
            if after_grads_start is not None:
                after_grads_start()
        t = self._phase("apply_items")
        be.apply_items(req_recv, grad_recv)
        self._end(t)
        t = self._phase("all_reduce + finish")
        scal = c.all_reduce_sum(scal)
        be.finish_step(scal)
        self._end(t)
        return logits, be.routed()["mine"], scal

    def _recv_buf(self, like):
        """a persistent receive buffer of the same shape (no allocation per step; an exchange in flight must not land in memory
        the caching allocator may hand out again)"""
        b = getattr(self, "_grad_recv", None)
        if b is None or b.shape != like.shape or b.device != like.device or b.dtype != like.dtype:
            b = self._grad_recv = torch.empty_like(like)
        return b

    def wire_bytes_per_step(self, batch_global):
        """bytes this rank sends over xGMI per step (requests + rows + gradient rows to the world-1 peers)"""
        _, slot_cap = self.capacities(batch_global)
        stride = packed_stride(self.D)
        return (self.world - 1) * slot_cap * (4 + 2 * stride * 4)

    def gather_global_tables(self):
        """All ranks' shards concatenated on every rank (tests / checkpoints)."""
        t = self.local_tables()
        out = {}
        for tid, rows, lo, hi in ((L.BU, self.U, self.u_lo, self.u_hi), (L.BI, self.I, self.i_lo, self.i_hi),
                                  (L.P, self.U, self.u_lo, self.u_hi), (L.Q, self.I, self.i_lo, self.i_hi)):
            part = np.asarray(t[tid])[: hi - lo]
            parts = [None] * self.world
            dist.all_gather_object(parts, part, group=self.comm.group)
            out[tid] = np.concatenate([x for x in parts if x.shape[0]], axis=0)
        out[L.MU] = np.asarray(t[L.MU])
        return out


# ------------------------------------------------------------------------------------ bench
def bench_entry(wl, K, W, rank, local_rank, world, workload_key="c3"):
    """bench.py --gpus N (N>1) for tables at the scale sharding is meant for: weak scaling - every rank owns 1/N of
    the rows and the global batch is N x the single-GPU batch, so per-GPU work is fixed.  Returns the JSON dict
    (rank 0 prints)."""
    import torch.cuda
    dev = torch.device("cuda", local_rank)
    # TFR_SHARD_SIDE_COMM=1: the next batch's small exchanges on a communicator of their own (they then never queue between this
    # step's row exchanges).  Default off: one communicator runs every collective of a rank in issue order, which cannot
    # deadlock whatever the placement of the RCCL kernels; the measured difference at world 1 is nil
    comm = Comm(side_group=dist.new_group() if os.environ.get("TFR_SHARD_SIDE_COMM") else None)
    U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
    Bg = B * world
    opts = dict(optimizer="adam", adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"])
    # capacities: the bench's ids are uniform by construction, so what a rank sends to one owner is Binomial(B, 1/W):
    # at B = 262144, W = 8 that is 32768 +- 169 - head-room of 10 % + 1024 slots is ~25 sigma; the class default (1.25) is for
    # skewed user ids.  Every exchanged byte is capacity, so the slack is wire volume.  Overflow would void the step loudly.
    slack = float(os.environ.get("TFR_SHARD_SLACK", "1.10"))
    m = ShardedSvd(U, I, D, comm, lambda ur, ir, d: HipShard(ur, ir, d, local_rank, **opts), device=dev, slack=slack)
    m.backend.model.init_tables(seed=13575 + rank)
    torch.cuda.set_stream(m.backend.stream)             # collectives queue behind the model's kernels: no fences
    # the same synthetic store and id stream on every rank (seeded), resident in HBM; the routing kernels read the user id of
    # every global sample from the rank's own copy (no communication) and take the whole record only of the samples it owns
    g = torch.Generator(device=dev)
    g.manual_seed(13575)
    N = min(wl["N"], 50_000_000)
    su = torch.randint(0, U, (N,), dtype=torch.int32, device=dev, generator=g)
    si = torch.randint(0, I, (N,), dtype=torch.int32, device=dev, generator=g)
    sr = torch.randint(1, 6, (N,), device=dev, generator=g).to(torch.float32)
    m.backend.set_store(su, si, sr)                      # every rank's own copy of the store; the kernels gather from it
    # pre-split batches with the id draw inside the loop, as at N=1: rank r draws ITS B rows per step with
    # np.random.seed(13575 + r); randint(0, N, (B,)) on the device generator (side stream, CH steps' ids per draw, two draws
    # ahead, three buffers); the records then travel to the owners of their user rows (train_step_local_ids).  A global stream
    # that every rank routes (train_step_ids) would make each rank generate and test world * B ids per step.
    np.random.seed(13575 + rank)
    be_model = m.backend.model
    be_model.rng_from_numpy()
    CH = 4
    ids_buf = torch.empty((3, CH * B), dtype=torch.int64, device=dev)
    issued, done = [0], [0]

    def issue():
        be_model.draw_ids_dev(N, CH * B, ids_buf[issued[0] % 3].data_ptr())
        issued[0] += 1
    issue(); issue()

    def ids_of(s):
        return ids_buf[(s // CH) % 3][(s % CH) * B:(s % CH + 1) * B]
    pipeline = not os.environ.get("TFR_SHARD_NO_PIPELINE")          # A/B switch: no front end of the next batch on the side stream

    def step(_s):
        s = done[0]
        if s % CH == 0:
            be_model.join_draw(s // CH + 1)              # this chunk's draw (issued two chunks ago), not the latest one
            issue()
        if pipeline and (s + 1) % CH == 0:
            be_model.join_draw((s + 1) // CH + 1)        # the next batch (first of the next chunk) is read during this step
        done[0] += 1
        return m.train_step_local_ids(ids_of(s), ids_of(s + 1) if pipeline else None)
    # untimed set-up before the W warm-up steps: one-off costs of a process's first collectives (a single 40 ms stall between
    # steps 10 and 20 at world 1: 1.33 ms per step in the steady state, 2.0 ms when it fell into a 40-step timed region)
    for s in range(SETUP_STEPS if W < SETUP_STEPS else 0):
        step(s % (W + K))
    m.backend.sync()
    for s in range(W):
        step(s)
    m.backend.sync()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        step(s)
    host_enqueue_s = time.perf_counter() - t0            # the host's share: when it equals the step time, the host is the bound
    m.backend.sync()
    torch.cuda.synchronize()
    comm.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if not comm.stage:
        el = el.to(dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # per-phase device times of a few more steps (events on the stream everything is queued on)
    m.timers = []
    for s in range(min(K, 10)):
        step(W + (s % K))
    torch.cuda.synchronize()
    phases = {}
    for name, e0, e1 in m.timers:
        phases[name] = phases.get(name, 0.0) + e0.elapsed_time(e1) * 1e3 / min(K, 10)
    m.timers = None
    wire = m.wire_bytes_per_step(Bg) + (world - 1) * m.pair_capacity(B) * 16      # + the sample records
    m.backend.sync()
    m.backend.model.close()                              # the shard's HBM goes back before the caller builds anything else
    step_s = elapsed / K
    exch_us = sum(v for k, v in phases.items() if k.startswith("all_to_all"))
    xgmi = wire / step_s / 1e9
    sample_cap, slot_cap = m.capacities(Bg)
    return dict(metric="training ratings/sec, %s, row-sharded over %d GPUs" % (wl["name"], world), value=K * Bg / elapsed,
                unit="ratings/s", n_gpus=world, steps=K, warmup=W, ms_per_step=step_s * 1e3,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload="%s: %s" % (workload_key, wl["name"]), untimed_setup_steps=SETUP_STEPS if W < SETUP_STEPS else 0, users=U, items=I, dim=D, global_batch=Bg, per_gpu_batch=B,
                            optimizer="adam", adam_mode=wl["adam_mode"], sample_cap=sample_cap, slot_cap=slot_cap, capacity_slack=slack,
                            parallelism="row-sharded tables x%d, pre-split batches (rank r draws its own B rows per step on the device, seed 13575 + r): "
                                        "device-side routing, 4 equal-split all-to-alls (sample records, request slots, packed rows, packed "
                                        "gradient rows) + one 16-byte all-reduce per step over RCCL, no host sync; the integer front end of batch "
                                        "s+1 (bucket, sample + request exchanges, routing, sorts) runs on a side stream beside step s, the user-side "
                                        "reduce beside the gradient exchange%s" % (world, "" if pipeline else " [front-end pipelining OFF]")),
                roofline=dict(kernel="all_to_all (%s)" % ("gloo rehearsal: staged through host memory, times meaningless" if comm.stage else "RCCL over xGMI"), bound="xgmi", achieved=xgmi, peak=XGMI_EGRESS_GBS, unit="GB/s",
                              frac=xgmi / XGMI_EGRESS_GBS, traffic=wire, algorithmic_bytes_per_step=wire,
                              exchange_us_per_step=exch_us, phases_us=phases, host_enqueue_us_per_step=host_enqueue_s / K * 1e6,
                              # north_star: "ratings/sec and achieved fraction of the HBM roofline" per GPU - SURVEY 8(d)'s 56D+104 bytes per
                              # rating of the lazy-Adam step x this GPU's B ratings per step, over the step time, against 8 TB/s
                              hbm_per_gpu=dict(algorithmic_bytes_per_step=B * (56 * D + 104), achieved_GBps=B * (56 * D + 104) / step_s / 1e9,
                                               peak=8000.0, frac=B * (56 * D + 104) / step_s / 1e9 / 8000.0),
                              note="egress bytes per rank and step (fixed-capacity slots: %d of them to each of %d peers, %d B each way per slot) "
                                   "over the whole step time; 7 links x 153 GB/s per GPU; with world=1 nothing crosses a link"
                                   % (slot_cap, world - 1, 2 * packed_stride(D) * 4 + 4)),
                cpu_baseline=None)
