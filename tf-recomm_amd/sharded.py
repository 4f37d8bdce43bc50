"""Row-sharded SVD training across the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI) for the exchanges, the HIP kernels of
this package for everything else.

Partition: block row-sharding - ``owner = id // ceil(rows / world)`` - of ``user_features`` /
``user_bias`` by user id and of ``item_features`` / ``item_bias`` by item id, Adam slots
co-located with their rows, ``bias_global`` replicated.

One step on the *global* batch (every rank sees the same ids; the reference has no distributed
path, the semantics are those of one ``minimize`` on the whole batch, ops.py:143-149):

  1. every rank keeps the samples whose USER row it owns (integer routing, no communication);
  2. it de-duplicates the item ids of its samples and asks their owners for the rows:
     all-to-all of ids (int32), owners gather, all-to-all of rows ``[n, D]`` + biases back;
  3. forward + backward locally (``tfr_shard_forward_reduce``): user rows are updated in place,
     item-row gradients - already reduced over the rank's own samples - come out per slot;
  4. all-to-all of the gradient rows back to the owners, which add them in rank order
     (deterministic) and apply the optimiser (``tfr_shard_apply_items``);
  5. all-reduce of 3 scalars (loss, regulariser, sum g) -> ``bias_global`` update everywhere.

Wire volume per rank and step is ~``2 * (D+1) * 4`` bytes per distinct non-local item row, each
way (fetching user rows as well would double it).  An all-to-all maps one-to-one onto the 7
direct xGMI links of a GPU; the path is xGMI-bound, not HBM-bound.

The compute backend is injected so the routing/exchange logic can be exercised on CPU with
``gloo`` (tests/test_sharded_cpu.py uses an oracle-backed stand-in; the product backend is
``HipShard`` below and has no CPU path).
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


def rows_per_rank(rows, world):
    return int(math.ceil(rows / world))


def shard_range(rows, world, rank):
    per = rows_per_rank(rows, world)
    lo = min(rows, rank * per)
    return lo, min(rows, lo + per)


class Comm(object):
    """all-to-all-v / all-reduce over ``torch.distributed``.  With a CPU-only backend (gloo)
    device tensors are staged through host memory - the rehearsal path; RCCL takes them as is."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.stage = dist.get_backend(group) == "gloo"

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        if self.stage and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def exchange_counts(self, send_counts):
        """send_counts: int64 [world] on the host -> recv_counts [world] on the host."""
        if self.stage:
            out = torch.empty_like(send_counts)
            dist.all_to_all_single(out, send_counts, group=self.group)
            return out
        dev = torch.device("cuda", torch.cuda.current_device())
        out = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(out, send_counts.to(dev), group=self.group)
        return out.cpu()

    def all_to_all_v(self, inp, send_counts, recv_counts):
        """rows of ``inp`` grouped by destination rank (send_counts[w] rows to rank w) ->
        rows grouped by source rank."""
        n_out = int(recv_counts.sum())
        out = torch.empty((n_out,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        return self._a2a(out, inp.contiguous(), [int(c) for c in recv_counts], [int(c) for c in send_counts])

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
        self.model.set_stream(self.stream.cuda_stream)

    def _sync_in(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _sync_out(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def gather_item_rows(self, ids_local):
        n = ids_local.numel()
        rows = torch.empty((n, self.D), dtype=torch.float32, device=self.device)
        bias = torch.empty((n,), dtype=torch.float32, device=self.device)
        self._sync_in()
        self.model.gather_item_rows(ids_local.data_ptr(), n, rows.data_ptr(), bias.data_ptr())
        self._sync_out()
        return rows, bias

    def forward_reduce(self, u_local, slot, rate, item_rows, item_bias):
        B, n = u_local.numel(), item_rows.shape[0]
        grad = torch.empty((n, self.D), dtype=torch.float32, device=self.device)
        bgrad = torch.empty((n,), dtype=torch.float32, device=self.device)
        logits = torch.empty((B,), dtype=torch.float32, device=self.device)
        scal = torch.zeros((4,), dtype=torch.float32, device=self.device)
        self._sync_in()
        self.model.shard_forward_reduce(u_local.data_ptr(), slot.data_ptr(), rate.data_ptr(), B,
                                        item_rows.data_ptr(), item_bias.data_ptr(), n, logits.data_ptr(),
                                        grad.data_ptr(), bgrad.data_ptr(), scal.data_ptr())
        self._sync_out()
        return grad, bgrad, scal, logits

    def apply_items(self, ids_local, grad, bgrad):
        self._sync_in()
        self.model.shard_apply_items(ids_local.data_ptr(), grad.data_ptr(), bgrad.data_ptr(), ids_local.numel())
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
    compute backend for the local shard."""

    def __init__(self, user_num, item_num, dim, comm, backend_factory, device="cpu"):
        self.U, self.I, self.D = int(user_num), int(item_num), int(dim)
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.per_u, self.per_i = rows_per_rank(self.U, self.world), rows_per_rank(self.I, self.world)
        self.u_lo, self.u_hi = shard_range(self.U, self.world, self.rank)
        self.i_lo, self.i_hi = shard_range(self.I, self.world, self.rank)
        self.device = torch.device(device)
        self.backend = backend_factory(self.u_hi - self.u_lo, self.i_hi - self.i_lo, self.D)
        self.last_plan = None

    # -- tables -------------------------------------------------------------------------
    def set_tables_from_global(self, mu, bu, bi, P, Q):
        """Every rank passes the full tables and keeps its slice (set_table stays a contiguous
        slice because the partition is by blocks)."""
        P, Q = np.asarray(P, np.float32), np.asarray(Q, np.float32)
        bu, bi = np.asarray(bu, np.float32), np.asarray(bi, np.float32)

        def sl(x, lo, hi, width=None):
            part = x[lo:hi]
            if part.shape[0] == 0:                      # an empty shard keeps one dummy row
                part = np.zeros((1,) + x.shape[1:], np.float32)
            return part
        self.backend.set_tables(np.float32(mu), sl(bu, self.u_lo, self.u_hi), sl(bi, self.i_lo, self.i_hi),
                                sl(P, self.u_lo, self.u_hi), sl(Q, self.i_lo, self.i_hi))

    def local_tables(self):
        return self.backend.tables()

    # -- routing (integer work, bit-exact) ----------------------------------------------
    def plan(self, u, i):
        """Which samples of the global batch this rank owns, their item slots and the request
        lists.  Pure integer torch ops; deterministic."""
        u, i = u.to(torch.int64), i.to(torch.int64)
        if u.numel() and (int(u.min()) < 0 or int(u.max()) >= self.U or int(i.min()) < 0 or int(i.max()) >= self.I):
            raise L.OutOfRangeError(L.ERR_OOB, "user/item id out of range [0,%d) / [0,%d)" % (self.U, self.I))
        mine = torch.nonzero(torch.div(u, self.per_u, rounding_mode="floor") == self.rank).reshape(-1)
        u_local = (u[mine] - self.u_lo).to(torch.int32)
        uniq, slot = torch.unique(i[mine], sorted=True, return_inverse=True)     # sorted => grouped by owner
        owner = torch.div(uniq, self.per_i, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=self.world).to(torch.int64).cpu()
        req_local = (uniq - owner * self.per_i).to(torch.int32)
        return dict(mine=mine, u_local=u_local, slot=slot.to(torch.int32), uniq=uniq, req_local=req_local,
                    send_counts=send_counts)

    # -- one step ------------------------------------------------------------------------
    def train_step(self, u, i, r):
        """``u, i, r``: the GLOBAL batch (identical on every rank), torch tensors on this rank's
        device.  Returns (logits of this rank's samples, their batch positions, global
        {loss, reg}) - the pre-update logits, like sess.run([train_op, logits])."""
        c, be = self.comm, self.backend
        p = self.plan(u, i)
        self.last_plan = p
        recv_counts = c.exchange_counts(p["send_counts"])
        req_recv = c.all_to_all_v(p["req_local"], p["send_counts"], recv_counts)     # ids asked of me
        rows_out, bias_out = be.gather_item_rows(req_recv)
        item_rows = c.all_to_all_v(rows_out, recv_counts, p["send_counts"])
        item_bias = c.all_to_all_v(bias_out, recv_counts, p["send_counts"])
        rate = r[p["mine"]].to(torch.float32).contiguous()
        grad, bgrad, scal, logits = be.forward_reduce(p["u_local"].contiguous(), p["slot"].contiguous(), rate,
                                                      item_rows, item_bias)
        grad_recv = c.all_to_all_v(grad, p["send_counts"], recv_counts)              # rank order = fixed add order
        bgrad_recv = c.all_to_all_v(bgrad, p["send_counts"], recv_counts)
        be.apply_items(req_recv, grad_recv, bgrad_recv)
        scal = c.all_reduce_sum(scal)
        be.finish_step(scal)
        return logits, p["mine"], scal

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
def bench_entry(wl, K, W, rank, local_rank, world):
    """bench.py --gpus N (N>1): weak scaling - every rank owns 1/N of the rows and the global batch
    is N x the single-GPU batch, so per-GPU work is fixed.  Returns the JSON dict (rank 0 prints)."""
    import torch.cuda
    dev = torch.device("cuda", local_rank)
    comm = Comm()
    U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
    Bg = B * world
    opts = dict(optimizer="adam", adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"])
    m = ShardedSvd(U, I, D, comm, lambda ur, ir, d: HipShard(ur, ir, d, local_rank, **opts), device=dev)
    m.backend.model.init_tables(seed=13575 + rank)
    # the same synthetic store and id stream on every rank (seeded), resident in HBM
    g = torch.Generator(device=dev)
    g.manual_seed(13575)
    N = min(wl["N"], 50_000_000)
    su = torch.randint(0, U, (N,), dtype=torch.int32, device=dev, generator=g)
    si = torch.randint(0, I, (N,), dtype=torch.int32, device=dev, generator=g)
    sr = torch.randint(1, 6, (N,), device=dev, generator=g).to(torch.float32)
    np.random.seed(13575)
    ids = torch.from_numpy(np.random.randint(0, N, (W + K, Bg))).to(dev)

    def step(s):
        sel = ids[s]
        return m.train_step(su[sel], si[sel], sr[sel])
    for s in range(W):
        step(s)
    m.backend.sync()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        step(s)
    m.backend.sync()
    torch.cuda.synchronize()
    comm.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if not comm.stage:
        el = el.to(dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    return dict(metric="training ratings/sec, MovieLens-1M SVD dim=64 @1 GPU (+ val RMSE)", value=K * Bg / elapsed,
                unit="ratings/s", n_gpus=world, steps=K, warmup=W, ms_per_step=elapsed / K * 1e3,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=wl["name"], users=U, items=I, dim=D, global_batch=Bg, per_gpu_batch=B,
                            optimizer="adam", adam_mode=wl["adam_mode"],
                            parallelism="row-sharded tables x%d, all-to-all row fetch + gradient return over RCCL" % world,
                            note="tables of this size fit one GPU thousands of times over: the exchange latency, not "
                                 "HBM, bounds the step (SURVEY 8e); --workload c3 is the scale sharding is meant for"),
                roofline=None, cpu_baseline=None)
