"""Data-parallel SVD training for tables that every GPU can hold (MovieLens scale): replicated
tables, each rank reduces its own minibatch to dense gradient buffers, ONE fixed-size
all-reduce per step (RCCL over xGMI - no host synchronisation, no variable-size exchange), then
the same dense optimiser step on every rank.  Mathematically this is one ``minimize``
(ops.py:143-149) on the union of the ranks' batches; the replicas stay bit-identical because
every rank applies the same all-reduced buffer.

Needs dense optimiser semantics: Adam "tf1" (what tf.train.AdamOptimizer does) or
SGD.  Tables that outgrow one GPU use ``sharded.py`` instead.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist

SETUP_STEPS = 30     # untimed steps ahead of the warm-up in bench_entry (one-off costs of a process's first collectives)

from .engine import SvdModel


class HipReplica(object):
    """Product backend: one full replica in HBM behind the C-ABI."""

    def __init__(self, user_num, item_num, dim, device, **opts):
        self.device = torch.device("cuda", device)
        self.model = SvdModel(user_num, item_num, dim, device=device, **opts)
        self.stream = torch.cuda.Stream(device=self.device)
        self.model.set_stream(self.stream.cuda_stream)
        self.flat = torch.zeros(self.model.dp_flat_size(), dtype=torch.float32, device=self.device)

    # The model's kernels run on self.stream.  When the caller has made that torch's current
    # stream (bench_entry does: torch.cuda.set_stream) the collective orders itself against it
    # and no cross-stream events are needed; otherwise fence both ways.
    def _foreign(self):
        cur = torch.cuda.current_stream(self.device)
        return None if cur == self.stream else cur

    def local_grads(self, u=None, i=None, r=None, store_ids_ptr=None, batch=None, next_ids_ptr=None):
        cur = self._foreign()
        if cur is not None:
            self.stream.wait_stream(cur)
        if store_ids_ptr is not None and next_ids_ptr is not None:
            self.model.dp_hint_next(next_ids_ptr)
        if store_ids_ptr is not None:
            self.model.dp_local_grads(None, None, None, batch, store_ids_ptr, self.flat.data_ptr())
        else:
            self.model.dp_local_grads(u.data_ptr(), i.data_ptr(), r.data_ptr(), u.numel(), None, self.flat.data_ptr())
        if cur is not None:
            cur.wait_stream(self.stream)
        return self.flat

    def apply(self, flat):
        cur = self._foreign()
        if cur is not None:
            self.stream.wait_stream(cur)
        self.model.dp_apply(flat.data_ptr())
        if cur is not None:
            cur.wait_stream(self.stream)

    def sync(self):
        self.model.sync()


class DataParallelSvd(object):
    def __init__(self, backend, group=None):
        self.backend, self.group = backend, group
        self.stage = dist.get_backend(group) == "gloo"

    def _all_reduce(self, flat):
        if self.stage and flat.is_cuda:                 # rehearsal path: gloo has no device collectives
            c = flat.cpu()
            dist.all_reduce(c, group=self.group)
            flat.copy_(c)
        else:
            dist.all_reduce(flat, group=self.group)
        return flat

    def train_step(self, u=None, i=None, r=None, store_ids_ptr=None, batch=None, want_scalars=True, next_ids_ptr=None):
        """This rank's slice of the global batch.  Returns a copy of the flat buffer's tail
        {loss, reg, sum_g, 0} (global sums), or None with ``want_scalars=False``."""
        if next_ids_ptr is not None:
            flat = self.backend.local_grads(u, i, r, store_ids_ptr, batch, next_ids_ptr)
        else:
            flat = self.backend.local_grads(u, i, r, store_ids_ptr, batch)
        self._all_reduce(flat)
        scal = flat[-4:].clone() if want_scalars else None
        self.backend.apply(flat)
        return scal


def bench_entry(wl, K, W, rank, local_rank, world, train, val, workload_key="c2"):
    """bench.py --gpus N (N>1) for tables that fit every GPU: weak scaling, global batch N x B."""
    dev = torch.device("cuda", local_rank)
    U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
    be = HipReplica(U, I, D, local_rank, optimizer="adam", adam_mode="tf1", lr=wl["lr"], reg=wl["reg"])
    be.model.init_tables(seed=13575)                   # counter-based initialiser: identical replicas
    be.model.upload_triples(*train)
    # the id draw is inside the loop, as at N=1: rank r draws ITS B ids per step with np.random.seed(13575 + r);
    # randint(0, N, (B,)) (svd_train_val.py:15, dataio.py:115; rank 0 = the reference's own stream) on the device generator
    # (rng.hip, side stream, eight steps' ids per draw, two draws ahead, three buffers) - no host draw, no id upload, no
    # communication.  One global stream cut into slices would make every rank generate all N * B ids per step (MT19937
    # cannot skip ahead cheaply): 52 us per step at N = 8, the next bound after the all-reduce; independent streams keep
    # the draw at 1/N of that.  The reference has no multi-GPU definition to follow (README.md:10).
    ntrain = len(train[0])
    np.random.seed(13575 + rank)
    be.model.rng_from_numpy()
    CH = 8                                             # steps per draw: one launch + two events per CH steps, not per step
    Bg = B                                             # ids this rank draws per step
    ids_buf = torch.empty((3, CH * Bg), dtype=torch.int64, device=dev)
    dp = DataParallelSvd(be)
    stage = dp.stage
    torch.cuda.set_stream(be.stream)                   # the collective queues behind the model's kernels
    issued = [0]

    def issue():                                       # chunk c = the ids of steps [c * CH, (c + 1) * CH)
        be.model.draw_ids_dev(ntrain, CH * Bg, ids_buf[issued[0] % 3].data_ptr())
        issued[0] += 1
    issue(); issue()
    done = [0]

    def ids_ptr(s):
        return ids_buf[(s // CH) % 3].data_ptr() + (s % CH) * Bg * 8

    def step(_s):
        s = done[0]
        if s % CH == 0:
            # chunk s / CH (its own draw event) and, because the step hints the next batch's ids, chunk s / CH + 1 when the
            # chunk ends - joined there; never the latest draw (one shared event made every join wait for it)
            be.model.join_draw(s // CH + 1)
            issue()                                    # chunk s / CH + 2: into the buffer chunk s / CH - 1 lived in
        if (s + 1) % CH == 0:
            be.model.join_draw((s + 1) // CH + 1)      # ids_ptr(s + 1) is hinted to this step's launch
        dp.train_step(store_ids_ptr=ids_ptr(s), batch=B, want_scalars=False, next_ids_ptr=ids_ptr(s + 1))
        done[0] += 1
    # untimed set-up before the W warm-up steps: the first few dozen collectives of a process pay one-off costs (RCCL channel
    # and buffer set-up, allocator growth - a single 40 ms stall was seen between steps 10 and 20 of the row-sharded loop);
    # they are paid here, on the first batches, and not inside a short timed region
    for s in range(SETUP_STEPS if W < SETUP_STEPS else 0):
        step(s % max(1, W + K - 1))
    be.sync()
    for s in range(W):
        step(s)
    be.sync()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        step(s)
    be.sync()
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if not stage:
        el = el.to(dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # per-phase device times of a few more steps (events on the stream everything is queued on)
    n_t = min(K, 20)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_t)]
    for k in range(n_t):
        s = done[0]
        if s % CH == 0:
            be.model.join_draw(s // CH + 1)
            issue()
        ev[k][0].record()
        flat = be.local_grads(None, None, None, ids_ptr(s), B)
        done[0] += 1
        ev[k][1].record()
        dp._all_reduce(flat)
        ev[k][2].record()
        be.apply(flat)
        ev[k][3].record()
    torch.cuda.synchronize()
    phases = {name: sum(ev[k][j].elapsed_time(ev[k][j + 1]) for k in range(n_t)) * 1e3 / n_t
              for j, name in enumerate(("local_grads", "all_reduce", "apply"))}
    sse, _ = be.model.eval(*val)
    nbytes = be.flat.numel() * 4
    wire = 2.0 * (world - 1) / world * nbytes            # ring / tree all-reduce: bytes each rank sends
    step_s = elapsed / K
    xgmi = wire / step_s / 1e9
    from .sharded import XGMI_EGRESS_GBS
    return dict(metric="training ratings/sec + val RMSE, MovieLens-1M SVD dim=64, data-parallel over %d GPUs" % world
                if workload_key == "c2" else "training ratings/sec, %s, data-parallel over %d GPUs" % (wl["name"], world),
                value=K * B * world / elapsed, unit="ratings/s", n_gpus=world, steps=K, warmup=W,
                ms_per_step=step_s * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f32", data="synthetic",
                config=dict(workload="%s: %s" % (workload_key, wl["name"]), untimed_setup_steps=SETUP_STEPS if W < SETUP_STEPS else 0, users=U, items=I, dim=D, global_batch=B * world, per_gpu_batch=B,
                            optimizer="adam", adam_mode="tf1", lr=wl["lr"], reg=wl["reg"],
                            id_stream="rank r: np.random.seed(13575 + r); randint(0, N, (B,)) per step, drawn inside the timed loop on the device",
                            parallelism="dp%d: replicated tables, one %.1f MB gradient all-reduce (RCCL) per step"
                                        % (world, nbytes / 1e6)),
                val_rmse=float(np.sqrt(sse / len(val[0]))),
                roofline=dict(kernel="all_reduce (%s)" % ("gloo rehearsal: staged through host memory, times meaningless" if stage else "RCCL over xGMI"), bound="xgmi", achieved=xgmi, peak=XGMI_EGRESS_GBS, unit="GB/s",
                              frac=xgmi / XGMI_EGRESS_GBS, traffic=wire, algorithmic_bytes_per_step=wire, phases_us=phases,
                              # per GPU against HBM (SURVEY 8(d), tf1 Adam: 8D+56 per rating + the dense term 24 (U+I)(D+1) per step)
                              hbm_per_gpu=dict(algorithmic_bytes_per_step=B * (8 * D + 56) + 24 * (U + I) * (D + 1),
                                               achieved_GBps=(B * (8 * D + 56) + 24 * (U + I) * (D + 1)) / step_s / 1e9, peak=8000.0,
                                               frac=(B * (8 * D + 56) + 24 * (U + I) * (D + 1)) / step_s / 1e9 / 8000.0),
                              note="bytes each rank sends per step, 2 (N-1)/N x the %.1f MB gradient buffer, over the whole step time; the "
                                   "2.6 MB model makes this step latency-bound (collective launch + a few link hops), not bandwidth-bound; "
                                   "with world=1 nothing crosses a link" % (nbytes / 1e6)),
                cpu_baseline=None)
