"""Oracle-backed stand-in for tfrecomm_amd.sharded.HipShard (TEST ONLY): the same calls (route, gather,
forward_reduce, apply_items, finish_step), computed with NumPy / oracle/svd_oracle.py in float64 on CPU
tensors, so the exchange logic of sharded.py can run under gloo without a GPU; its `route` is also the
reference the device routing kernels (csrc/shard.hip) are compared with on the GPU."""
import numpy as np
import torch

from oracle import svd_oracle as so


class OracleShard(object):
    def __init__(self, u_rows, i_rows, dim, loss="mse", item_abs=False, reg_bias=False, optimizer="adam",
                 adam_mode="tf1", lr=1e-3, reg=0.05):
        self.o = so.SvdOracle(max(1, u_rows), max(1, i_rows), dim, loss=loss, item_abs=item_abs, reg_bias=reg_bias,
                              optimizer=optimizer, adam_mode=adam_mode, lr=lr, reg=reg, dtype=np.float64)

    def set_tables(self, mu, bu, bi, P, Q):
        self.o.set_tables(np.float64(mu), np.asarray(bu, np.float64), np.asarray(bi, np.float64),
                          np.asarray(P, np.float64), np.asarray(Q, np.float64))

    def tables(self):
        return {k: np.array(v) for k, v in self.o.tables().items()}

    def sync(self):
        pass

    # ---- the two routed-batch sets of the product backend (tfr_shard_select / _presort): ShardedSvd prepares batch s+1 in the
    #      other set while step s uses its own; here the "side stream" is just the call order
    def select(self, which):
        self._sets = getattr(self, "_sets", {})
        if hasattr(self, "_r"):
            self._sets[getattr(self, "_set", 0)] = self._r
        self._set = int(which)
        if self._set in self._sets:
            self._r = self._sets[self._set]

    def presort(self, req_recv):
        pass

    def on_stream(self, stream):
        pass

    # ---- routing: the NumPy statement of csrc/shard.hip (what the device kernels must reproduce bit for bit)
    def route(self, u, i, r, rank, world, U, I, sample_cap, slot_cap):
        u, i, r = u.numpy().astype(np.int64), i.numpy().astype(np.int64), r.numpy().astype(np.float64)
        if u.size and (u.min() < 0 or u.max() >= U or i.min() < 0 or i.max() >= I):
            raise IndexError("user/item id out of range")
        per_u, per_i = -(-U // world), -(-I // world)
        own = np.flatnonzero(u // per_u == rank)
        n = own.size
        if n > sample_cap:
            raise IndexError("sample capacity exceeded")
        uq, inv = np.unique(i[own], return_inverse=True)           # sorted: grouped by owner
        owner = uq // per_i
        cnt = np.bincount(owner, minlength=world)
        if cnt.max(initial=0) > slot_cap:
            raise IndexError("slot capacity exceeded")
        start = np.cumsum(cnt) - cnt
        slot_of_uq = owner * slot_cap + (np.arange(uq.size) - start[owner])
        req = np.full(world * slot_cap, -1, np.int32)
        req[slot_of_uq] = (uq - owner * per_i).astype(np.int32)
        mine = np.full(sample_cap, -1, np.int32); mine[:n] = own
        u_local = np.full(sample_cap, self.o.U, np.int32); u_local[:n] = u[own] - rank * per_u
        slot = np.full(sample_cap, world * slot_cap, np.int32); slot[:n] = slot_of_uq[inv]
        self._r = dict(n=n, u=u[own] - rank * per_u, slot=slot_of_uq[inv], rate=r[own], nslots=world * slot_cap,
                       mine=torch.from_numpy(mine), u_local=torch.from_numpy(u_local), slot_t=torch.from_numpy(slot),
                       counts=torch.from_numpy(np.concatenate([[n, uq.size], cnt]).astype(np.int32)))
        return torch.from_numpy(req)

    def set_store(self, u, i, r):
        self._store = (u.clone(), i.clone(), r.clone())

    def route_ids(self, ids, rank, world, U, I, sample_cap, slot_cap):
        su, si, sr = self._store
        sel = ids.long()
        return self.route(su[sel], si[sel], sr[sel], rank, world, U, I, sample_cap, slot_cap)

    def bucket_ids(self, ids, world, U, pair_cap):
        su, si, sr = self._store
        sel = ids.long()
        u, i, r = su[sel].numpy().astype(np.int64), si[sel].numpy().astype(np.int64), sr[sel].numpy().astype(np.float32)
        per_u = -(-U // world)
        send = np.full((world * pair_cap, 4), -1, np.int32)
        owner = u // per_u
        for w in range(world):
            k = np.flatnonzero(owner == w)                 # batch order inside a group
            if k.size > pair_cap:
                raise IndexError("pair capacity exceeded")
            send[w * pair_cap: w * pair_cap + k.size, 0] = u[k]
            send[w * pair_cap: w * pair_cap + k.size, 1] = i[k]
            send[w * pair_cap: w * pair_cap + k.size, 2] = r[k].view(np.int32)
            send[w * pair_cap: w * pair_cap + k.size, 3] = k
        return torch.from_numpy(send)

    def route_recs(self, recv, rank, world, U, I, sample_cap, slot_cap):
        rec = recv.numpy()
        ok = np.flatnonzero(rec[:, 0] >= 0)
        u = torch.from_numpy(rec[ok, 0].astype(np.int32)); i = torch.from_numpy(rec[ok, 1].astype(np.int32))
        r = torch.from_numpy(rec[ok, 2].copy().view(np.float32))
        req = self.route(u, i, r, rank, world, U, I, sample_cap, slot_cap)
        # positions refer to the received buffer, as on the device
        n = self._r["n"]
        mine = self._r["mine"].numpy().copy()
        mine[:n] = ok[mine[:n]]
        self._r["mine"] = torch.from_numpy(mine)
        return req

    def routed(self):
        return dict(mine=self._r["mine"], u_local=self._r["u_local"], slot=self._r["slot_t"], counts=self._r["counts"])

    @property
    def stride(self):
        D = self.o.D
        return D + 4 if D % 4 == 0 else D + 2

    def gather(self, req_recv):
        ids = req_recv.numpy().astype(np.int64)
        out = np.zeros((ids.size, self.stride))
        ok = ids >= 0
        out[ok, : self.o.D] = self.o.Q[ids[ok]]
        out[ok, self.o.D] = self.o.bi[ids[ok]]
        return torch.from_numpy(out)

    def _apply(self, tid, var, ids, occ):
        o = self.o
        if (o.frozen >> tid) & 1:
            return
        if o.optimizer == so.SGD:
            so.sgd_sparse(var, ids, occ, o.lr)
            return
        uniq, inv = so.dedup(ids)
        gsum = so.segment_sum(occ, inv, uniq.size)
        fn = so.adam_sparse_tf1 if o.adam_mode == so.TF1 else so.adam_sparse_lazy
        fn(var, o.slots[tid], uniq, gsum, o.lr, o.b1p, o.b2p, o.b1, o.b2, o.eps)

    def forward_reduce(self, item_rows):
        o, R = self.o, self._r
        u, s, r = R["u"].astype(np.int64), R["slot"].astype(np.int64), R["rate"]
        rows = item_rows.numpy().astype(np.float64)
        Qf, bif = np.ascontiguousarray(rows[:, : o.D]), np.ascontiguousarray(rows[:, o.D])
        logits = so.forward(o.P, Qf, o.bu, bif, o.mu, u, s, o.item_abs)
        g = so.dlogits(logits, r, o.loss)
        dP, dQ, dbu, dbi, dmu = so.occurrence_grads(o.P, Qf, o.bu, bif, u, s, g, o.reg, o.item_abs, o.reg_bias)
        loss = so.data_loss(logits, r, o.loss) if u.size else 0.0
        reg = so.regularizer(o.P, Qf, o.bu, bif, u, s, o.reg_bias) if u.size else 0.0
        grad = np.zeros((R["nslots"], self.stride))
        np.add.at(grad[:, : o.D], s, dQ)
        np.add.at(grad[:, o.D], s, dbi)
        self._apply(so.PF, o.P, u, dP)
        self._apply(so.BU, o.bu, u, dbu)
        scal = torch.tensor([loss, reg, dmu, 0.0], dtype=torch.float64)
        lg = np.zeros(R["mine"].numel())
        lg[: R["n"]] = logits
        return torch.from_numpy(grad), scal, torch.from_numpy(lg)

    def apply_items(self, req_recv, grad_recv):
        ids = req_recv.numpy().astype(np.int64)
        ok = ids >= 0                                               # buffer order = rank order of the requesters
        g = grad_recv.numpy().astype(np.float64)
        self._apply(so.QF, self.o.Q, ids[ok], np.ascontiguousarray(g[ok, : self.o.D]))
        self._apply(so.BI, self.o.bi, ids[ok], np.ascontiguousarray(g[ok, self.o.D]))

    def finish_step(self, scal):
        o = self.o
        dmu = np.float64(scal[2].item())
        if not (o.frozen >> so.MU) & 1:
            if o.optimizer == so.SGD:
                o.mu -= o.lr * dmu
            else:
                so.adam_dense(o.mu, o.slots[so.MU], dmu, o.lr, o.b1p, o.b2p, o.b1, o.b2, o.eps)
        if o.optimizer != so.SGD:
            o.b1p = o.b1p * o.b1
            o.b2p = o.b2p * o.b2
        o.step += 1
