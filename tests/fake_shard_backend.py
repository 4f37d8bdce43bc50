"""Oracle-backed stand-in for tfrecomm_amd.sharded.HipShard (TEST ONLY): the same four calls,
computed with oracle/svd_oracle.py in float64 on CPU tensors, so the routing / exchange logic of
sharded.py can run under gloo without a GPU."""
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

    def gather_item_rows(self, ids_local):
        ids = ids_local.numpy().astype(np.int64)
        return torch.from_numpy(self.o.Q[ids].copy()), torch.from_numpy(self.o.bi[ids].copy())

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

    def forward_reduce(self, u_local, slot, rate, item_rows, item_bias):
        o = self.o
        u = u_local.numpy().astype(np.int64)
        s = slot.numpy().astype(np.int64)
        r = rate.numpy().astype(np.float64)
        Qf, bif = item_rows.numpy().astype(np.float64), item_bias.numpy().astype(np.float64)
        logits = so.forward(o.P, Qf, o.bu, bif, o.mu, u, s, o.item_abs)
        g = so.dlogits(logits, r, o.loss)
        dP, dQ, dbu, dbi, dmu = so.occurrence_grads(o.P, Qf, o.bu, bif, u, s, g, o.reg, o.item_abs, o.reg_bias)
        loss = so.data_loss(logits, r, o.loss) if u.size else 0.0
        reg = so.regularizer(o.P, Qf, o.bu, bif, u, s, o.reg_bias) if u.size else 0.0
        n = Qf.shape[0]
        grad = np.zeros((n, o.D))
        bgrad = np.zeros(n)
        np.add.at(grad, s, dQ)
        np.add.at(bgrad, s, dbi)
        self._apply(so.PF, o.P, u, dP)
        self._apply(so.BU, o.bu, u, dbu)
        scal = torch.tensor([loss, reg, dmu, 0.0], dtype=torch.float64)
        return torch.from_numpy(grad), torch.from_numpy(bgrad), scal, torch.from_numpy(logits)

    def apply_items(self, ids_local, grad, bgrad):
        ids = ids_local.numpy().astype(np.int64)
        self._apply(so.QF, self.o.Q, ids, grad.numpy().astype(np.float64))
        self._apply(so.BI, self.o.bi, ids, bgrad.numpy().astype(np.float64))

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
