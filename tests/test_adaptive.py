"""The per-user fine-tuning drivers (tf-recomm_amd/adaptive_test.py, cats.py; reference adaptive_test.py,
non_adaptive_test.py, cats.py).  CPU part: the driver logic on the float64 oracle behind the model interface, the
selectors, the rank-sum AUC against sklearn.  GPU part (-m gpu): tfr_train_steps_repeat == that many tfr_train_step calls
bit for bit, and the drivers on the HIP path against the same drivers on the oracle.  Parity unpinned by reference
fixtures: the reference scripts need TensorFlow and a checkpoint that does not ship."""
import numpy as np
import pytest

import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
from tfrecomm_amd import adaptive_test as AT
from tfrecomm_amd import cats
from oracle import svd_oracle as so
from tests.util import assert_close, make_oracle, rand_tables


class OracleModel(object):
    """the oracle behind the three calls the drivers make of an SvdModel"""
    def __init__(self, U, I, D, tables, **kw):
        self.loss = kw.get("loss", "mse")
        self.o = make_oracle(U, I, D, tables, **kw)
        self.calls = []

    def set_frozen(self, mask):
        self.o.frozen = mask

    def forward(self, u, i):
        return np.asarray(self.o.forward(np.asarray(u, np.int32), np.asarray(i, np.int32)), np.float64)

    def train_steps_repeat(self, u, i, r, nsteps, want_logits=True, want_loss=True):
        u, i, r = np.asarray(u, np.int32), np.asarray(i, np.int32), np.asarray(r, np.float32)
        self.calls.append((u.copy(), i.copy(), r.copy(), nsteps))
        loss = np.empty(nsteps, np.float64)
        logits = None
        for s in range(nsteps):
            logits, loss[s], _ = self.o.train_step(u, i, r)
        return np.asarray(logits), loss


def _frame(rs, U, I, n, binary):
    import pandas as pd
    u = np.sort(rs.randint(0, U, n))                     # the reference's test frame is grouped by user
    i = rs.randint(0, I, n)
    r = (rs.rand(n) < 0.5).astype(np.float32) if binary else rs.randint(1, 6, n).astype(np.float32)
    return pd.DataFrame(dict(user=u.astype(np.int32), item=i.astype(np.int32), outcome=r, wins=0.0, fails=0.0))


def test_roc_auc_is_sklearn_s():
    from sklearn.metrics import roc_auc_score
    rs = np.random.RandomState(5)
    for n in (2, 7, 100, 1000):
        y = (rs.rand(n) < 0.4).astype(np.float32)
        y[0], y[1] = 0, 1
        s = np.round(rs.rand(n), 2 if n > 7 else 1)      # ties
        assert abs(AT.roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12
    assert np.isnan(AT.roc_auc([1, 1], [0.2, 0.3]))


def test_selectors():
    c = cats.Next([5, 3, 9])
    assert [c.next_item(), c.next_item()] == [5, 3] and c.asked == [5, 3] and c.available_item_ids == [9]
    pop = np.array([0, 1, 7, 3, 9, 2])
    c = cats.Popular([1, 2, 3, 5], pop)
    assert [c.next_item() for _ in range(4)] == [2, 3, 5, 1]
    import random
    random.seed(3)
    c = cats.Random(range(10))
    got = [c.next_item() for _ in range(10)]
    assert sorted(got) == list(range(10)) and c.asked == got and len(c) == 0
    with pytest.raises(NotImplementedError):
        cats.Fisher([1, 2])


def test_non_adaptive_flow_on_the_oracle():
    U, I, D = 12, 9, 4
    rs = np.random.RandomState(0)
    t = rand_tables(rs, U, I, D)
    df = _frame(rs, U, I, 30, binary=True)
    m = OracleModel(U, I, D, t, loss="nll", optimizer="sgd", lr=0.05, reg=0.1)
    q0 = m.o.tables()[so.QF].copy()
    seen = []
    res = AT.non_adaptive_test(m, df, epoch_max=7, max_user=8, log=seen.append)
    kept = int((df["user"] <= 8).sum())                  # rows before the first user id above max_user
    assert len(res["pred"]) == kept == len(m.calls) == len(seen)
    # each call trains on the user's history so far, one more row than that user's previous call
    per_user = {}
    for (u, i, r, n), (_, row) in zip(m.calls, df.iterrows()):
        per_user[int(row["user"])] = per_user.get(int(row["user"]), 0) + 1
        assert n == 7 and len(u) == per_user[int(row["user"])] and set(u.tolist()) == {int(row["user"])}
        assert i[-1] == int(row["item"]) and r[-1] == row["outcome"]
    assert np.array_equal(m.o.tables()[so.QF], q0)       # var_list=[user_bias, user_features]: items never move
    assert 0.0 <= res["accuracy"] <= 1.0
    assert res["truth"] == [float(x) for x in df["outcome"][:kept]]


def test_adaptive_flow_on_the_oracle():
    U, I, D = 6, 40, 3
    rs = np.random.RandomState(1)
    t = rand_tables(rs, U, I, D)
    import pandas as pd
    rows = [(u, i, float((u + i) % 2)) for u in (4, 2, 5, 0) for i in rs.permutation(I)[:12]]
    df = pd.DataFrame(rows, columns=["user", "item", "outcome"])
    m = OracleModel(U, I, D, t, loss="nll", optimizer="adam", adam_mode="tf1", lr=5e-3, reg=0.0)
    out = AT.adaptive_test(m, df, budget=5, epoch_max=4, max_users=3)
    assert [r["user"] for r in out] == [4, 2, 5]          # first-appearance order, first three users
    for k, rec in enumerate(out):
        mine = df[df["user"] == rec["user"]]
        assert rec["asked"] == list(mine["item"][:5]) and rec["size"] == 5          # cats.Next: in the frame's order
        assert rec["outcome"] == [float(x) for x in mine["outcome"][:5]]
        calls = m.calls[5 * k:5 * k + 5]
        assert [len(c[0]) for c in calls] == [1, 2, 3, 4, 5] and all(c[3] == 4 for c in calls)
    with pytest.raises(ValueError):
        AT.adaptive_test(m, df, budget=13, epoch_max=1)


# ------------------------------------------------------------------ GPU
def _model(U, I, D, t, **kw):
    m = T.SvdModel(U, I, D, **kw)
    m.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("U,I,D,B,opt,mode,loss,frozen", [
    (300, 200, 20, 7, "adam", "tf1", "nll", AT.FROZEN_BUT_USER),
    (300, 200, 20, 1, "sgd", "tf1", "mse", AT.FROZEN_BUT_USER),
    (6040, 3952, 64, 50, "adam", "tf1", "mse", 0),
    (30000, 20000, 32, 300, "adam", "lazy", "mse", AT.FROZEN_BUT_USER),
])
def test_repeat_is_that_many_single_steps_bit_for_bit(U, I, D, B, opt, mode, loss, frozen):
    rs = np.random.RandomState(B)
    t = rand_tables(rs, U, I, D)
    u = rs.randint(0, U, B).astype(np.int32)
    if frozen:
        u[:] = u[0]
    i = rs.randint(0, I, B).astype(np.int32)
    r = (rs.rand(B) < 0.5).astype(np.float32) if loss == "nll" else rs.randint(1, 6, B).astype(np.float32)
    kw = dict(loss=loss, optimizer=opt, adam_mode=mode, lr=5e-3, reg=0.1)
    n = 9
    with _model(U, I, D, t, **kw) as a, _model(U, I, D, t, **kw) as b:
        a.set_frozen(frozen); b.set_frozen(frozen)
        logits, loss_a = a.train_steps_repeat(u, i, r, n)
        loss_b = np.empty(n, np.float32)
        for s in range(n):
            lg, loss_b[s], _ = b.train_step(u, i, r)
        assert np.array_equal(logits, lg) and np.array_equal(loss_a, loss_b)
        ta, tb = a.tables(), b.tables()
        for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
            assert np.array_equal(ta[tid], tb[tid])
        assert a.step == b.step == n
        with pytest.raises(T.OutOfRangeError):
            a.train_steps_repeat(np.array([U], np.int32), i[:1], r[:1], 3)
        assert a.step == n                                 # a bad id voids the whole call
        for tid in (L.P, L.BU):
            assert np.array_equal(a.tables()[tid], ta[tid])


@pytest.mark.gpu
@pytest.mark.parametrize("opt,tol", [("sgd", 2e-4), ("adam", 5e-3)])
def test_drivers_on_the_hip_path_match_the_oracle(opt, tol):
    U, I, D = 40, 60, 8
    rs = np.random.RandomState(11)
    t = rand_tables(rs, U, I, D)
    df = _frame(rs, U, I, 60, binary=True)
    kw = dict(loss="nll", optimizer=opt, adam_mode="tf1", lr=5e-3 if opt == "adam" else 0.05, reg=0.1)
    orc = OracleModel(U, I, D, t, **kw)
    want = AT.non_adaptive_test(orc, df, epoch_max=25, max_user=30)
    with _model(U, I, D, t, **kw) as m:
        got = AT.non_adaptive_test(m, df, epoch_max=25, max_user=30)
        assert len(got["pred"]) == len(want["pred"]) > 20
        assert_close(got["pred"], want["pred"], rtol=tol, what="predictions made between the fine-tuning rounds")
        tabs = m.tables()
        assert_close(tabs[L.P], orc.o.tables()[so.PF], rtol=10 * tol, what="user features")
        assert np.array_equal(tabs[L.Q], t["Q"]) and np.array_equal(tabs[L.BI], t["bi"])
    import pandas as pd
    rows = [(u, i, float(rs.rand() < 0.5)) for u in (7, 3, 21) for i in rs.permutation(I)[:10]]
    df2 = pd.DataFrame(rows, columns=["user", "item", "outcome"])
    orc = OracleModel(U, I, D, t, **kw)
    want = AT.adaptive_test(orc, df2, budget=6, epoch_max=30)
    with _model(U, I, D, t, **kw) as m:
        got = AT.adaptive_test(m, df2, budget=6, epoch_max=30)
    for g, w in zip(got, want):
        assert g["asked"] == w["asked"] and g["outcome"] == w["outcome"]
        assert_close(g["predicted"], w["predicted"], rtol=tol, what="predictions of user %d" % g["user"])
        assert abs(g["mcost"] - w["mcost"]) <= tol * max(1.0, abs(w["mcost"]))


@pytest.mark.gpu
def test_a_loop_spelled_like_non_adaptive_test_gives_the_drivers_numbers():
    """The reference's own spelling of the loop (non_adaptive_test.py:20-31,36-37,56-87: placeholders, the fork's
    inference_svd / optimization with var_list=[user_bias, user_features], one sess.run per epoch) through the Session
    mirror, against tfrecomm_amd.adaptive_test.non_adaptive_test on an identical model: same predictions bit for bit (the
    driver's one-call inner loop is the same arithmetic), item tables untouched."""
    from tfrecomm_amd import graph as tf, ops
    U, I, D, EPOCH_MAX = 30, 40, 8, 12
    rs = np.random.RandomState(3)
    t = rand_tables(rs, U, I, D)
    df = _frame(rs, U, I, 25, binary=True)
    tf.reset_default_graph()
    user_batch = tf.placeholder(tf.int32, shape=[None], name="id_user")
    item_batch = tf.placeholder(tf.int32, shape=[None], name="id_item")
    rate_batch = tf.placeholder(tf.float32, shape=[None])
    wins_batch = tf.placeholder(tf.float32, shape=[None], name="nb_wins")
    fails_batch = tf.placeholder(tf.float32, shape=[None], name="nb_fails")
    infer, logits, regularizer, user_bias, user_features, item_bias, item_features = ops.inference_svd(
        user_batch, item_batch, wins_batch, fails_batch, user_num=U, item_num=I, dim=D, device="/cpu:0", fork_semantics=True)
    tf.train.get_or_create_global_step()
    cost, train_op = ops.optimization(infer, logits, regularizer, rate_batch, learning_rate=5e-3, reg=0.1, device="/cpu:0",
                                      var_list=[user_bias, user_features])
    preds = []
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        sess.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        kw = dict(loss=sess.model.loss, item_abs=sess.model.item_abs, reg_bias=sess.model.reg_bias, optimizer=sess.model.optimizer,
                  adam_mode=sess.model.adam_mode, lr=5e-3, reg=0.1)
        hist = {}
        for user_id, item_id, outcome in zip(df["user"], df["item"], df["outcome"]):
            h = hist.setdefault(int(user_id), ([], [], []))
            h[0].append(user_id); h[1].append(item_id); h[2].append(outcome)
            item_logit = sess.run(logits, feed_dict={user_batch: [user_id], item_batch: [item_id], wins_batch: [0], fails_batch: [0]})
            preds.append(float(ops.sigmoid(item_logit)[0]))
            for _ in range(EPOCH_MAX):
                sess.run([train_op, infer, user_bias, user_features], feed_dict={
                    user_batch: h[0], item_batch: h[1], rate_batch: h[2], wins_batch: h[2], fails_batch: h[2]})
        q_after = sess.run(item_features)
    assert np.array_equal(q_after, t["Q"])
    with _model(U, I, D, t, **kw) as m:
        got = AT.non_adaptive_test(m, df, epoch_max=EPOCH_MAX)
    assert got["pred"] == preds
