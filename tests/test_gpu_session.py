"""The reference-shaped surface on the GPU: Session.run over ops.inference_svd /
ops.optimization, the svd() driver loop and Saver (svd_train_val.py:40-72,106-198)."""
import numpy as np
import pytest

from tfrecomm_amd import _lib as L
from tfrecomm_amd import dataio, graph as tf, ops, svd_train_val
from oracle import svd_oracle as so
from tests.util import RTOL, assert_close, make_oracle, rand_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fresh_graph():
    tf.reset_default_graph()
    yield
    tf.reset_default_graph()


def test_session_run_matches_oracle_canonical():
    U, I, D, B = 120, 90, 15, 256
    rs = np.random.RandomState(0)
    t = rand_tables(rs, U, I, D)
    user_batch = tf.placeholder("int32", shape=[None], name="id_user")
    item_batch = tf.placeholder("int32", shape=[None], name="id_item")
    rate_batch = tf.placeholder("float32", shape=[None])
    infer, regularizer = ops.inference_svd(user_batch, item_batch, user_num=U, item_num=I, dim=D)
    global_step = tf.get_or_create_global_step()
    cost, train_op = ops.optimization(infer, regularizer, rate_batch, learning_rate=1e-3, reg=0.05)
    logits = tf.get_default_graph().node("logits", "logits")
    orc = make_oracle(U, I, D, t, optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05)
    with tf.Session() as sess:
        sess.run(tf.group(tf.global_variables_initializer(), tf.local_variables_initializer()))
        sess.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])       # inject: init is not a parity target
        it = dataio.ShuffleIterator([rs.randint(0, U, 5000).astype(np.int32), rs.randint(0, I, 5000).astype(np.int32),
                                     rs.randint(1, 6, 5000).astype(np.float32)], batch_size=B)
        np.random.seed(13575)
        for s in range(3):
            users, items, rates = next(it)                                    # float64 columns
            _, lg, inf, c, rg = sess.run([train_op, logits, infer, cost, regularizer],
                                         feed_dict={user_batch: users, item_batch: items, rate_batch: rates})
            wl, wloss, wreg = orc.train_step(users, items, rates)
            assert_close(lg, wl, rtol=RTOL * (s + 1))
            assert np.array_equal(inf, lg)                                    # canonical infer = logits
            assert_close(c, wloss, rtol=RTOL * (s + 1))
            assert_close(rg, wreg, rtol=RTOL * (s + 1))
        assert sess.run(global_step) == 3
        vu, vi = rs.randint(0, U, 777), rs.randint(0, I, 777)
        lg, inf = sess.run([logits, infer], feed_dict={user_batch: vu, item_batch: vi})
        assert_close(lg, orc.forward(vu, vi), rtol=4 * RTOL)
        P = sess.run(ops.variables()["user_features"])
        assert P.shape == (U, D)
        assert_close(P, orc.P, rtol=4 * RTOL)
        with pytest.raises(IndexError):
            sess.run(logits, feed_dict={user_batch: np.array([U]), item_batch: np.array([0])})


def test_session_fork_style_nll():
    U, I, D, B = 60, 50, 20, 128
    rs = np.random.RandomState(1)
    t = rand_tables(rs, U, I, D)
    ub, ib, rb = tf.placeholder("int32"), tf.placeholder("int32"), tf.placeholder("float32")
    wb, fb = tf.placeholder("float32", name="nb_wins"), tf.placeholder("float32", name="nb_fails")
    infer, logits, regularizer, user_bias, user_features, item_bias, item_features = ops.inference_svd(
        ub, ib, wb, fb, user_num=U, item_num=I, dim=D, device="/cpu:0", fork_semantics=True)
    tf.get_or_create_global_step()
    cost_nll, train_op = ops.optimization(infer, logits, regularizer, rb, learning_rate=5e-3, reg=0.01, device="/cpu:0")
    orc = make_oracle(U, I, D, t, loss="nll", item_abs=True, reg_bias=True, optimizer="sgd", lr=5e-3, reg=0.01)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        sess.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        u, i = rs.randint(0, U, B), rs.randint(0, I, B)
        r = (rs.rand(B) < 0.5).astype(np.float32)
        _, lg, inf = sess.run([train_op, logits, infer], feed_dict={ub: u, ib: i, rb: r, wb: r, fb: r})
        wl, wloss, _ = orc.train_step(u, i, r)
        assert_close(lg, wl)
        assert np.array_equal(inf, so.head(lg.astype(np.float64), "nll"))
        nll = sess.run(cost_nll, feed_dict={rb: r, logits: lg})               # svd_train_val.py:94
        assert_close(nll, wloss, rtol=1e-5)
        assert_close(sess.run(user_features), orc.P, rtol=2 * RTOL)
        # the handles are the gathered embeddings (ops.py:13-14,37-38): fed ids select rows, adaptive_test.py:42-44 style
        allu = sess.run(user_features, feed_dict={ub: range(U)})
        assert allu.shape == (U, D) and np.array_equal(allu, sess.run(user_features))
        some_u, some_i = np.array([5, 5, 0, 59]), np.array([49, 1])
        got_pu, got_bu, got_q = sess.run([user_features, user_bias, item_features], feed_dict={ub: some_u, ib: some_i})
        assert np.array_equal(got_pu, allu[some_u]) and got_bu.shape == (4,) and got_q.shape == (2, D)
        assert_close(got_q, orc.Q[some_i], rtol=2 * RTOL)


def test_svd_driver_prints_reference_rows_and_learns(tmp_path):
    U, I = 300, 200
    train, val = svd_train_val.synthetic_frames(U, I, 40000, seed=5)
    np.random.seed(13575)
    lines = []
    rows = svd_train_val.svd(train, val, user_num=U, item_num=I, dim=15, batch_size=1000, epoch_max=6,
                             learning_rate=5e-3, reg=0.02, save_path=str(tmp_path / "fm.ckpt"), log=lines.append)
    assert lines[0] == "epoch train_error val_error elapsed_time"            # README.md:49
    assert len(rows) == 6 and [r[0] for r in rows] == list(range(6))
    assert lines[1].endswith("(s)") and len(lines[1].split()) == 4
    assert rows[0][2] > 2.0                                                   # epoch 0 = after ONE step (README.md:50)
    assert rows[-1][2] < rows[1][2] < rows[0][2]                              # validation error falls
    assert rows[-1][2] < 1.2
    ck = np.load(str(tmp_path / "fm.ckpt.npz"))
    assert ck["user_features"].shape == (U, 15) and int(ck["global_step"]) == 6 * 36
    assert "user_features/Adam" in ck.files and "bias_global" in ck.files


def test_saver_roundtrip_resumes_bit_identically(tmp_path):
    U, I, D, B = 80, 60, 8, 200
    rs = np.random.RandomState(2)
    batches = [(rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(1, 6, B).astype(np.float32)) for _ in range(4)]

    def build():
        tf.reset_default_graph()
        ub, ib, rb = tf.placeholder("int32"), tf.placeholder("int32"), tf.placeholder("float32")
        infer, reg = ops.inference_svd(ub, ib, U, I, D)
        tf.get_or_create_global_step()
        _, train_op = ops.optimization(infer, reg, rb, learning_rate=1e-2, reg=0.05)
        return ub, ib, rb, train_op
    ub, ib, rb, train_op = build()
    with tf.Session(seed=3) as sess:
        sess.run(tf.global_variables_initializer())
        for b in batches[:2]:
            sess.run(train_op, feed_dict={ub: b[0], ib: b[1], rb: b[2]})
        tf.Saver().save(sess, str(tmp_path / "ck"))
        for b in batches[2:]:
            sess.run(train_op, feed_dict={ub: b[0], ib: b[1], rb: b[2]})
        want = sess.model.tables()
    ub, ib, rb, train_op = build()
    with tf.Session(seed=99) as sess:
        tf.Saver().restore(sess, str(tmp_path / "ck"))
        assert sess.model.step == 2
        for b in batches[2:]:
            sess.run(train_op, feed_dict={ub: b[0], ib: b[1], rb: b[2]})
        got = sess.model.tables()
    for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
        assert np.array_equal(got[tid], want[tid])


def test_device_initialisers_have_the_reference_distributions():
    """ops.py:9-12,29-32: truncated normal sd 1 (biases) / sd 0.02 (features), |x| <= 2 sd."""
    import tfrecomm_amd as T
    with T.SvdModel(20000, 3000, 16) as m:
        m.init_tables(seed=1)
        t = m.tables()
    for tid, sd in ((L.P, 0.02), (L.Q, 0.02), (L.BU, 1.0), (L.BI, 1.0)):
        x = t[tid].reshape(-1).astype(np.float64)
        assert np.abs(x).max() <= 2 * sd * (1 + 1e-6)
        assert abs(x.mean()) < 0.05 * sd
        assert abs(x.std() / sd - 0.8796) < 0.02            # std of a normal truncated at 2 sigma
    assert abs(float(t[L.MU])) <= np.sqrt(3) + 1e-6
    assert not np.array_equal(t[L.P][:100], t[L.Q][:100])


def test_resident_driver_and_device_eval(tmp_path):
    """resident store + resident validation set; device-side RMSE == host RMSE."""
    import json
    import tfrecomm_amd as T
    U, I = 300, 200
    train, val = svd_train_val.synthetic_frames(U, I, 40000, seed=5)
    np.random.seed(13575)
    lines = []
    rows = svd_train_val.svd_resident(train, val, user_num=U, item_num=I, dim=16, batch_size=1000, epoch_max=5,
                                      learning_rate=5e-3, reg=0.02, json_log=str(tmp_path / "log.jsonl"), log=lines.append)
    assert lines[0] == "epoch train_error val_error elapsed_time" and len(rows) == 5
    assert rows[0][2] > 2.0 and rows[-1][2] < rows[1][2] < rows[0][2]
    recs = [json.loads(x) for x in open(str(tmp_path / "log.jsonl"))]
    assert [r["epoch"] for r in recs] == list(range(5)) and all(r["ratings_per_sec"] > 0 for r in recs)
    # device-side metric against the host computation on the same model
    rs = np.random.RandomState(0)
    t = rand_tables(rs, U, I, 16)
    with T.SvdModel(U, I, 16) as m:
        m.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        m.upload_eval_triples(val["user"], val["item"], val["outcome"])
        sse, neq, n = m.eval_resident()
        logits = m.forward(val["user"], val["item"]).astype(np.float64)
    assert n == len(val["user"])
    want = float(np.sum((logits - val["outcome"]) ** 2))
    assert abs(sse - want) <= 1e-5 * want
    with T.SvdModel(U, I, 16) as m:
        with pytest.raises(T.TfrError):
            m.eval_resident()


def test_reference_spellings_run_after_the_import_swaps(tmp_path):
    """A driver spelled the way svd_train_val.py:20-57,66-72,120-122,189-198 spells things - tf.int32 / tf.float32
    placeholders, the fork's 7-tuple inference_svd call with wins/fails, tf.train.get_or_create_global_step,
    tf.train.Saver, tf.summary.FileWriter + summary_pb2 scalar summaries, sess.graph - runs on the HIP path once
    `import tensorflow as tf`, `from tensorflow.core.framework import summary_pb2` and `import ops, dataio` point
    here.  Values are checked against the oracle."""
    import json
    from tfrecomm_amd import graph as tf                      # was: import tensorflow as tf
    from tfrecomm_amd.graph import summary_pb2                # was: from tensorflow.core.framework import summary_pb2
    from tfrecomm_amd import dataio, ops                      # was: import dataio; import ops

    def make_scalar_summary(name, val):
        return summary_pb2.Summary(value=[summary_pb2.Summary.Value(tag=name, simple_value=val)])

    USER_NUM, ITEM_NUM, DIM, BATCH_SIZE, DEVICE = 70, 50, 20, 100, "/cpu:0"
    LEARNING_RATE, LAMBDA_REG = 5e-3, 0.01
    rs = np.random.RandomState(5)
    n = 1000
    train = {"user": rs.randint(0, USER_NUM, n).astype(np.int32), "item": rs.randint(0, ITEM_NUM, n).astype(np.int32),
             "outcome": (rs.rand(n) < 0.5).astype(np.float32), "wins": rs.randint(0, 5, n).astype(np.float32),
             "fails": rs.randint(0, 5, n).astype(np.float32)}
    iter_train = dataio.ShuffleIterator([train["user"], train["item"], train["outcome"], train["wins"], train["fails"]],
                                        batch_size=BATCH_SIZE)
    iter_test = dataio.OneEpochIterator([train["user"], train["item"], train["outcome"], train["wins"], train["fails"]],
                                        batch_size=-1)

    user_batch = tf.placeholder(tf.int32, shape=[None], name="id_user")
    item_batch = tf.placeholder(tf.int32, shape=[None], name="id_item")
    rate_batch = tf.placeholder(tf.float32, shape=[None])
    wins_batch = tf.placeholder(tf.float32, shape=[None], name="nb_wins")
    fails_batch = tf.placeholder(tf.float32, shape=[None], name="nb_fails")
    infer, logits, regularizer, user_bias, user_features, item_bias, item_features = ops.inference_svd(
        user_batch, item_batch, wins_batch, fails_batch, user_num=USER_NUM, item_num=ITEM_NUM, dim=DIM, device=DEVICE,
        fork_semantics=True)
    global_step = tf.train.get_or_create_global_step()
    cost_nll, train_op = ops.optimization(infer, logits, regularizer, rate_batch, learning_rate=LEARNING_RATE,
                                          reg=LAMBDA_REG, device=DEVICE)
    init_op = tf.group(tf.global_variables_initializer(), tf.local_variables_initializer())
    saver = tf.train.Saver()
    t = rand_tables(rs, USER_NUM, ITEM_NUM, DIM)
    orc = make_oracle(USER_NUM, ITEM_NUM, DIM, t, loss="nll", item_abs=True, reg_bias=True, optimizer="sgd",
                      lr=LEARNING_RATE, reg=LAMBDA_REG)
    logdir = str(tmp_path / "log")
    with tf.Session() as sess:
        sess.run(init_op)
        sess.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])         # init is not a parity target
        summary_writer = tf.summary.FileWriter(logdir=logdir, graph=sess.graph)
        assert tf.local_variables() == []
        np.random.seed(13575)
        for i in range(4):
            train_users, train_items, train_rates, train_wins, train_fails = next(iter_train)
            _, train_logits, train_infer = sess.run(
                [train_op, logits, infer], feed_dict={user_batch: train_users, item_batch: train_items, rate_batch: train_rates,
                                                      wins_batch: train_wins, fails_batch: train_fails})
            wl, wloss, _ = orc.train_step(train_users, train_items, train_rates)
            assert_close(train_logits, wl, rtol=RTOL * (i + 1))
            nll_batch = sess.run(cost_nll, feed_dict={rate_batch: train_rates, logits: train_logits})
            assert_close(nll_batch, wloss, rtol=RTOL * (i + 1))
            proba_batch = ops.sigmoid(train_logits)
            assert np.array_equal(np.round(proba_batch), train_infer)
            summary_writer.add_summary(make_scalar_summary("training_error", float(np.mean(np.round(proba_batch) == train_rates))), i)
        for test_users, test_items, test_rates, test_wins, test_fails in iter_test:
            test_logits, test_infer = sess.run([logits, infer], feed_dict={user_batch: test_users, item_batch: test_items,
                                                                           wins_batch: test_wins, fails_batch: test_fails})
            assert_close(test_logits, orc.forward(test_users, test_items), rtol=8 * RTOL)
        assert sess.run(global_step) == 4
        path = saver.save(sess, str(tmp_path / "fm.ckpt"))
        assert path.endswith(".npz")
        summary_writer.close()
    lines = [json.loads(x) for x in open(logdir + "/events.jsonl")]
    assert [x["step"] for x in lines if x.get("tag") == "training_error"] == [0, 1, 2, 3]


def test_fork_epoch_metrics_on_device():
    """svd_train_val.py:94-98,170-178 computes, per batch on the host, round(sigmoid(logits)) == rates, the fed-logits NLL
    and sklearn's roc_auc_score.  eval_binary gives the three from the device: the AUC is an exact integer rank sum over the
    radix-sorted logits (ties share their mean rank), so against sklearn on the SAME logits it agrees to rounding."""
    import torch
    from sklearn.metrics import roc_auc_score
    import tfrecomm_amd as T
    U, I, D, B = 400, 300, 20, 30000
    rs = np.random.RandomState(3)
    t = rand_tables(rs, U, I, D)
    u, i = rs.randint(0, U, B), rs.randint(0, I, B)
    r = (rs.rand(B) < 0.4).astype(np.float32)
    orc = make_oracle(U, I, D, t, loss="nll", item_abs=True, reg_bias=True, optimizer="sgd")
    with T.SvdModel(U, I, D, loss="nll", item_abs=True, reg_bias=True, optimizer="sgd", device=0) as m:
        m.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        got = m.eval_binary(u, i, r)
        lg = m.forward(u, i)
        want_logits = orc.forward(u, i)
        p = 1.0 / (1.0 + np.exp(-want_logits))
        assert abs(got["acc"] - np.mean(np.round(p) == r)) <= 2.0 / B              # a logit within rounding of 0 may flip
        nll = np.mean(np.maximum(want_logits, 0) - want_logits * r + np.log1p(np.exp(-np.abs(want_logits))))
        assert abs(got["mean_nll"] - nll) <= 1e-5 * nll
        assert abs(got["auc"] - roc_auc_score(r, lg.astype(np.float64))) <= 1e-12   # same logits: exact rank arithmetic
        assert abs(got["auc"] - roc_auc_score(r, p)) <= 1e-6
        # the resident validation set gives the same numbers
        m.upload_eval_triples(u, i, r)
        res = m.eval_binary_resident()
        assert res["n"] == B and res["auc"] == got["auc"] and res["acc"] == got["acc"]
        # heavy ties, both signs, zeros of both signs, one class empty
        dev = torch.device("cuda", 0)
        for n, levels in ((5000, 7), (100000, 3), (64, 64), (1, 1)):
            sc = rs.randint(-levels, levels + 1, n).astype(np.float32) * np.float32(0.25)
            sc[sc == 0] *= rs.choice([-1.0, 1.0], int((sc == 0).sum())).astype(np.float32)
            lab = (rs.rand(n) < 0.3).astype(np.float32)
            d_sc, d_lab = torch.from_numpy(sc).to(dev), torch.from_numpy(lab).to(dev)     # keep both alive across the call
            torch.cuda.synchronize()
            a = m.auc_dev(d_sc.data_ptr(), d_lab.data_ptr(), n)
            if 0 < lab.sum() < n:
                assert abs(a - roc_auc_score(lab, sc.astype(np.float64))) <= 1e-12, (n, levels)
            else:
                assert np.isnan(a)
    with T.SvdModel(U, I, D, device=0) as m2:                    # the canonical (mse) model has no such metrics
        with pytest.raises(T.TfrError):
            m2.eval_binary(u, i, r)


def test_discrete_driver_logs_the_forks_epoch_line():
    """svd(discrete=True): the fork's `TRAIN(size, macc, mauc, mnll) TEST(size, macc, auc, mnll)` line (svd_train_val.py:170-178),
    AUC / NLL / accuracy from the device; the printed test AUC equals sklearn's on the model's own logits."""
    import re
    from sklearn.metrics import roc_auc_score
    rs = np.random.RandomState(9)
    U, I, n = 200, 150, 6000
    pu, qi = rs.normal(0, 1, (U, 4)), rs.normal(0, 1, (I, 4))
    u, i = rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32)
    y = (rs.rand(n) < 1 / (1 + np.exp(-(pu[u] * np.abs(qi[i])).sum(1)))).astype(np.float32)
    mk = lambda s: {"user": u[s], "item": i[s], "outcome": y[s]}
    train, val = mk(slice(0, 5000)), mk(slice(5000, n))
    lines = []
    np.random.seed(13575)
    # learning rate: the fork's SGD runs on the SUMMED loss (ops.py:140), so the effective step is lr x 500.  The float64
    # oracle on the same data (3 initialisations) diverges at 0.05 (test NLL 0.85 -> 2-4) and falls monotonically at 0.01
    # (0.85 -> 0.71 over six epochs): 0.01 is where "the driver improves the NLL" is a property of the arithmetic.
    rows = svd_train_val.svd(train, val, user_num=U, item_num=I, dim=8, batch_size=500, epoch_max=6, learning_rate=0.01,
                             reg=0.001, discrete=True, log=lines.append)
    ep = [l for l in lines if "TRAIN(" in l]
    assert len(ep) == len(rows) == 6
    m = re.search(r"TRAIN\(size=500/5000, macc=([\d.]+), mauc=([\d.]+), mnll=([\d.]+)\) TEST\(size=1000, macc=([\d.]+), auc=([\d.]+), mnll=([\d.]+)\)", ep[-1])
    assert m, ep[-1]
    tr_acc, tr_auc, tr_nll, te_acc, te_auc, te_nll = map(float, m.groups())
    assert 0.5 < tr_auc <= 1.0 and 0.5 < te_auc <= 1.0 and 0 < tr_nll < 10 and 0 < te_nll < 10
    tests = [re.search(r"TEST\(size=1000, macc=([\d.]+), auc=([\d.]+), mnll=([\d.]+)\)", l) for l in ep]
    nlls, aucs = [float(t.group(3)) for t in tests], [float(t.group(2)) for t in tests]
    assert aucs[-1] > aucs[0]                                    # it learns to rank
    assert nlls[-1] < nlls[0] - 0.02 and all(b < a + 1e-3 for a, b in zip(nlls, nlls[1:])), nlls   # and the test NLL falls, epoch by epoch
