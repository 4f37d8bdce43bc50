"""GPU parity: the HIP path (through the C-ABI) against the float64 oracle and the committed
golden vectors.  Tolerance: 1e-5 scale-relative for floating point (north_star), bit-exact
for index work.  Run with ``-m gpu`` on the MI355X box; nothing here reads /root/reference."""
import numpy as np
import pytest

import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
from oracle import svd_oracle as so
from tests.util import RTOL, assert_close, dup_heavy_ids, make_oracle, rand_tables, rel_err, TABLE_NAMES

pytestmark = pytest.mark.gpu

TIDS = (L.MU, L.BU, L.BI, L.P, L.Q)


def model_from(U, I, D, tables, **kw):
    frozen = kw.pop("frozen", 0)
    m = T.SvdModel(U, I, D, **kw)
    m.set_tables(tables["mu"], tables["bu"], tables["bi"], tables["P"], tables["Q"])
    if frozen:
        m.set_frozen(frozen)
    return m


# ------------------------------------------------------------------ forward
@pytest.mark.parametrize("U,I,D,B", [(7, 5, 3, 4), (50, 40, 15, 64), (300, 200, 64, 257), (300, 200, 128, 512),
                                     (64, 64, 5, 33), (100, 80, 20, 100), (30, 30, 4, 1), (30, 30, 8, 63),
                                     (30, 30, 32, 65), (40, 40, 256, 129), (40, 40, 1, 10), (40, 40, 63, 70)])
@pytest.mark.parametrize("item_abs", [False, True])
def test_forward_matches_oracle(U, I, D, B, item_abs):
    rs = np.random.RandomState(U * 1000 + D)
    t = rand_tables(rs, U, I, D)
    u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
    with model_from(U, I, D, t, item_abs=item_abs) as m:
        got = m.forward(u, i)
    want = make_oracle(U, I, D, t, item_abs=item_abs).forward(u, i)
    assert got.dtype == np.float32 and got.shape == (B,)
    assert_close(got, want, what="logits")


def test_forward_golden(golden):
    g = golden("svd_forward_grad.npz")
    for key in sorted({k.split("/")[0] for k in g.files}):
        U, I, D, B = (int(x[1:]) for x in key.split("_"))
        t = {k: g["%s/%s" % (key, k)] for k in ("mu", "bu", "bi", "P", "Q")}
        u, i = g[key + "/u"], g[key + "/i"]
        for ia in (0, 1):
            with model_from(U, I, D, t, item_abs=bool(ia)) as m:
                got = m.forward(u, i)
            assert_close(got, g["%s/mse_abs%d_rb0/logits" % (key, ia)], what=key)


def test_forward_accepts_float64_id_columns():
    # the reference's iterator yields float64 columns (dataio.py:103,117)
    rs = np.random.RandomState(3)
    t = rand_tables(rs, 20, 20, 16)
    u, i = rs.randint(0, 20, 50), rs.randint(0, 20, 50)
    with model_from(20, 20, 16, t) as m:
        a = m.forward(u.astype(np.float64), i.astype(np.float64))
        b = m.forward(u.astype(np.int32), i.astype(np.int64))
        assert np.array_equal(a, b)
        with pytest.raises(ValueError):
            m.forward(u + 0.5, i)


# ------------------------------------------------------------------ loss / grads via one SGD step
@pytest.mark.parametrize("loss", ["mse", "nll"])
@pytest.mark.parametrize("item_abs", [0, 1])
@pytest.mark.parametrize("reg_bias", [0, 1])
def test_loss_reg_and_gradients_golden(golden, loss, item_abs, reg_bias):
    """loss / regulariser values, and the reduced gradients recovered from one SGD step with
    lr=1 (var_new = var - grad), against the float64 golden gradients."""
    g = golden("svd_forward_grad.npz")
    for key in sorted({k.split("/")[0] for k in g.files}):
        U, I, D, B = (int(x[1:]) for x in key.split("_"))
        t = {k: g["%s/%s" % (key, k)] for k in ("mu", "bu", "bi", "P", "Q")}
        u, i, r = g[key + "/u"], g[key + "/i"], g["%s/r_%s" % (key, loss)]
        tag = "%s/%s_abs%d_rb%d" % (key, loss, item_abs, reg_bias)
        with model_from(U, I, D, t, loss=loss, item_abs=bool(item_abs), reg_bias=bool(reg_bias),
                        optimizer="sgd", lr=1.0, reg=0.05) as m:
            logits, lossv, regv = m.train_step(u, i, r)
            after = m.tables()
        assert_close(logits, g[tag + "/logits"], what=tag + " logits")
        assert_close(lossv, g[tag + "/loss"], what=tag + " loss")
        assert_close(regv, g[tag + "/reg"], what=tag + " reg")
        uq_u, uq_i = g[tag + "/uniq_u"], g[tag + "/uniq_i"]
        gP = t["P"][uq_u].astype(np.float64) - after[L.P][uq_u]
        gQ = t["Q"][uq_i].astype(np.float64) - after[L.Q][uq_i]
        gbu = t["bu"][uq_u].astype(np.float64) - after[L.BU][uq_u]
        gbi = t["bi"][uq_i].astype(np.float64) - after[L.BI][uq_i]
        gmu = float(t["mu"]) - float(after[L.MU])
        # recovered through an fp32 subtraction: tolerance relative to max(|var|,|grad|)
        for got, want, var, name in ((gP, g[tag + "/gP"], t["P"], "gP"), (gQ, g[tag + "/gQ"], t["Q"], "gQ"),
                                     (gbu, g[tag + "/gbu"], t["bu"], "gbu"), (gbi, g[tag + "/gbi"], t["bi"], "gbi")):
            scale = max(np.abs(want).max(), np.abs(var).max())
            assert np.abs(got - want).max() <= 4e-6 * scale, "%s %s" % (tag, name)
        assert abs(gmu - float(g[tag + "/gmu"])) <= 4e-6 * max(1.0, abs(float(g[tag + "/gmu"])))
        # rows not in the batch are untouched by SGD
        mask = np.ones(U, bool); mask[uq_u] = False
        assert np.array_equal(after[L.P][mask], t["P"][mask])


# ------------------------------------------------------------------ trajectories
def _check_tables(m, want_tables, tag, rtol):
    got = m.tables()
    for tid in TIDS:
        assert_close(got[tid], want_tables[tid], rtol=rtol, what="%s table %s" % (tag, TABLE_NAMES[tid]))


def test_trajectories_golden(golden):
    g = golden("svd_trajectories.npz")
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) >= 6
    from tests.golden.make_golden import TRAJ_CASES, NSTEPS
    for name, U, I, D, B, kw in TRAJ_CASES:
        kw = dict(kw)
        frozen = kw.pop("frozen", 0)
        t = {k: g["%s/init/%s" % (name, k)] for k in ("mu", "bu", "bi", "P", "Q")}
        with model_from(U, I, D, t, frozen=frozen, **kw) as m:
            for s in range(NSTEPS):
                p = "%s/step%d/" % (name, s)
                logits, lossv, regv = m.train_step(g[p + "u"], g[p + "i"], g[p + "r"])
                # error accumulates over steps: allow 1e-5 per step taken
                tol = RTOL * (s + 1)
                assert_close(logits, g[p + "logits"], rtol=tol, what=p + "logits")
                assert_close(lossv, g[p + "loss"], rtol=tol, what=p + "loss")
                assert_close(regv, g[p + "reg"], rtol=tol, what=p + "reg")
                want = {tid: g[p + TABLE_NAMES[tid]] for tid in TIDS}
                _check_tables(m, want, p, tol)
                if kw["optimizer"] == "adam":
                    for tid in TIDS:
                        # bias_global's gradient is sum_k g_k, a cancelling sum of B fp32 values:
                        # its relative error is cond = sum|g| / |sum g| times the per-term 1e-7,
                        # and v squares it.  Scale that one scalar's tolerance by cond.
                        cond = 1.0
                        if tid == L.MU:
                            gk = so.dlogits(g[p + "logits"], g[p + "r"].astype(np.float64), kw["loss"])
                            cond = max(1.0, float(np.abs(gk).sum() / max(abs(gk.sum()), 1e-30)))
                        assert_close(m.get_table(tid | L.SLOT_M), g[p + TABLE_NAMES[tid] + "_m"], rtol=tol * cond, what=p + "m")
                        assert_close(m.get_table(tid | L.SLOT_V), g[p + TABLE_NAMES[tid] + "_v"], rtol=2 * tol * cond, what=p + "v")
            assert m.step == NSTEPS


@pytest.mark.parametrize("opt,mode", [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")])
@pytest.mark.parametrize("loss,item_abs,reg_bias", [("mse", 0, 0), ("nll", 1, 1)])
@pytest.mark.parametrize("D", [15, 64, 128])
def test_trajectory_vs_oracle_seeded(opt, mode, loss, item_abs, reg_bias, D):
    """10 steps on seeded inputs against the float64 oracle run side by side."""
    U, I, B = 500, 300, 700
    rs = np.random.RandomState(D + 17)
    t = rand_tables(rs, U, I, D)
    kw = dict(loss=loss, item_abs=bool(item_abs), reg_bias=bool(reg_bias), optimizer=opt, adam_mode=mode,
              lr=2e-3, reg=0.03)
    orc = make_oracle(U, I, D, t, **kw)
    with model_from(U, I, D, t, **kw) as m:
        for s in range(10):
            u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < 0.5).astype(np.float32) if loss == "nll" else rs.randint(1, 6, B).astype(np.float32)
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(u, i, r)
            tol = RTOL * (s + 1)
            assert_close(logits, wl, rtol=tol, what="step %d logits" % s)
            assert_close(lossv, wloss, rtol=tol, what="step %d loss" % s)
            assert_close(regv, wreg, rtol=tol, what="step %d reg" % s)
        _check_tables(m, orc.tables(), "final", RTOL * 10)
        # RMSE of the final model on a held-out batch (svd_train_val.py:149)
        u, i = rs.randint(0, U, 1000), rs.randint(0, I, 1000)
        r = rs.randint(1, 6, 1000).astype(np.float32)
        got = so.rmse(r, so.head(m.forward(u, i).astype(np.float64), loss))
        want = so.rmse(r, orc.infer(u, i))
        assert abs(got - want) <= RTOL * want


# ------------------------------------------------------------------ index work: bit-exact
def test_sort_segments_bit_exact(golden):
    g = golden("svd_segments.npz")
    for key in sorted({k.split("/")[0] for k in g.files}):
        n = int(key.split("_")[0][1:])
        ids = g[key + "/ids"]
        with T.SvdModel(n, n, 4) as m:
            for side in (0, 1):
                ks, ps = m.sort_segments(side, ids)
                assert np.array_equal(ks, g[key + "/sorted_ids"]), key
                assert np.array_equal(ps, g[key + "/sorted_pos"]), key
        heads = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
        assert np.array_equal(np.concatenate((heads, [ks.size])), g[key + "/seg_start"])
        # unique in first-occurrence order == ids at the sorted positions of each head, re-ordered
        first = np.sort(ps[heads])
        assert np.array_equal(ids[first], g[key + "/unique_first_occurrence"])


@pytest.mark.parametrize("B", [262144, 3_000_000, 40_000_000])
def test_sort_large_random_matches_numpy(B):
    """262144: the C3 batch; 3 M keys: the scatter takes the scan blocks' totals from its LDS prefix (FM non-zeros are sorted at
    this size); 40 M keys: more scan blocks than that prefix holds."""
    rs = np.random.RandomState(5)
    n = 10_000_000
    ids = rs.randint(0, n, B).astype(np.int32)
    with T.SvdModel(n, 16, 4, optimizer="sgd") as m:
        ks, ps = m.sort_segments(0, ids)
    want = np.argsort(ids, kind="stable").astype(np.int32)
    assert np.array_equal(ps, want)
    assert np.array_equal(ks, ids[want])


@pytest.mark.parametrize("n", [20_000, 300_000, 10_000_000, 20_000_000])
def test_radix_sort_edges(n):
    """The radix sort at 2, 3 and 4 passes (15, 19, 24, 25 key bits): one key, partial waves and tiles, a hot id on a third of the
    batch and a block of equal keys - against np.argsort(kind="stable")."""
    rs = np.random.RandomState(n % 1000 + 3)
    with T.SvdModel(n, 16, 4, optimizer="sgd") as m:
        for B in (1, 63, 64, 65, 1023, 1024, 1025, 4097, 100_000, 262_144, 300_001):
            ids = rs.randint(0, n, B).astype(np.int32)
            if B > 1000:
                ids[rs.rand(B) < 0.33] = ids[0]
                ids[B // 2: B // 2 + 500] = n - 1
            want = np.argsort(ids, kind="stable").astype(np.int32)
            ks, ps = m.sort_segments(0, ids)
            assert np.array_equal(ps, want), (n, B)
            assert np.array_equal(ks, ids[want]), (n, B)


# ------------------------------------------------------------------ determinism
@pytest.mark.parametrize("opt,mode", [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")])
def test_run_to_run_bit_identical(opt, mode):
    U, I, D, B = 2000, 1500, 64, 5000
    rs = np.random.RandomState(11)
    t = rand_tables(rs, U, I, D)
    batches = [(dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B), rs.randint(1, 6, B).astype(np.float32)) for _ in range(4)]
    outs = []
    for rep in range(2):
        with model_from(U, I, D, t, optimizer=opt, adam_mode=mode) as m:
            res = [m.train_step(*b) for b in batches]
            outs.append((res, m.tables()))
    for (la, lossa, rega), (lb, lossb, regb) in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(la, lb) and lossa == lossb and rega == regb
    for tid in TIDS:
        assert np.array_equal(outs[0][1][tid], outs[1][1][tid])


# ------------------------------------------------------------------ edge cases
def test_out_of_range_ids_raise_and_leave_state_untouched():
    U, I, D = 50, 40, 16
    rs = np.random.RandomState(2)
    t = rand_tables(rs, U, I, D)
    with model_from(U, I, D, t, optimizer="adam", adam_mode="tf1") as m:
        u, i = rs.randint(0, U, 64).astype(np.int32), rs.randint(0, I, 64).astype(np.int32)
        r = rs.randint(1, 6, 64).astype(np.float32)
        for bad_u, bad_i in ((U, 0), (-1, 0), (0, I), (0, -5)):
            uu, ii = u.copy(), i.copy()
            uu[7], ii[9] = (bad_u if bad_u not in (0,) else uu[7]), (bad_i if bad_i not in (0,) else ii[9])
            with pytest.raises(T.OutOfRangeError):
                m.forward(uu, ii)
            with pytest.raises(IndexError):
                m.train_step(uu, ii, r)
            assert m.step == 0
            after = m.tables()
            for tid, name in TABLE_NAMES.items():
                assert np.array_equal(after[tid].reshape(-1), np.asarray(t[name]).reshape(-1)), name
        m.train_step(u, i, r)          # still usable afterwards
        assert m.step == 1


@pytest.mark.parametrize("opt,mode", [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")])
@pytest.mark.parametrize("B", [0, 1, 63, 64, 65])
def test_ragged_and_empty_batches(opt, mode, B):
    U, I, D = 30, 20, 20
    rs = np.random.RandomState(B)
    t = rand_tables(rs, U, I, D)
    kw = dict(optimizer=opt, adam_mode=mode, lr=1e-2, reg=0.05)
    orc = make_oracle(U, I, D, t, **kw)
    with model_from(U, I, D, t, **kw) as m:
        for s in range(2):
            u, i = rs.randint(0, U, B).astype(np.int32), rs.randint(0, I, B).astype(np.int32)
            r = rs.randint(1, 6, B).astype(np.float32)
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(u, i, r)
            assert logits.shape == (B,)
            if B:
                assert_close(logits, wl, rtol=2 * RTOL)
                assert_close(lossv, wloss, rtol=2 * RTOL)
                assert_close(regv, wreg, rtol=2 * RTOL)
            else:
                assert lossv == 0.0 and regv == 0.0
        _check_tables(m, orc.tables(), "B=%d" % B, 2 * RTOL)
        assert m.forward(np.zeros(0, np.int32), np.zeros(0, np.int32)).shape == (0,)


def test_all_identical_ids_long_segment():
    """one row with thousands of duplicates in the batch (worst case for the segmented reduce)"""
    U, I, D, B = 10, 10, 64, 5000
    rs = np.random.RandomState(8)
    t = rand_tables(rs, U, I, D, scale=0.05)
    kw = dict(optimizer="adam", adam_mode="lazy", lr=1e-3, reg=0.05)
    orc = make_oracle(U, I, D, t, **kw)
    u, i = np.full(B, 3, np.int32), np.full(B, 7, np.int32)
    r = rs.randint(1, 6, B).astype(np.float32)
    with model_from(U, I, D, t, **kw) as m:
        logits, lossv, regv = m.train_step(u, i, r)
        wl, wloss, wreg = orc.train_step(u, i, r)
        assert_close(logits, wl)
        assert_close(lossv, wloss)
        assert_close(regv, wreg)
        _check_tables(m, orc.tables(), "identical", RTOL)


def test_frozen_tables_var_list():
    """var_list=[user_bias, user_features] (adaptive_test.py:28): the other three never move."""
    U, I, D, B = 60, 50, 20, 200
    rs = np.random.RandomState(4)
    t = rand_tables(rs, U, I, D)
    frozen = (1 << L.MU) | (1 << L.BI) | (1 << L.Q)
    for kw in (dict(optimizer="adam", adam_mode="tf1"), dict(optimizer="adam", adam_mode="lazy"), dict(optimizer="sgd")):
        orc = make_oracle(U, I, D, t, frozen=frozen, **kw)
        with model_from(U, I, D, t, frozen=frozen, **kw) as m:
            for s in range(3):
                u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
                r = rs.randint(1, 6, B).astype(np.float32)
                m.train_step(u, i, r)
                orc.train_step(u, i, r)
            after = m.tables()
            assert np.array_equal(after[L.Q], t["Q"]) and np.array_equal(after[L.BI], t["bi"])
            assert float(after[L.MU]) == float(t["mu"])
            assert not np.array_equal(after[L.P], t["P"])
            _check_tables(m, orc.tables(), "frozen", 3 * RTOL)


# ------------------------------------------------------------------ resident store / eval
@pytest.mark.parametrize("case", range(14))
def test_lookahead_pipelines_equal_single_steps(case):
    """Multi-step calls sort the NEXT batch ahead of time (small tables: spare blocks of the current
    launch; big tables: a second stream).  Whatever the shapes, the optimiser and the way the steps are
    cut into calls, losses and tables must equal host-fed single steps bit for bit."""
    rs = np.random.RandomState(1000 + case)
    big = case % 5 == 4                                   # rows beyond the LDS-bin limit: radix sort + stream look-ahead
    U = int(rs.randint(17000, 30000)) if big else int(rs.choice([40, 700, 6040, 16384]))
    I = int(rs.randint(200, 20000)) if big else int(rs.choice([30, 500, 3952, 9000]))
    D = int(rs.choice([8, 20, 64, 128]))
    B = int(rs.choice([1, 63, 1000, 1024, 1025, 4097, 10000]))
    if case >= 10:                                        # 13..16 tiles: the widest sweep variant; 17: past the tile path;
        # case 13: the headline configuration (10 tiles: k_dense_tiles<16,4,false,10>)
        U, I, D, B = 6040, 3952, [64, 128, 32, 64][case - 10], [16384, 12289, 16385, 10000][case - 10]
    N = 50000
    K = 7
    opt, mode = [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")][case % 3]
    if big and mode == "tf1" and opt == "adam":
        mode = "lazy"
    kw = dict(optimizer=opt, adam_mode=mode, loss=["mse", "nll"][case % 2], item_abs=bool(case & 2), reg_bias=bool(case & 4))
    if case == 13:
        kw = dict(optimizer="adam", adam_mode="tf1", loss="mse", item_abs=False, reg_bias=False)
    t = rand_tables(rs, U, I, D)
    su, si = dup_heavy_ids(rs, U, N), dup_heavy_ids(rs, I, N)
    sr = (rs.rand(N) < 0.5).astype(np.float32) if kw["loss"] == "nll" else rs.randint(1, 6, N).astype(np.float32)
    ids = rs.randint(0, N, (K, B))
    with model_from(U, I, D, t, **kw) as a, model_from(U, I, D, t, **kw) as b:
        a.upload_triples(su, si, sr)
        a.stage_ids(ids)
        la = []
        for first, n in ((0, 1), (1, 2), (3, 4)):         # 1 step (no look-ahead), 2 steps, 4 steps
            la.extend(a.train_steps_staged(first, B, n, want_loss=True))
        lb = [b.train_step(su[ids[k]], si[ids[k]], sr[ids[k]])[1] for k in range(K)]
        assert np.array_equal(np.array(la, np.float32), np.array(lb, np.float32)), (U, I, D, B, kw)
        ta, tb = a.tables(), b.tables()
        for tid in TIDS:
            assert np.array_equal(ta[tid], tb[tid]), (tid, U, I, D, B, kw)
        assert a.step == b.step == K


def test_resident_store_equals_host_fed_steps():
    U, I, D, N, B, K = 300, 200, 64, 5000, 256, 6
    rs = np.random.RandomState(21)
    t = rand_tables(rs, U, I, D)
    su, si = rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32)
    sr = rs.randint(1, 6, N).astype(np.float32)
    ids = rs.randint(0, N, (K, B))
    kw = dict(optimizer="adam", adam_mode="lazy")
    with model_from(U, I, D, t, **kw) as a, model_from(U, I, D, t, **kw) as b:
        a.upload_triples(su, si, sr)
        la = a.train_steps_resident(ids, B)
        lb = [b.train_step(su[ids[k]], si[ids[k]], sr[ids[k]])[1] for k in range(K)]
        assert np.array_equal(la, np.array(lb, np.float32))
        for tid in TIDS:
            assert np.array_equal(a.get_table(tid), b.get_table(tid))
        # staged ids: same again, split in two calls
        c = model_from(U, I, D, t, **kw)
        c.upload_triples(su, si, sr)
        c.stage_ids(ids)
        c.train_steps_staged(0, B, 2)
        lc = c.train_steps_staged(2, B, K - 2, want_loss=True)
        assert np.array_equal(lc, la[2:])
        assert np.array_equal(c.get_table(L.P), a.get_table(L.P))
        assert np.array_equal(c.forward_resident(10, 500), a.forward(su[10:500], si[10:500]))
        with pytest.raises(T.OutOfRangeError):
            c.train_steps_resident(np.array([[N] * B]), B)
        c.close()


@pytest.mark.parametrize("loss", ["mse", "nll"])
def test_eval_sse_and_accuracy(loss):
    U, I, D, N = 200, 150, 15, 30001
    rs = np.random.RandomState(6)
    t = rand_tables(rs, U, I, D)
    u, i = rs.randint(0, U, N), rs.randint(0, I, N)
    r = (rs.rand(N) < 0.5).astype(np.float32) if loss == "nll" else rs.randint(1, 6, N).astype(np.float32)
    orc = make_oracle(U, I, D, t, loss=loss)
    with model_from(U, I, D, t, loss=loss) as m:
        sse, neq = m.eval(u, i, r)
    inf = orc.infer(u, i)
    want_sse = float(np.sum((inf - r) ** 2))
    if loss == "mse":
        assert abs(sse - want_sse) <= RTOL * want_sse
        assert abs(np.sqrt(sse / N) - so.rmse(r, inf)) <= RTOL * so.rmse(r, inf)
    else:
        # rounding of sigmoid(x) at exactly 0.5 can differ only where |x| < 1e-6
        assert abs(neq - int(np.sum(inf == r))) <= 2
        assert abs(sse - want_sse) <= 2


# ------------------------------------------------------------------ full-size properties (C3-shaped forward)
def test_forward_large_properties():
    """BASELINE config 3 shape scaled to fit a test (D=128, B=262144, 2M x 200k rows):
    permutation equivariance (bitwise), bias linearity, and a sampled oracle check."""
    U, I, D, B = 2_000_000, 200_000, 128, 262144
    rs = np.random.RandomState(9)
    gen = np.random.default_rng(9)
    P = gen.standard_normal((U, D), dtype=np.float32) * np.float32(0.1)
    Q = gen.standard_normal((I, D), dtype=np.float32) * np.float32(0.1)
    bu, bi = gen.standard_normal(U, dtype=np.float32), gen.standard_normal(I, dtype=np.float32)
    u, i = rs.randint(0, U, B).astype(np.int32), rs.randint(0, I, B).astype(np.int32)
    with T.SvdModel(U, I, D, optimizer="sgd") as m:
        m.set_tables(0.25, bu, bi, P, Q)
        a = m.forward(u, i)
        perm = rs.permutation(B)
        b = m.forward(u[perm], i[perm])
        assert np.array_equal(a[perm], b)                      # per-rating result independent of batch position
        m.set_table(L.MU, np.float32(1.25))
        c = m.forward(u, i)
        assert np.abs((c - a) - 1.0).max() <= 1e-5             # logits are affine in bias_global
    sel = rs.randint(0, B, 4096)
    want = (P[u[sel]].astype(np.float64) * Q[i[sel]]).sum(1) + 0.25 + bu[u[sel]] + bi[i[sel]]
    assert_close(a[sel], want, what="sampled logits")


# ------------------------------------------------------------------ randomized sweep
def _sweep_cases():
    rs = np.random.RandomState(2024)
    dims = [1, 2, 3, 4, 7, 12, 16, 24, 33, 48, 60, 64, 100, 128, 200, 252, 256]
    cases = []
    for n in range(24):
        D = dims[rs.randint(len(dims))]
        U, I = int(rs.randint(1, 3000)), int(rs.randint(1, 2000))
        B = int(rs.choice([1, 2, 31, 64, 100, 777, 1024, 1025, 4097, 20000]))
        opt, mode = [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")][rs.randint(3)]
        loss = ["mse", "nll"][rs.randint(2)]
        cases.append((n, U, I, D, B, opt, mode, loss, bool(rs.randint(2)), bool(rs.randint(2))))
    # both sort paths with small tables: B big enough that the counting sort's table would not fit
    cases.append((100, 6040, 3952, 64, 300000, "adam", "tf1", "mse", False, False))
    cases.append((101, 20000, 17000, 32, 5000, "adam", "lazy", "mse", False, False))     # > 16384 rows: radix
    cases.append((102, 3, 2, 64, 9000, "adam", "lazy", "nll", True, True))               # very long runs
    # small-table step at its widest: four pieces per block (dim 128), 16 tiles, a full 16384-row table
    cases.append((103, 6040, 3952, 128, 10000, "adam", "tf1", "mse", False, False))
    cases.append((104, 16384, 500, 128, 10000, "adam", "lazy", "nll", True, True))
    cases.append((105, 6040, 3952, 64, 16384, "adam", "tf1", "mse", False, True))
    cases.append((106, 9000, 16384, 256, 12289, "sgd", "tf1", "nll", True, False))
    # the headline configuration itself (BASELINE configs[1]): k_tile_step<16,4,2> + k_dense_tiles<16,4,false,10>
    cases.append((107, 6040, 3952, 64, 10000, "adam", "tf1", "mse", False, False))
    # BASELINE configs[0] at its exact shape (the reference's CPU-runnable case): VEC=1 rows of 15 floats, one tile
    cases.append((108, 6040, 3952, 15, 1000, "adam", "tf1", "mse", False, False))
    # big tables (radix sort + the fused kernels' three-round load form) with tiny and ragged batches: blocks that are mostly
    # lanes past the entries; one dim that is not full-width (general form on big tables)
    cases.append((109, 40000, 30000, 128, 1, "adam", "lazy", "mse", False, False))
    cases.append((110, 40000, 30000, 64, 31, "sgd", "tf1", "nll", True, True))
    cases.append((111, 70000, 20000, 16, 1057, "adam", "lazy", "mse", False, True))
    cases.append((112, 20000, 70000, 256, 33, "adam", "lazy", "nll", True, False))
    cases.append((113, 30000, 30000, 24, 2049, "adam", "lazy", "mse", False, False))
    return cases


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: "n%d_U%d_I%d_D%d_B%d_%s_%s_%s" % c[:8])
def test_random_shapes_two_steps(case):
    n, U, I, D, B, opt, mode, loss, item_abs, reg_bias = case
    rs = np.random.RandomState(1000 + n)
    t = rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
    kw = dict(loss=loss, item_abs=item_abs, reg_bias=reg_bias, optimizer=opt, adam_mode=mode, lr=3e-3, reg=0.02)
    orc = make_oracle(U, I, D, t, **kw)
    with model_from(U, I, D, t, **kw) as m:
        for s in range(2):
            u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < 0.5).astype(np.float32) if loss == "nll" else rs.randint(1, 6, B).astype(np.float32)
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(u, i, r)
            tol = 2 * RTOL * (s + 1)
            assert_close(logits, wl, rtol=tol, what="logits")
            assert_close(lossv, wloss, rtol=tol, what="loss")
            assert_close(regv, wreg, rtol=tol, what="reg")
        got = m.tables()
        for tid in TIDS:
            # dup_heavy_ids puts 70 % of the batch on 10 % of the rows: a hot row's gradient is an fp32
            # sum of ~7*B/rows terms; rounding error grows ~sqrt(terms) (loss/logits above stay 1e-5)
            run = 7.0 * B / max(1, min(U, I))
            # Adam's first steps amplify fp32 rounding of near-cancelling gradient elements (see the
            # full-size test); SGD through the same kernels stays at 4e-5
            base = 2e-4 if opt == "adam" else 4 * RTOL
            assert_close(got[tid], orc.tables()[tid], rtol=base * max(1.0, np.sqrt(run / 64)),
                         what="table %s" % TABLE_NAMES[tid])


@pytest.mark.parametrize("opt,mode,D", [("adam", "lazy", 128), ("sgd", "tf1", 128), ("adam", "lazy", 20)])
def test_hot_rows_cut_into_hundreds_of_pieces(opt, mode, D):
    """Big-table (radix sort, fused in-place) step with one item on half of the batch and one user on a third of it: the
    hot runs are cut into several hundred block-sized pieces, finished by k_apply_rows' wide piece walk (popularity-skewed
    data, SURVEY 8d's Zipf run)."""
    U, I, B = 40000, 30000, 20000
    rs = np.random.RandomState(77 + D)
    t = rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
    kw = dict(loss="mse", item_abs=False, reg_bias=False, optimizer=opt, adam_mode=mode, lr=3e-3, reg=0.02)
    orc = make_oracle(U, I, D, t, **kw)
    with model_from(U, I, D, t, **kw) as m:
        for s in range(2):
            u = rs.randint(0, U, B).astype(np.int32)
            i = rs.randint(0, I, B).astype(np.int32)
            i[rs.rand(B) < 0.5] = 4242 + s
            u[rs.rand(B) < 0.33] = 31000 - s
            r = rs.randint(1, 6, B).astype(np.float32)
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(u, i, r)
            tol = 2 * RTOL * (s + 1)
            assert_close(logits, wl, rtol=tol, what="logits")
            assert_close(lossv, wloss, rtol=tol, what="loss")
            assert_close(regv, wreg, rtol=tol, what="reg")
        got = m.tables()
        for tid in TIDS:
            base = 2e-4 if opt == "adam" else 4 * RTOL           # as in the sweep above: fp32 sums of ~B/2 terms on the hot rows
            assert_close(got[tid], orc.tables()[tid], rtol=base * np.sqrt(B / 2 / 64), what="table %s" % TABLE_NAMES[tid])
    # the shared walk adds the lane groups' partial sums in group order: a second run gives the same bits
    rs = np.random.RandomState(77 + D)
    rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
    with model_from(U, I, D, t, **kw) as m2:
        for s in range(2):
            u = rs.randint(0, U, B).astype(np.int32)
            i = rs.randint(0, I, B).astype(np.int32)
            i[rs.rand(B) < 0.5] = 4242 + s
            u[rs.rand(B) < 0.33] = 31000 - s
            m2.train_step(u, i, rs.randint(1, 6, B).astype(np.float32))
        again = m2.tables()
    for tid in TIDS:
        assert np.array_equal(got[tid], again[tid]), TABLE_NAMES[tid]


# ------------------------------------------------------------------ full-size training step (C3-shaped)
@pytest.mark.parametrize("opt,mode", [("adam", "lazy"), ("sgd", "tf1")])
def test_train_step_large_against_compacted_oracle(opt, mode):
    """BASELINE config 3 shape scaled to fit a test (D=128, B=262144, 2M x 200k rows, radix sort path,
    fused in-place updates): the oracle runs on the compacted problem of the rows the batch touches,
    which is exact for lazy Adam / SGD (untouched rows do not move - also checked)."""
    U, I, D, B = 2_000_000, 200_000, 128, 262144
    gen = np.random.default_rng(17)
    P = gen.standard_normal((U, D), dtype=np.float32) * np.float32(0.05)
    Q = gen.standard_normal((I, D), dtype=np.float32) * np.float32(0.05)
    bu, bi = gen.standard_normal(U, dtype=np.float32) * np.float32(0.3), gen.standard_normal(I, dtype=np.float32) * np.float32(0.3)
    rs = np.random.RandomState(17)
    u = rs.randint(0, U, B).astype(np.int32)
    i = np.where(rs.rand(B) < 0.3, rs.randint(0, 50, B), rs.randint(0, I, B)).astype(np.int32)   # 50 very hot items
    r = rs.randint(1, 6, B).astype(np.float32)
    kw = dict(optimizer=opt, adam_mode=mode, lr=2e-3, reg=0.03)
    uu, ui = np.unique(u), np.unique(i)
    orc = so.SvdOracle(uu.size, ui.size, D, dtype=np.float64, **kw)
    orc.set_tables(0.2, bu[uu].astype(np.float64), bi[ui].astype(np.float64), P[uu].astype(np.float64), Q[ui].astype(np.float64))
    cu, ci = np.searchsorted(uu, u), np.searchsorted(ui, i)
    with T.SvdModel(U, I, D, **kw) as m:
        m.set_tables(0.2, bu, bi, P, Q)
        for s in range(2):
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(cu, ci, r)
            assert_close(logits, wl, rtol=2 * RTOL * (s + 1), what="logits")
            assert_close(lossv, wloss, rtol=2 * RTOL * (s + 1), what="loss")
            assert_close(regv, wreg, rtol=2 * RTOL * (s + 1), what="reg")
        gP, gQ, gbu, gbi = m.get_table(L.P), m.get_table(L.Q), m.get_table(L.BU), m.get_table(L.BI)
        gmu = float(m.get_table(L.MU))
    run = 0.3 * B / 50                                               # a hot item's run length
    # Adam's first steps move an element by ~lr * g / (0.03 |g| + eps): among 31 M elements a few
    # have |g| ~ 1e-6, where fp32 rounding of g is amplified - SGD (same kernels) has no such term
    base = (2e-4 if opt == "adam" else 4 * RTOL)
    assert_close(gP[uu], orc.P, rtol=base, what="P")
    assert_close(gQ[ui], orc.Q, rtol=base * np.sqrt(run / 64), what="Q")
    assert_close(gbu[uu], orc.bu, rtol=base, what="bu")
    assert_close(gbi[ui], orc.bi, rtol=base * np.sqrt(run / 64), what="bi")
    assert abs(gmu - float(orc.mu)) <= 1e-4 * max(1.0, abs(float(orc.mu)))
    mask = np.ones(U, bool); mask[uu] = False
    assert np.array_equal(gP[mask], P[mask]) and np.array_equal(gbu[mask], bu[mask])      # untouched rows never move
    mask = np.ones(I, bool); mask[ui] = False
    assert np.array_equal(gQ[mask], Q[mask])


# ------------------------------------------------------------------ two-table form of the fused big-table step
@pytest.mark.parametrize("opt,mode,item_abs", [("adam", "lazy", False), ("sgd", "tf1", True)])
def test_two_table_item_rows_settle_before_every_other_reader(opt, mode, item_abs):
    """The fused big-table step keeps no copy of the pre-update item rows: an updated row goes to the alternate item table and
    the row's word flips (svd_kernels.h RedArgs::sel).  Everything else that looks at item_features - forward, eval, get /
    set_table, the row-sharded and data-parallel entry points - must first see every row back in the main table.  Steps and readers are
    interleaved here and each result is held against the float64 oracle; the same item rows are touched again and again
    (they flip back and forth), some by runs cut at a block boundary (finished in place by k_apply_rows)."""
    U, I, D, B = 40000, 30000, 64, 20000
    rs = np.random.RandomState(31)
    t = rand_tables(rs, U, I, D, scale=0.15)
    # SGD runs on the summed loss: a row with a run of ~2000 entries needs a small rate to stay finite
    kw = dict(loss="mse", item_abs=item_abs, reg_bias=False, optimizer=opt, adam_mode=mode, lr=3e-3 if opt == "adam" else 2e-5, reg=0.02)
    orc = make_oracle(U, I, D, t, **kw)
    hot = rs.randint(0, I, 400)
    with model_from(U, I, D, t, **kw) as m:
        for s in range(5):
            u = rs.randint(0, U, B).astype(np.int32)
            i = np.where(rs.rand(B) < 0.6, hot[rs.randint(0, 400, B)], rs.randint(0, I, B)).astype(np.int32)
            i[rs.rand(B) < 0.1] = hot[0]                             # one run of ~2000 entries: dozens of pieces
            r = rs.randint(1, 6, B).astype(np.float32)
            logits, lossv, regv = m.train_step(u, i, r)
            wl, wloss, wreg = orc.train_step(u, i, r)
            tol = 2 * RTOL * (s + 1)
            assert_close(logits, wl, rtol=tol, what="logits step %d" % s)
            assert_close(lossv, wloss, rtol=tol, what="loss")
            if s == 1:                                               # a forward between two fused steps
                fu, fi = rs.randint(0, U, 5000).astype(np.int32), hot[rs.randint(0, 400, 5000)].astype(np.int32)
                assert_close(m.forward(fu, fi), orc.forward(fu, fi), rtol=tol, what="forward after step 1")
            if s == 2:                                               # get_table, then put a changed table back
                q = m.get_table(L.Q)
                assert_close(q, orc.tables()[L.Q], rtol=(2e-4 if opt == "adam" else 4 * RTOL) * 6, what="Q after step 2")
                q[hot[:7]] *= np.float32(0.5)
                m.set_table(L.Q, q)
                oq = orc.tables()[L.Q]; oq[hot[:7]] = q[hot[:7]].astype(np.float64)
                orc.set_tables(orc.mu, orc.bu, orc.bi, orc.P, oq)
            if s == 3:                                               # a tiny batch (a single partly filled block) in between
                su, si = rs.randint(0, U, 300).astype(np.int32), hot[rs.randint(0, 400, 300)].astype(np.int32)
                sr = rs.randint(1, 6, 300).astype(np.float32)
                l2, _, _ = m.train_step(su, si, sr)
                w2, _, _ = orc.train_step(su, si, sr)
                assert_close(l2, w2, rtol=tol, what="small-batch step")
        got = m.tables()
    for tid in TIDS:
        assert_close(got[tid], orc.tables()[tid], rtol=(2e-4 if opt == "adam" else 4 * RTOL) * 8, what="table %s" % TABLE_NAMES[tid])


_TWO_TABLE_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
U, I, D, B = 300000, 50000, 128, 100000
rs = np.random.RandomState(3)
with T.SvdModel(U, I, D, optimizer="adam", adam_mode="lazy", lr=2e-3, reg=0.02) as m:
    m.init_tables(seed=5)
    N = 400000
    u = rs.randint(0, U, N).astype(np.int32)
    i = np.where(rs.rand(N) < 0.3, rs.randint(0, 20, N), rs.randint(0, I, N)).astype(np.int32)
    m.upload_triples(u, i, rs.randint(1, 6, N).astype(np.float32))
    np.random.seed(1)
    m.rng_from_numpy()
    loss = m.train_steps_drawn(B, 7, want_loss=True)          # look-ahead pipeline, fused steps
    lg, l1, _ = m.train_step(u[:B], i[:B], np.ones(B, np.float32))
    h = hashlib.sha256()
    for tid in (L.MU, L.BU, L.BI, L.P, L.Q, L.Q | L.SLOT_M, L.Q | L.SLOT_V):
        h.update(np.ascontiguousarray(m.get_table(tid)).tobytes())
    h.update(loss.tobytes()); h.update(lg.tobytes())
    print("HASH", h.hexdigest())
"""


def test_two_table_form_is_bit_identical_to_the_copy_form():
    """TFR_DUALQ=0 restores the per-entry copy of the pre-update item rows.  Both forms add the same numbers in the same order:
    tables, Adam slots, losses and logits after eight steps (seven through the look-ahead pipeline) hash identically."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for flag in ("1", "0"):
        env = dict(os.environ, TFR_DUALQ=flag)
        p = subprocess.run([sys.executable, "-c", _TWO_TABLE_SCRIPT % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out.append([l for l in p.stdout.decode().splitlines() if l.startswith("HASH")][0])
    assert out[0] == out[1]


_LOAD_ROUNDS_SCRIPT = r"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, %r)
import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
c = json.loads(os.environ["TFR_TEST_CASE"])
U, I, D, B = c["U"], c["I"], c["D"], c["B"]
rs = np.random.RandomState(11)
with T.SvdModel(U, I, D, **c["kw"]) as m:
    m.init_tables(seed=5)
    N = 4 * B
    u = rs.randint(0, U, N).astype(np.int32)
    i = np.where(rs.rand(N) < c["hot"], rs.randint(0, 20, N), rs.randint(0, I, N)).astype(np.int32)    # hot rows: runs cut into many pieces
    r = (rs.randint(0, 2, N) if c["kw"].get("loss") == "nll" else rs.randint(1, 6, N)).astype(np.float32)
    m.upload_triples(u, i, r)
    np.random.seed(1)
    m.rng_from_numpy()
    loss = m.train_steps_drawn(B, 5, want_loss=True)          # look-ahead pipeline, fused steps
    lg, l1, _ = m.train_step(u[:B - 37], i[:B - 37], r[:B - 37])          # a ragged last block
    h = hashlib.sha256()
    for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
        h.update(np.ascontiguousarray(m.get_table(tid)).tobytes())
    if c["kw"]["optimizer"] == "adam":
        for tid in (L.P | L.SLOT_M, L.P | L.SLOT_V, L.Q | L.SLOT_M, L.Q | L.SLOT_V, L.BU | L.SLOT_M, L.BI | L.SLOT_V):
            h.update(np.ascontiguousarray(m.get_table(tid)).tobytes())
    h.update(loss.tobytes()); h.update(lg.tobytes()); h.update(np.float32(l1).tobytes())
    print("PLAN", m.kernel_plan(B))
    print("HASH", h.hexdigest())
"""


@pytest.mark.parametrize("case", [
    dict(U=300000, I=50000, D=128, B=60000, hot=0.3, kw=dict(optimizer="adam", adam_mode="lazy", lr=2e-3, reg=0.02)),
    dict(U=200000, I=80000, D=64, B=50000, hot=0.0, kw=dict(optimizer="sgd", lr=2e-5, reg=0.02, loss="nll", reg_bias=True)),
    dict(U=150000, I=90000, D=16, B=70000, hot=0.5, kw=dict(optimizer="adam", adam_mode="lazy", lr=1e-3, reg=0.05, item_abs=True)),
], ids=["adam-d128-hot", "sgd-nll-d64", "adam-abs-d16-hot"])
def test_three_round_load_form_is_bit_identical_to_the_general_form(case):
    """TFR_FAST=0 makes every fused launch take the general form of k_seg_reduce (loads in program order, guarded, eight
    dependent rounds); the default takes the three-round form where it applies.  Same numbers added in the same order: tables,
    Adam slots, losses and logits hash identically (big tables, two-table step, hot rows cut into pieces, a ragged last block)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, plans = [], []
    for flag in ("1", "0"):
        env = dict(os.environ, TFR_FAST=flag, TFR_TEST_CASE=json.dumps(case))
        p = subprocess.run([sys.executable, "-c", _LOAD_ROUNDS_SCRIPT % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = p.stdout.decode().splitlines()
        out.append([l for l in lines if l.startswith("HASH")][0])
        plans.append([l for l in lines if l.startswith("PLAN")][0])
    assert "true, true, true>" in plans[0] and "true, true, false>" in plans[1], plans       # the two runs did take the two forms
    assert out[0] == out[1]


# ------------------------------------------------------------------ BASELINE config 3 at its true size
def test_config3_full_size_properties():
    """10M users x 1M items, dim=128, batch=262144 (5.6 GB of tables + 11 GB of Adam state, initialised
    on the device): properties that do not need an oracle at this size."""
    U, I, D, B = 10_000_000, 1_000_000, 128, 262144
    rs = np.random.RandomState(33)
    batches = [(rs.randint(0, U, B).astype(np.int32), rs.randint(0, I, B).astype(np.int32),
                rs.randint(1, 6, B).astype(np.float32)) for _ in range(3)]
    probe_u, probe_i = rs.randint(0, U, 100000).astype(np.int32), rs.randint(0, I, 100000).astype(np.int32)
    # make the probe overlap rows the batches touch, so it sees the updates
    probe_u[:50000], probe_i[:50000] = batches[0][0][:50000], batches[1][1][:50000]
    runs = []
    for rep in range(2):
        with T.SvdModel(U, I, D, optimizer="adam", adam_mode="lazy", lr=5e-3, reg=0.02) as m:
            m.init_tables(seed=5)
            if rep == 0:
                a = m.forward(batches[0][0], batches[0][1])
                perm = rs.permutation(B)
                assert np.array_equal(a[perm], m.forward(batches[0][0][perm], batches[0][1][perm]))
                assert np.isfinite(a).all() and a.std() > 0.5           # biases ~ N(0,1) truncated
            losses = [m.train_step(*b, want_logits=False)[1] for b in batches]
            rep_losses = [m.train_step(*batches[0], want_logits=False)[1] for _ in range(4)]
            runs.append((losses, rep_losses, m.forward(probe_u, probe_i), m.step))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]          # bit-identical trajectory
    assert np.array_equal(runs[0][2], runs[1][2])
    assert runs[0][3] == 7
    assert runs[0][1][-1] < runs[0][1][0] < runs[0][0][0]                 # repeated batch: loss falls


# ------------------------------------------------------------------ BASELINE config 4's tables on ONE GPU
def test_config4_tables_on_one_gpu_properties():
    """100M users x 10M items, dim=128 with lazy Adam: 56 GB of tables + 113 GB of Adam state = 169 GB in HBM (a table +
    state outgrows one 288 GB MI355X only past ~187M rows), initialised on the device.  Size-independent properties at
    batch 262144: permutation equivariance of the forward, a bit-identical trajectory from the same seed, a falling loss on a
    repeated batch, the device-drawn id stream == the host-drawn one, id validation at the far end of the tables."""
    U, I, D, B = 100_000_000, 10_000_000, 128, 262144
    rs = np.random.RandomState(44)
    batches = [(rs.randint(0, U, B).astype(np.int32), rs.randint(0, I, B).astype(np.int32),
                rs.randint(1, 6, B).astype(np.float32)) for _ in range(2)]
    # the far end of both tables is reachable: the last rows take part
    batches[0][0][:8], batches[0][1][:8] = U - 1 - np.arange(8), I - 1 - np.arange(8)
    probe_u, probe_i = batches[0][0][:60000].copy(), batches[1][1][:60000].copy()
    N = 3_000_000
    su, si = rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32)
    sr = rs.randint(1, 6, N).astype(np.float32)
    with T.SvdModel(U, I, D, optimizer="adam", adam_mode="lazy", lr=5e-3, reg=0.02) as m:
        runs = []
        for rep in range(2):                             # the same handle re-initialised: 169 GB are allocated once
            m.init_tables(seed=6)
            if rep == 0:
                a = m.forward(batches[0][0], batches[0][1])
                perm = rs.permutation(B)
                assert np.array_equal(a[perm], m.forward(batches[0][0][perm], batches[0][1][perm]))
                assert np.isfinite(a).all() and a.std() > 0.5
                with pytest.raises(IndexError):
                    m.forward(np.array([U], np.int64), np.array([0], np.int64))
            losses = [m.train_step(*b, want_logits=False)[1] for b in batches]
            rep_losses = [m.train_step(*batches[0], want_logits=False)[1] for _ in range(3)]
            # resident store: two steps on device-drawn ids (rep 0) / on the same ids drawn by NumPy (rep 1)
            m.upload_triples(su, si, sr)
            np.random.seed(13575)
            if rep == 0:
                m.rng_from_numpy()
                drawn = m.train_steps_drawn(B, 2, want_loss=True)
            else:
                drawn = m.train_steps_resident(np.random.randint(0, N, (2, B)), B)
            runs.append((losses, rep_losses, drawn.tolist(), m.forward(probe_u, probe_i), m.step))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert np.array_equal(runs[0][3], runs[1][3])
    assert runs[0][4] == runs[1][4] == 7
    assert runs[0][1][-1] < runs[0][1][0] < runs[0][0][0]                 # repeated batch: loss falls
