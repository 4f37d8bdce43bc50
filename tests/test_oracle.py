"""The oracle pinned against everything available offline: the committed golden
vectors, torch-CPU float64 autograd (independent derivation of the gradients), the explicit
FM pairwise sum, and the C restatement against the NumPy one.  CPU only."""
import numpy as np
import pytest

from oracle import svd_oracle as so
from tests.util import dup_heavy_ids, make_oracle, rand_tables, rel_err


def test_goldens_reproduce(golden):
    """make_golden.py is deterministic: regenerating part 1 in memory gives the committed values."""
    g = golden("svd_forward_grad.npz")
    key = "U50_I40_D15_B64"
    t = {k: g["%s/%s" % (key, k)].astype(np.float64) for k in ("mu", "bu", "bi", "P", "Q")}
    u, i = g[key + "/u"], g[key + "/i"]
    for loss in ("mse", "nll"):
        r = g["%s/r_%s" % (key, loss)].astype(np.float64)
        for ia in (0, 1):
            for rb in (0, 1):
                tag = "%s/%s_abs%d_rb%d" % (key, loss, ia, rb)
                lg = so.forward(t["P"], t["Q"], t["bu"], t["bi"], t["mu"], u, i, bool(ia))
                assert np.array_equal(lg, g[tag + "/logits"])
                assert so.data_loss(lg, r, loss) == g[tag + "/loss"]
                assert so.regularizer(t["P"], t["Q"], t["bu"], t["bi"], u, i, bool(rb)) == g[tag + "/reg"]
                gk = so.dlogits(lg, r, loss)
                dP, dQ, dbu, dbi, dmu = so.occurrence_grads(t["P"], t["Q"], t["bu"], t["bi"], u, i, gk, 0.05, bool(ia), bool(rb))
                uq, inv = so.dedup(u)
                assert np.array_equal(uq, g[tag + "/uniq_u"])
                assert np.array_equal(so.segment_sum(dP, inv, uq.size), g[tag + "/gP"])


@pytest.mark.parametrize("loss", ["mse", "nll"])
@pytest.mark.parametrize("item_abs", [False, True])
@pytest.mark.parametrize("reg_bias", [False, True])
def test_gradients_match_torch_autograd(loss, item_abs, reg_bias):
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(1)
    U, I, D, B, lam = 7, 5, 3, 16, 0.07
    P, Q, bu, bi, mu = rs.normal(size=(U, D)), rs.normal(size=(I, D)), rs.normal(size=U), rs.normal(size=I), np.array(0.3)
    u, i = rs.randint(0, U, B), rs.randint(0, I, B)
    r = (rs.rand(B) > 0.5).astype(float) if loss == "nll" else rs.uniform(1, 5, B)
    logits = so.forward(P, Q, bu, bi, mu, u, i, item_abs)
    g = so.dlogits(logits, r, loss)
    dP, dQ, dbu, dbi, dmu = so.occurrence_grads(P, Q, bu, bi, u, i, g, lam, item_abs, reg_bias)

    def dense(occ, ids, n):
        uq, inv = so.dedup(ids)
        out = np.zeros((n,) + occ.shape[1:])
        out[uq] = so.segment_sum(occ, inv, uq.size)
        return out
    tP, tQ, tbu, tbi, tmu = [torch.tensor(x, requires_grad=True) for x in (P, Q, bu, bi, mu)]
    tu, ti, tr = torch.tensor(u), torch.tensor(i), torch.tensor(r)
    pu, qi = tP[tu], tQ[ti]
    x = (pu * (qi.abs() if item_abs else qi)).sum(1) + tmu + tbu[tu] + tbi[ti]
    reg = 0.5 * (pu ** 2).sum() + 0.5 * (qi ** 2).sum()
    if reg_bias:
        reg = reg + 0.5 * (tbu[tu] ** 2).sum() + 0.5 * (tbi[ti] ** 2).sum()
    data = 0.5 * ((x - tr) ** 2).sum() if loss == "mse" else \
        torch.nn.functional.binary_cross_entropy_with_logits(x, tr, reduction="sum")
    (data + lam * reg).backward()
    assert np.abs(dense(dP, u, U) - tP.grad.numpy()).max() < 1e-12
    assert np.abs(dense(dQ, i, I) - tQ.grad.numpy()).max() < 1e-12
    assert np.abs(dense(dbu, u, U) - tbu.grad.numpy()).max() < 1e-12
    assert np.abs(dense(dbi, i, I) - tbi.grad.numpy()).max() < 1e-12
    assert abs(dmu - tmu.grad.item()) < 1e-12
    assert abs(so.data_loss(logits, r, loss) - data.item()) < 1e-12
    assert abs(so.regularizer(P, Q, bu, bi, u, i, reg_bias) - reg.item()) < 1e-12


def test_tf1_adam_matches_torch_adam_on_dense_gradients():
    """TF1's sparse Adam is dense over the table: with eps_hat placement aside it is
    torch.optim.Adam on the densified gradient - checked over 5 steps with duplicate ids."""
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(3)
    U, I, D, B = 9, 6, 4, 20
    t = rand_tables(rs, U, I, D)
    orc = make_oracle(U, I, D, t, optimizer="adam", adam_mode="tf1", lr=1e-2, reg=0.03)
    params = [torch.tensor(np.asarray(t[k], np.float64), requires_grad=True) for k in ("mu", "bu", "bi", "P", "Q")]
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-2
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    for step in range(1, 6):
        u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
        r = rs.randint(1, 6, B).astype(np.float64)
        orc.train_step(u, i, r)
        mu_, bu_, bi_, P_, Q_ = params
        tu, ti = torch.tensor(u.astype(np.int64)), torch.tensor(i.astype(np.int64))
        x = (P_[tu] * Q_[ti]).sum(1) + mu_ + bu_[tu] + bi_[ti]
        cost = 0.5 * ((x - torch.tensor(r)) ** 2).sum() + 0.03 * (0.5 * (P_[tu] ** 2).sum() + 0.5 * (Q_[ti] ** 2).sum())
        grads = torch.autograd.grad(cost, params)
        a = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)       # TF's lr_t with epsilon-hat
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                mm.mul_(b1).add_(g, alpha=1 - b1)
                vv.mul_(b2).add_(g * g, alpha=1 - b2)
                p.sub_(a * mm / (vv.sqrt() + eps))
    for p, k in zip(params, (so.MU, so.BU, so.BI, so.PF, so.QF)):
        assert rel_err(orc.tables()[k], p.detach().numpy()) < 1e-10


def test_lazy_adam_only_touches_batch_rows_and_sgd_accumulates_duplicates():
    rs = np.random.RandomState(4)
    U, I, D, B = 30, 20, 5, 12
    t = rand_tables(rs, U, I, D)
    u, i = rs.randint(0, 10, B), rs.randint(0, 8, B)
    r = rs.randint(1, 6, B).astype(np.float64)
    lazy = make_oracle(U, I, D, t, optimizer="adam", adam_mode="lazy")
    lazy.train_step(u, i, r)
    untouched = np.setdiff1d(np.arange(U), u)
    assert np.array_equal(lazy.P[untouched], np.asarray(t["P"], np.float64)[untouched])
    assert not np.array_equal(lazy.P[np.unique(u)], np.asarray(t["P"], np.float64)[np.unique(u)])
    # second step: tf1 moves rows that are NOT in the batch (momentum), lazy does not
    tf1 = make_oracle(U, I, D, t, optimizer="adam", adam_mode="tf1")
    tf1.train_step(u, i, r)
    before = tf1.P.copy()
    u2, i2 = rs.randint(10, 20, B), rs.randint(8, 16, B)
    tf1.train_step(u2, i2, r)
    assert not np.array_equal(tf1.P[np.unique(u)], before[np.unique(u)])
    # SGD with all-identical ids: the row moves by lr * sum of the B occurrence gradients
    sgd = make_oracle(U, I, D, t, optimizer="sgd", lr=1e-3, reg=0.0)
    uu, ii = np.full(B, 3), np.full(B, 2)
    lg = sgd.forward(uu, ii)
    P0, Q0 = sgd.P.copy(), sgd.Q.copy()
    sgd.train_step(uu, ii, r)
    assert np.allclose(P0[3] - sgd.P[3], 1e-3 * np.sum(lg - r) * Q0[2], rtol=1e-12)


def test_out_of_range_ids_raise():
    orc = so.SvdOracle(5, 4, 3)
    with pytest.raises(IndexError):
        orc.forward([5], [0])
    with pytest.raises(IndexError):
        orc.train_step([0], [-1], [1.0])


def test_head_rounds_half_to_even_and_rmse():
    assert np.array_equal(so.head(np.array([0.0, 1e-9, -1e-9, 3.0, -3.0]), "nll"), [0.0, 1.0, 0.0, 1.0, 0.0])
    assert so.rmse(np.array([1.0, 2.0]), np.array([2.0, 4.0])) == pytest.approx(np.sqrt(2.5))


def test_c_restatement_matches_numpy_oracle():
    COracle = pytest.importorskip("oracle.c_oracle").COracle
    try:
        from oracle import c_oracle
        c_oracle.load()
    except ImportError:
        pytest.skip("oracle/libsvd_oracle.so not built")
    rs = np.random.RandomState(0)
    U, I, D, B = 200, 150, 20, 300
    cases = (dict(optimizer="adam", adam_mode="tf1"), dict(optimizer="adam", adam_mode="lazy"),
             dict(optimizer="sgd", loss="nll", item_abs=True, reg_bias=True),
             dict(optimizer="adam", adam_mode="tf1", loss="nll", reg_bias=True))
    for kw in cases:
        t = rand_tables(rs, U, I, D)
        o = make_oracle(U, I, D, t, **kw)
        c = COracle(U, I, D, **kw)
        c.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        for s in range(5):
            u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < .5).astype(np.float32) if kw.get("loss") == "nll" else rs.randint(1, 6, B).astype(np.float32)
            a, b = o.train_step(u, i, r), c.train_step(u, i, r)
            assert rel_err(b[0], a[0]) < 5e-6 and rel_err(b[1], a[1]) < 5e-6 and rel_err(b[2], a[2]) < 1e-5
        for k in range(5):
            assert rel_err(c.table(k), o.tables()[k]) < 1e-5
        assert c.step == 5
        with pytest.raises(IndexError):
            c.train_step(np.array([U]), np.array([0]), np.array([1.0]))
        c.close()


def test_fm_forward_matches_pairwise_sum_and_golden(golden):
    g = golden("fm_forward.npz")
    for key in sorted({k.split("/")[0] for k in g.files}):
        V, W, mu = g[key + "/V"].astype(np.float64), g[key + "/W"].astype(np.float64), float(g[key + "/mu"])
        indptr, indices, data = g[key + "/indptr"], g[key + "/indices"], g[key + "/data"].astype(np.float64)
        y = so.fm_forward(mu, W, V, indptr, indices, data)
        assert np.allclose(y, g[key + "/y"], rtol=1e-13, atol=1e-13)
        for row in range(0, len(indptr) - 1, 7):                       # explicit sum_{a<b} x_a x_b <V_a, V_b>
            f, x = indices[indptr[row]:indptr[row + 1]], data[indptr[row]:indptr[row + 1]]
            pair = sum(x[a] * x[b] * V[f[a]].dot(V[f[b]]) for a in range(len(f)) for b in range(a + 1, len(f)))
            assert abs(y[row] - (mu + x.dot(W[f]) + pair)) < 1e-11
        yref = so.fm_forward_reference_form(mu, W, V, indptr, indices, data)    # forward.py:22 literally
        if key.endswith("binary"):
            assert np.allclose(y, yref, rtol=1e-12, atol=1e-12)           # coincide on 0/1 features
        else:
            assert not np.allclose(y, yref, rtol=1e-3)                    # and differ on counts (SURVEY 8c ii)


# the shapes of tests/test_gpu_parity.py::_sweep_cases where the GPU's Adam tables are held to 2e-4 instead of 1e-5
_DRIFT_CASES = [
    (107, 6040, 3952, 64, 10000, "adam", "tf1", "mse", False, False),      # the headline configuration
    (104, 16384, 500, 128, 10000, "adam", "lazy", "nll", True, True),
    (20, 1859, 1586, 60, 4097, "adam", "lazy", "nll", False, False),
    (106, 9000, 16384, 256, 12289, "sgd", "tf1", "nll", True, False),
    (2, 248, 203, 252, 1025, "sgd", "tf1", "mse", False, False),
]


def test_adam_table_drift_is_float32_arithmetic_not_the_kernels():
    """north_star's 1e-5 is on loss / RMSE.  The GPU parity tests hold Adam-updated TABLES to 2e-4 (x a
    run-length factor) and SGD tables to 4e-5.  This test shows that tolerance is the arithmetic's: the
    oracle itself, run in float32 (sequential NumPy sums, no GPU, no kernels), drifts from its float64 self
    by up to ~1.6e-4 on the same shapes under Adam - a handful of elements whose gradient nearly cancels,
    where g / (sqrt(v) + eps) turns an fp32 rounding of g into a move of order lr - while logits and loss
    stay at 1e-7 and SGD through the same code stays below 1e-6."""
    worst = {}
    for n, U, I, D, B, opt, mode, loss, item_abs, reg_bias in _DRIFT_CASES:
        rs = np.random.RandomState(1000 + n)                   # the same draws as test_random_shapes_two_steps
        t = rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
        kw = dict(loss=loss, item_abs=item_abs, reg_bias=reg_bias, optimizer=opt, adam_mode=mode, lr=3e-3, reg=0.02)
        o64, o32 = make_oracle(U, I, D, t, **kw), make_oracle(U, I, D, t, dtype=np.float32, **kw)
        for s in range(2):
            u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < 0.5).astype(np.float32) if loss == "nll" else rs.randint(1, 6, B).astype(np.float32)
            l64, l32 = o64.train_step(u, i, r), o32.train_step(u, i, r)
            assert rel_err(l32[0], l64[0]) <= 1e-5 and rel_err(l32[1], l64[1]) <= 1e-5          # logits, loss: the north-star bound holds
        drift = max(rel_err(o32.tables()[tid], o64.tables()[tid]) for tid in o64.tables())
        run = 7.0 * B / max(1, min(U, I))
        tol = (2e-4 if opt == "adam" else 4e-5) * max(1.0, np.sqrt(run / 64))                   # what the GPU tests allow
        assert drift <= tol, (n, drift, tol)
        worst[opt] = max(worst.get(opt, 0.0), drift)
        if n == 107:
            # localise it: the worst element is one whose float64 update is far below the typical one (near-cancelling gradient)
            d = np.abs(o32.P.astype(np.float64) - o64.P)
            k = np.unravel_index(np.argmax(d), d.shape)
            moved = np.abs(o64.P - t["P"].astype(np.float64))
            assert moved[k] < 0.5 * np.median(moved[moved > 0])
    assert worst["adam"] > 5e-5          # a pure float32 restatement cannot meet 1e-5 on Adam tables ...
    assert worst["sgd"] < 2e-6           # ... while SGD through the same restatement is at rounding level


def test_c_fm_forward_matches_the_numpy_statement_and_the_explicit_pairwise_sum():
    """oracle/svd_oracle.c fmo_forward (the all-core CPU baseline of BASELINE configs[4]) against oracle/svd_oracle.py's
    fm_forward and, on a few rows, against the explicit sum over pairs of forward.py:21-22's model."""
    from oracle import c_oracle
    rs = np.random.RandomState(4)
    F, D, n, nnz = 700, 24, 300, 7
    W, V = rs.normal(0, .3, F).astype(np.float32), rs.normal(0, .2, (F, D)).astype(np.float32)
    indices = rs.randint(0, F, n * nnz).astype(np.int32)
    data = rs.randint(1, 4, n * nnz).astype(np.float32)
    indptr = np.arange(n + 1, dtype=np.int64) * nnz
    got = c_oracle.fm_forward(0.25, W, V, indptr, indices, data)
    want = so.fm_forward(np.float64(0.25), W.astype(np.float64), V.astype(np.float64), indptr, indices, data)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for r in range(5):
        f, x = indices[r * nnz:(r + 1) * nnz], data[r * nnz:(r + 1) * nnz].astype(np.float64)
        pair = sum(x[a] * x[b] * float(V[f[a]].astype(np.float64) @ V[f[b]].astype(np.float64)) for a in range(nnz) for b in range(a + 1, nnz))
        assert abs(got[r] - (0.25 + float(x @ W[f].astype(np.float64)) + pair)) <= 1e-10
    with pytest.raises(IndexError):
        c_oracle.fm_forward(0.0, W, V, indptr, indices + F, data)
