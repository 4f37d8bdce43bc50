"""FM second-order forward (BASELINE config 5) through the C-ABI against the float64 oracle
(forward.py:21-22 generalised to count-valued features) and the committed golden rows."""
import numpy as np
import pytest
import scipy.sparse as sp

import tfrecomm_amd as T
from oracle import svd_oracle as so
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def test_fm_golden(golden):
    g = golden("fm_forward.npz")
    keys = sorted({k.split("/")[0] for k in g.files})
    assert len(keys) == 3
    for key in keys:
        V, W, mu = g[key + "/V"], g[key + "/W"], float(g[key + "/mu"])
        with T.FmModel(V.shape[0], V.shape[1]) as m:
            m.set(mu, W, V)
            y = m.forward_csr(g[key + "/indptr"], g[key + "/indices"], g[key + "/data"])
        assert y.dtype == np.float32
        assert_close(y, g[key + "/y"], what=key)
        assert abs(y[0] - mu) < 1e-6                      # the empty row predicts mu


@pytest.mark.parametrize("F,D,n,nnz", [(5000, 64, 3000, 8), (777, 15, 500, 3), (100000, 128, 4096, 12), (64, 4, 10, 64)])
def test_fm_random_csr_matches_oracle_and_scipy_path(F, D, n, nnz):
    rs = np.random.RandomState(F + D)
    V = rs.normal(0, 0.1, (F, D)).astype(np.float32)
    W = rs.normal(0, 0.1, F).astype(np.float32)
    mu = np.float32(0.37)
    X = sp.random(n, F, density=min(1.0, nnz / F), format="csr", random_state=rs,
                  data_rvs=lambda k: rs.randint(1, 4, k).astype(np.float64)).astype(np.float32)
    with T.FmModel(F, D) as m:
        m.set(mu, W, V)
        y = m.fma(X)                                       # forward.py's fma(x)
        y2 = m.fma(X.tocoo())
    want = so.fm_forward(np.float64(mu), W.astype(np.float64), V.astype(np.float64), X.indptr, X.indices,
                         X.data.astype(np.float64))
    assert_close(y, want, rtol=2e-5, what="fm")
    assert np.array_equal(y, y2)
    # on 0/1 features the reference's literal x.dot(V**2) form gives the same numbers
    Xb = X.copy()
    Xb.data[:] = 1.0
    with T.FmModel(F, D) as m:
        m.set(mu, W, V)
        yb = m.fma(Xb)
    ref = so.fm_forward_reference_form(np.float64(mu), W.astype(np.float64), V.astype(np.float64), Xb.indptr, Xb.indices,
                                       Xb.data.astype(np.float64))
    assert_close(yb, ref, rtol=2e-5, what="fm binary")


def test_fm_config5_full_size_properties():
    """BASELINE config 5 at its full table size (1M features x 64: V = 256 MB, read by the non-temporal variant of the
    kernel) on 2^18 rows x 8 non-zeros: a row's prediction depends on its own non-zeros only, so (i) 4096 rows drawn from
    the batch equal the oracle on those rows alone, (ii) the batch in another row order gives the same numbers, bit for
    bit, in that order, (iii) the empty row predicts mu."""
    F, D, n, nnz = 1_000_000, 64, 1 << 18, 8
    rs = np.random.RandomState(5)
    V = rs.normal(0, 0.1, (F, D)).astype(np.float32)
    W = rs.normal(0, 0.1, F).astype(np.float32)
    mu = np.float32(-0.21)
    indptr = np.arange(0, (n + 1) * nnz, nnz, dtype=np.int64)
    indptr[1:] -= nnz                                     # row 0 is empty
    indptr[0] = 0
    indices = rs.randint(0, F, (n - 1) * nnz).astype(np.int32)
    data = rs.randint(1, 4, (n - 1) * nnz).astype(np.float32)
    with T.FmModel(F, D) as m:
        m.set(mu, W, V)
        y = m.forward_csr(indptr, indices, data)
        pick = np.concatenate([[0], rs.choice(np.arange(1, n), 4095, replace=False)])
        sub_ptr = np.zeros(pick.size + 1, np.int64)
        sub_ptr[1:] = np.cumsum(indptr[pick + 1] - indptr[pick])
        sub_idx = np.concatenate([indices[indptr[r]:indptr[r + 1]] for r in pick])
        sub_val = np.concatenate([data[indptr[r]:indptr[r + 1]] for r in pick])
        want = so.fm_forward(np.float64(mu), W.astype(np.float64), V.astype(np.float64), sub_ptr, sub_idx, sub_val.astype(np.float64))
        assert_close(y[pick], want, rtol=2e-5, what="fm rows against the oracle")
        assert abs(y[0] - mu) < 1e-6
        perm = rs.permutation(n)
        X = sp.csr_matrix((data, indices, indptr), shape=(n, F))[perm]
        y2 = m.forward_csr(X.indptr.astype(np.int64), X.indices.astype(np.int32), X.data.astype(np.float32))
        assert np.array_equal(y2, y[perm])


def test_fm_errors():
    with T.FmModel(100, 8) as m:
        m.init(seed=1)
        with pytest.raises(IndexError):
            m.forward_csr([0, 1], [100], [1.0])
        with pytest.raises(T.TfrError):
            m.forward_csr([0, 2, 1], [1, 2], [1.0, 1.0])
        assert m.forward_csr([0], [], []).shape == (0,)
        with pytest.raises(ValueError):
            m.fma(sp.identity(7, format="csr"))
    with pytest.raises(T.TfrError):
        T.FmModel(10, 300)


# ------------------------------------------------------------------ FM training
def _random_design(rs, n, F, nnz, hot=20):
    """CSR rows with a few very hot features (long runs in the backward) and an empty row."""
    rows, cols, vals = [], [], []
    for r in range(n):
        k = 0 if r == 3 else rs.randint(1, nnz + 1)
        picks = set(rs.randint(0, hot, k // 2 + 1).tolist()[:k]) | set(rs.randint(0, F, k).tolist())
        picks = sorted(picks)[:k]
        rows += [r] * len(picks)
        cols += picks
        vals += rs.randint(1, 4, len(picks)).tolist()
    return sp.csr_matrix((np.array(vals, np.float32), (rows, cols)), shape=(n, F))


@pytest.mark.parametrize("loss", ["mse", "nll"])
@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
@pytest.mark.parametrize("F,D,n,nnz", [(400, 16, 700, 6), (5000, 64, 3000, 8), (90, 5, 257, 4)])
def test_fm_training_matches_oracle(loss, optimizer, F, D, n, nnz):
    rs = np.random.RandomState(F + n)
    V0 = rs.normal(0, 0.1, (F, D)).astype(np.float32)
    W0 = rs.normal(0, 0.1, F).astype(np.float32)
    mu0 = np.float32(0.1)
    lr, lam = (0.02, 0.01) if optimizer == "sgd" else (0.002, 0.01)   # Adam error scales with lr (sign-like first steps)
    V, W, mu = V0.astype(np.float64), W0.astype(np.float64), np.float64(mu0)
    state = so.fm_adam_state(F, D) if optimizer == "adam" else None
    with T.FmModel(F, D, loss=loss, optimizer=optimizer, lr=lr, reg=lam) as m:
        m.set(mu0, W0, V0)
        for s in range(3):
            X = _random_design(rs, n, F, nnz)
            y = (rs.rand(n) < 0.5).astype(np.float32) if loss == "nll" else rs.normal(0, 1, n).astype(np.float32)
            pred, lossv = m.train_step(X, y)
            want_pred, want_loss, mu = so.fm_train_step(mu, W, V, X.indptr, X.indices, X.data.astype(np.float64),
                                                        y.astype(np.float64), lr, lam, loss, optimizer, state)
            # hot features sum ~400 fp32 terms per step; Adam's normalisation amplifies their rounding
            tol = (1e-4 if optimizer == "adam" else 3e-5) * (s + 1)
            assert_close(pred, want_pred, rtol=tol, what="pred step %d" % s)
            assert_close(lossv, want_loss, rtol=tol, what="loss step %d" % s)
        gmu, gW, gV = m.get()
    assert_close(gV, V, rtol=4e-4, what="V")
    assert_close(gW, W, rtol=4e-4, what="W")
    assert abs(gmu - mu) <= 2e-4 * max(1.0, abs(mu))
    assert not np.array_equal(gV, V0)


def test_fm_training_is_deterministic_and_learns():
    rs = np.random.RandomState(1)
    F, D, n = 2000, 16, 4000
    Vt = rs.normal(0, 0.3, (F, 4))
    X = _random_design(rs, n, F, 6)
    logit = np.array([0.5 * ((X[r].toarray() @ Vt) ** 2).sum() - 1.0 for r in range(n)])
    y = (rs.rand(n) < 1 / (1 + np.exp(-logit))).astype(np.float32)
    outs = []
    for rep in range(2):
        with T.FmModel(F, D, loss="nll", optimizer="adam", lr=0.05, reg=0.001) as m:
            m.init(seed=4, stddev=0.05)
            losses = [m.train_step(X, y)[1] for _ in range(30)]
            outs.append((losses, m.get()))
    assert outs[0][0] == outs[1][0]                                  # bit-identical run to run
    assert np.array_equal(outs[0][1][2], outs[1][1][2])
    assert outs[0][0][-1] < 0.8 * outs[0][0][0]                      # the data loss falls


_FM_LOAD_ROUNDS_SCRIPT = r"""
import hashlib, os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, %r)
import tfrecomm_amd as T
opt = os.environ["TFR_TEST_OPT"]
F, D, n, nnz = 20000, 64, 30000, 8
rs = np.random.RandomState(7)
with T.FmModel(F, D, loss="nll", optimizer=opt, lr=0.02 if opt == "sgd" else 0.002, reg=0.01) as m:
    m.init(seed=3, stddev=0.05)
    h = hashlib.sha256()
    for s in range(3):
        cols = np.where(rs.rand(n, nnz) < 0.3, rs.randint(0, 12, (n, nnz)), rs.randint(0, F, (n, nnz)))      # hot features
        cols = np.sort(cols, axis=1)
        X = sp.csr_matrix((rs.rand(n * nnz).astype(np.float32) + 0.5, cols.reshape(-1), np.arange(n + 1) * nnz), shape=(n, F))
        y = (rs.rand(n) < 0.5).astype(np.float32)
        pred, loss = m.train_step(X, y)
        h.update(np.asarray(pred).tobytes()); h.update(np.float64(loss).tobytes())
    mu, W, V = m.get()
    h.update(np.float32(mu).tobytes()); h.update(np.ascontiguousarray(W).tobytes()); h.update(np.ascontiguousarray(V).tobytes())
    print("HASH", h.hexdigest())
"""


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_fm_backward_three_round_load_form_is_bit_identical_to_the_general_form(optimizer):
    """TFR_FAST=0: the FM backward through the general form of k_seg_reduce; default: the three-round form (entry records).
    Predictions, losses, W and V after three steps on rows with hot features hash identically."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for flag in ("1", "0"):
        env = dict(os.environ, TFR_FAST=flag, TFR_TEST_OPT=optimizer)
        p = subprocess.run([sys.executable, "-c", _FM_LOAD_ROUNDS_SCRIPT % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out.append([l for l in p.stdout.decode().splitlines() if l.startswith("HASH")][0])
    assert out[0] == out[1]
