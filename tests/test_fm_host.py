"""FM host code and oracle (CPU): the design-matrix builder against a hand-built expectation
(fm.py:61-93) and the FM training oracle against torch autograd."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import svd_oracle as so
from tfrecomm_amd.fm import df_to_sparse


def test_design_matrix_blocks_and_order():
    df = {"user": np.array([0, 2, 1, 2]), "item": np.array([1, 0, 1, 2]), "outcome": np.array([1, 0, 1, 1.0]),
          "wins": np.array([0, 1, 2, 0.0]), "fails": np.array([3, 0, np.nan, 1.0])}
    X = df_to_sparse(df, 3, 3, ["items", "users"]).toarray()          # order follows AGENTS, not the argument
    want = np.zeros((4, 6), np.float32)
    for r, (u, i) in enumerate(zip(df["user"], df["item"])):
        want[r, u] = 1
        want[r, 3 + i] = 1
    assert X.dtype == np.float32 and np.array_equal(X, want)
    X2 = df_to_sparse(df, 3, 3, ["users", "items", "item_wins", "item_fails"]).toarray()
    assert X2.shape == (4, 12)
    assert np.array_equal(X2[:, 6:9], np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 0]], np.float32))
    assert np.array_equal(X2[:, 9:12], np.array([[0, 3, 0], [0, 0, 0], [0, 0, 0], [0, 0, 1]], np.float32))   # NaN -> 0
    q = sp.csr_matrix(np.array([[1, 1, 0, 0], [0, 1, 0, 1], [0, 0, 1, 0]], np.float32))    # 3 items x 4 skills
    X3 = df_to_sparse(df, 3, 3, ["users", "skills"], qmatrix=q).toarray()
    assert np.array_equal(X3[:, 3:], q.toarray()[df["item"]])
    with pytest.raises(ValueError):
        df_to_sparse(df, 3, 3, ["users", "attempts"])


@pytest.mark.parametrize("loss", ["mse", "nll"])
def test_fm_training_oracle_matches_autograd(loss):
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(0)
    F, D, n = 30, 4, 12
    V, W, mu = rs.normal(0, .3, (F, D)), rs.normal(0, .3, F), 0.2
    lens = rs.randint(0, 5, n)
    indptr = np.concatenate(([0], np.cumsum(lens)))
    indices = np.concatenate([rs.choice(F, l, replace=False) for l in lens] + [np.zeros(0, int)]).astype(np.int64)
    data = rs.randint(1, 4, indices.size).astype(float)
    y = (rs.rand(n) < .5).astype(float)
    V2, W2 = V.copy(), W.copy()
    lr, lam = 0.1, 0.05
    yh, l, mu2 = so.fm_train_step(mu, W2, V2, indptr, indices, data, y, lr, lam, loss)
    tV, tW, tmu = [torch.tensor(a, requires_grad=True) for a in (V, W, np.array(mu))]
    rows = np.repeat(np.arange(n), np.diff(indptr))
    tx, tf, tr = torch.tensor(data), torch.tensor(indices), torch.tensor(rows)
    xv = tx[:, None] * tV[tf]
    S = torch.zeros(n, D, dtype=torch.float64).index_add(0, tr, xv)
    Q = torch.zeros(n, dtype=torch.float64).index_add(0, tr, (xv ** 2).sum(1))
    lin = torch.zeros(n, dtype=torch.float64).index_add(0, tr, tx * tW[tf])
    yhat = tmu + lin + 0.5 * ((S ** 2).sum(1) - Q)
    dl = 0.5 * ((yhat - torch.tensor(y)) ** 2).sum() if loss == "mse" else \
        torch.nn.functional.binary_cross_entropy_with_logits(yhat, torch.tensor(y), reduction="sum")
    reg = 0.5 * (tV[tf] ** 2).sum() + 0.5 * (tW[tf] ** 2).sum()       # per-occurrence L2, as in the SVD regulariser
    (dl + lam * reg).backward()
    assert np.abs((V - V2) / lr - tV.grad.numpy()).max() < 1e-12
    assert np.abs((W - W2) / lr - tW.grad.numpy()).max() < 1e-12
    assert abs((mu - mu2) / lr - tmu.grad.item()) < 1e-12 and abs(l - dl.item()) < 1e-12
    assert np.allclose(yh, yhat.detach().numpy(), atol=1e-12)
