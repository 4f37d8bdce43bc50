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
