"""ALS.  CPU: the NumPy restatement against the trajectory recorded from the REAL
reference class (parity pinned).  GPU: the HIP path against the same golden."""
import contextlib
import io

import numpy as np
import pytest

from oracle.als_oracle import AlsOracle

CASES = ("small", "d8")


def _case(g, name):
    U, W, d, iters = (int(x) for x in g[name + "/shape"])
    return U, W, d, iters, float(g[name + "/lam"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_run(golden, name):
    g = golden("als_trajectory.npz")
    U, W, d, iters, lam = _case(g, name)
    o = AlsOracle(U, W, d, iters, lam)
    np.random.seed(7)
    o.init_vars()
    o.load(g[name + "/X"], g[name + "/y"])
    for _ in range(iters):
        o.sweep()
    assert np.array_equal(o.U, g[name + "/U"]) and np.array_equal(o.V, g[name + "/V"])
    assert np.array_equal(o.W_user, g[name + "/W_user"]) and np.array_equal(o.W_work, g[name + "/W_work"])
    assert o.bias == float(g[name + "/bias"])
    assert np.allclose(o.predict(g[name + "/Xt"]), g[name + "/pred"], rtol=0, atol=1e-13)
    assert abs(o.compute_rmse(g[name + "/yt"], o.predict(g[name + "/Xt"])) - float(g[name + "/rmse"])) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_als_matches_reference_run(golden, name):
    import tfrecomm_amd as T
    g = golden("als_trajectory.npz")
    U, W, d, iters, lam = _case(g, name)
    als = T.MangakiALS3(nb_components=d, nb_iterations=iters, lambda_=lam)
    als.nb_users, als.nb_works = U, W
    np.random.seed(7)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        als.fit(g[name + "/X"], g[name + "/y"], g[name + "/yt"], g[name + "/Xt"])
    assert buf.getvalue().count("Step") == iters                      # als3.py:31
    st = als.state()
    for k in ("U", "V", "W_user", "W_work"):
        err = np.abs(st[k] - g[name + "/" + k]).max() / np.abs(g[name + "/" + k]).max()
        assert err < 1e-9, "%s: %.3e" % (k, err)                      # float64 both sides; LU vs Cholesky rounding
    assert st["bias"] == float(g[name + "/bias"])
    pred = als.predict(g[name + "/Xt"])
    assert np.allclose(pred, g[name + "/pred"], rtol=1e-9, atol=1e-9)
    assert abs(als.compute_rmse(g[name + "/yt"], pred) - float(g[name + "/rmse"])) < 1e-9
    assert als.get_shortname() == "als3-%d" % d
    with pytest.raises(IndexError):
        als.predict(np.array([[U, 0]]))
    als.close()


@pytest.mark.gpu
def test_gpu_als_larger_vs_oracle_and_deterministic():
    import tfrecomm_amd as T
    rs = np.random.RandomState(3)
    U, W, n, d = 700, 500, 60000, 20
    X = np.stack([rs.randint(0, U, n), rs.randint(0, W, n)], 1)
    y = rs.randint(1, 6, n).astype(np.float64)
    Xt = np.stack([rs.randint(0, U, 1000), rs.randint(0, W, 1000)], 1)
    yt = rs.randint(1, 6, 1000).astype(np.float64)
    o = AlsOracle(U, W, d, 2, 0.1)
    np.random.seed(11)
    o.init_vars()
    o.load(X, y)
    o.sweep(); o.sweep()
    outs = []
    for rep in range(2):
        als = T.MangakiALS3(nb_components=d, nb_iterations=2, lambda_=0.1, verbose=False)
        als.nb_users, als.nb_works = U, W
        np.random.seed(11)
        als.fit(X, y, yt, Xt)
        outs.append(als.state())
        als.close()
    assert np.array_equal(outs[0]["U"], outs[1]["U"]) and np.array_equal(outs[0]["W_work"], outs[1]["W_work"])
    assert np.abs(outs[0]["U"] - o.U).max() < 1e-9 and np.abs(outs[0]["V"] - o.V).max() < 1e-9
    assert np.abs(outs[0]["W_user"] - o.W_user).max() < 1e-9


@pytest.mark.gpu
def test_gpu_als_long_lists_are_chunked_and_match_oracle():
    """One work holds a third of all ratings and one user a tenth: their lists are far longer than the
    chunk size, so their normal equations come from several blocks' partial sums."""
    import tfrecomm_amd as T
    rs = np.random.RandomState(5)
    U, W, n, d = 400, 300, 40000, 12
    u = rs.randint(0, U, n); w = rs.randint(0, W, n)
    w[rs.rand(n) < 0.33] = 7                      # ~13000 ratings of one work
    u[rs.rand(n) < 0.10] = 3                      # ~4000 ratings of one user
    X = np.stack([u, w], 1)
    y = rs.randint(1, 6, n).astype(np.float64)
    Xt = np.stack([rs.randint(0, U, 500), rs.randint(0, W, 500)], 1)
    yt = rs.randint(1, 6, 500).astype(np.float64)
    o = AlsOracle(U, W, d, 2, 0.1)
    np.random.seed(21)
    o.init_vars()
    o.load(X, y)
    o.sweep(); o.sweep()
    als = T.MangakiALS3(nb_components=d, nb_iterations=2, lambda_=0.1, verbose=False)
    als.nb_users, als.nb_works = U, W
    np.random.seed(21)
    als.fit(X, y, yt, Xt)
    st = als.state()
    als.close()
    for k, want in (("U", o.U), ("V", o.V), ("W_user", o.W_user), ("W_work", o.W_work)):
        assert np.abs(st[k] - want).max() < 1e-9, k
