"""Shared helpers for the parity tests."""
import numpy as np

from oracle import svd_oracle as so

TABLE_NAMES = {so.MU: "mu", so.BU: "bu", so.BI: "bi", so.PF: "P", so.QF: "Q"}

# north_star: "fp32 loss/RMSE match the TF-CPU reference within 1e-5 relative"
RTOL = 1e-5


def rand_tables(rs, U, I, D, scale=0.3):
    return dict(mu=np.float32(rs.uniform(-1, 1)), bu=rs.normal(0, 0.5, U).astype(np.float32),
                bi=rs.normal(0, 0.5, I).astype(np.float32), P=rs.normal(0, scale, (U, D)).astype(np.float32),
                Q=rs.normal(0, scale, (I, D)).astype(np.float32))


def dup_heavy_ids(rs, n, B):
    hot = rs.randint(0, n, max(1, n // 10))
    pick = rs.rand(B) < 0.7
    return np.where(pick, hot[rs.randint(0, hot.size, B)], rs.randint(0, n, B)).astype(np.int32)


def make_oracle(U, I, D, tables, dtype=np.float64, **kw):
    frozen = kw.pop("frozen", 0)
    o = so.SvdOracle(U, I, D, dtype=dtype, **kw)
    o.set_tables(*(np.asarray(tables[k], dtype=dtype) for k in ("mu", "bu", "bi", "P", "Q")))
    o.frozen = frozen
    return o


def rel_err(got, want):
    """max |got-want| / max(|want|) - the scale-relative error of an array."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if want.size == 0:
        return 0.0
    scale = max(np.abs(want).max(), 1e-30)
    return float(np.abs(got - want).max() / scale)


def assert_close(got, want, rtol=RTOL, what=""):
    e = rel_err(got, want)
    assert e <= rtol, "%s: scale-relative error %.3e > %.1e" % (what, e, rtol)
