#!/usr/bin/env python3
"""One-off fuzz (GPU box): random small-table (or, with a third argument "big", large-table) shapes, one or two training steps each, the HIP path
(single-step and multi-step look-ahead) against the NumPy oracle.  Not part of the test suite - a
wider net than tests/test_gpu_parity.py::test_random_shapes_two_steps, run by hand:
    python tools/fuzz_small_tables.py [n_cases] [seed] [big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
from tests.util import RTOL, dup_heavy_ids, make_oracle, rand_tables, rel_err

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
big = len(sys.argv) > 3 and sys.argv[3] == "big"       # rows beyond the LDS-bin limit: radix sort, stream look-ahead
rs = np.random.RandomState(seed)
dims = [1, 3, 4, 8, 12, 16, 20, 32, 48, 64, 96, 100, 128, 192, 252, 256]
bad = 0
for n in range(n_cases):
    D = int(rs.choice(dims))
    U = int(rs.choice([1, 5, 300, 2048, 6040, 16384, int(rs.randint(1, 16385))]))
    I = int(rs.choice([1, 7, 200, 4096, 3952, 16384, int(rs.randint(1, 16385))]))
    B = int(rs.choice([1, 2, 63, 64, 65, 1023, 1024, 1025, 2048, 5000, 10000, 12288, 12289, 16384, int(rs.randint(1, 16385))]))
    opt, mode = [("adam", "tf1"), ("adam", "lazy"), ("sgd", "tf1")][rs.randint(3)]
    if big:
        U = int(rs.choice([16385, 20000, 70000, 200000]))
        I = int(rs.choice([1, 300, 16384, 16385, 50000]))
        B = int(rs.choice([1, 31, 1000, 1024, 4097, 20000, 50000]))
    kw = dict(loss=["mse", "nll"][rs.randint(2)], item_abs=bool(rs.randint(2)), reg_bias=bool(rs.randint(2)),
              optimizer=opt, adam_mode=mode, lr=3e-3, reg=0.02)
    t = rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
    N = 20000
    su, si = dup_heavy_ids(rs, U, N), dup_heavy_ids(rs, I, N)
    sr = (rs.rand(N) < 0.5).astype(np.float32) if kw["loss"] == "nll" else rs.randint(1, 6, N).astype(np.float32)
    ids = rs.randint(0, N, (3, B))
    orc = make_oracle(U, I, D, t, **kw)
    tag = "case %d U=%d I=%d D=%d B=%d %s" % (n, U, I, D, B, kw)
    try:
        with T.SvdModel(U, I, D, **kw) as a, T.SvdModel(U, I, D, **kw) as b:
            for m in (a, b):
                m.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
            a.upload_triples(su, si, sr)
            la = a.train_steps_resident(ids, B)
            lb = [b.train_step(su[k], si[k], sr[k])[1] for k in ids]
            want = [orc.train_step(su[k], si[k], sr[k])[1] for k in ids]
            ok = np.array_equal(np.asarray(la, np.float32), np.asarray(lb, np.float32))
            ta, tb = a.tables(), b.tables()
            ok = ok and all(np.array_equal(ta[x], tb[x]) for x in (L.MU, L.BU, L.BI, L.P, L.Q))
            run = 7.0 * B / max(1, min(U, I))
            tol = (2e-4 if opt == "adam" else 4 * RTOL) * max(1.0, np.sqrt(run / 64)) * 3
            errs = [rel_err(tb[x], orc.tables()[x]) for x in (L.MU, L.BU, L.BI, L.P, L.Q)]
            if kw["item_abs"] and errs[4] > tol:
                # |item_features| makes dQ discontinuous at 0 (sign(Q)): an element that rounding puts on the other
                # side of zero after an earlier step moves by ~2*lr under Adam.  Isolated elements only.
                wq = np.asarray(orc.tables()[L.Q], np.float64)
                dq = np.abs(np.asarray(tb[L.Q], np.float64) - wq) / max(np.abs(wq).max(), 1e-30)
                nbad = int((dq > tol).sum())
                if nbad <= max(2, dq.size // 200000):
                    print("note: case %d: %d of %d item_features elements flipped sign at 0 (|item| kink), max rel %.1e" % (n, nbad, dq.size, dq.max()), flush=True)
                    errs[4] = 0.0
            lerr = max(abs(x - y) / max(abs(y), 1e-6) for x, y in zip(lb, want))
            if not ok or max(errs) > tol or lerr > 1e-4:
                bad += 1
                print("MISMATCH", tag, "pipelines equal:", ok, "table rel err:", ["%.2e" % e for e in errs], "tol %.1e" % tol, "loss err %.1e" % lerr, flush=True)
    except Exception as e:                                 # noqa: BLE001 - report and go on
        bad += 1
        print("ERROR", tag, repr(e), flush=True)
    if n % 20 == 19:
        print("... %d cases, %d bad" % (n + 1, bad), flush=True)
print("fuzz done: %d cases, %d bad" % (n_cases, bad))
sys.exit(1 if bad else 0)
