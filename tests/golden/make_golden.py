#!/usr/bin/env python3
"""Generates the committed golden fixtures in this directory.  Run from the repo root in the
BUILD container (needs /root/reference for part 4 only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is a fixture here: inputs and expected outputs only (small .npz).
  1. svd_forward_grad.npz  forward / loss / regulariser / reduced gradients for all 2x2x2 mode
                           switches at four shapes - float64 truth (oracle/svd_oracle.py).
  2. svd_trajectories.npz  5-step trajectories for SGD, Adam-tf1, Adam-lazy with duplicate-heavy
                           batches and one all-identical-id batch; var_list (frozen) case.
  3. svd_segments.npz      stable-sort / segment structures for id batches (integer, bit-exact).
  4. iter_streams.npz      ShuffleIterator / OneEpochIterator outputs produced by importing the
                           REAL reference module /root/reference/dataio.py (it needs only
                           numpy/pandas/yaml), with np.random.seed(13575) as
                           svd_train_val.py:15 sets it.  This is the one part of the path
                           whose parity is pinned by the reference's own code.
  5. fm_forward.npz        FM second-order forward on random CSR rows (binary and count-valued).
  6. als_trajectory.npz    ALS fits produced by running the REAL reference class
                           /root/reference/als3.py (numpy/scipy/sklearn only), np.random.seed(7).

The SVD arithmetic (1-3) is "parity unpinned": TensorFlow is not installable here, so the
expected values come from our float64 restatement (cross-checked against torch autograd in
tests/test_oracle.py), not from a run of the reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import svd_oracle as so  # noqa: E402

SHAPES = [(7, 5, 3, 4), (50, 40, 15, 64), (300, 200, 64, 257), (300, 200, 128, 512)]


def rand_tables(rs, U, I, D):
    return dict(P=rs.normal(0, 0.3, (U, D)), Q=rs.normal(0, 0.3, (I, D)), bu=rs.normal(0, 0.5, U),
                bi=rs.normal(0, 0.5, I), mu=np.array(rs.uniform(-1, 1)))


def dup_heavy_ids(rs, n, B):
    """Zipf-ish: a few rows take most of the batch (thousands of duplicates at scale)."""
    hot = rs.randint(0, n, max(1, n // 10))
    pick = rs.rand(B) < 0.7
    return np.where(pick, hot[rs.randint(0, hot.size, B)], rs.randint(0, n, B)).astype(np.int32)


def part1():
    rs = np.random.RandomState(20240601)
    out = {}
    for (U, I, D, B) in SHAPES:
        t = rand_tables(rs, U, I, D)
        u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
        r_mse = rs.randint(1, 6, B).astype(np.float64)
        r_nll = (rs.rand(B) < 0.5).astype(np.float64)
        key = "U%d_I%d_D%d_B%d" % (U, I, D, B)
        for k, v in t.items():
            out[key + "/" + k] = v.astype(np.float32)
        out[key + "/u"], out[key + "/i"] = u, i
        out[key + "/r_mse"], out[key + "/r_nll"] = r_mse.astype(np.float32), r_nll.astype(np.float32)
        t64 = {k: v.astype(np.float32).astype(np.float64) for k, v in t.items()}   # truth on the f32 inputs
        for loss in (so.MSE, so.NLL):
            r = r_mse if loss == so.MSE else r_nll
            for ia in (0, 1):
                for rb in (0, 1):
                    lam = 0.05
                    lg = so.forward(t64["P"], t64["Q"], t64["bu"], t64["bi"], t64["mu"], u, i, bool(ia))
                    g = so.dlogits(lg, r, loss)
                    dP, dQ, dbu, dbi, dmu = so.occurrence_grads(t64["P"], t64["Q"], t64["bu"], t64["bi"], u, i, g,
                                                                lam, bool(ia), bool(rb))
                    tag = "%s/%s_abs%d_rb%d" % (key, loss, ia, rb)
                    out[tag + "/logits"] = lg
                    out[tag + "/infer"] = so.head(lg, loss)
                    out[tag + "/loss"] = np.array(so.data_loss(lg, r, loss))
                    out[tag + "/reg"] = np.array(so.regularizer(t64["P"], t64["Q"], t64["bu"], t64["bi"], u, i, bool(rb)))
                    uu, inv = so.dedup(u)
                    ii, inv_i = so.dedup(i)
                    out[tag + "/uniq_u"], out[tag + "/uniq_i"] = uu, ii
                    out[tag + "/gP"] = so.segment_sum(dP, inv, uu.size)
                    out[tag + "/gQ"] = so.segment_sum(dQ, inv_i, ii.size)
                    out[tag + "/gbu"] = so.segment_sum(dbu, inv, uu.size)
                    out[tag + "/gbi"] = so.segment_sum(dbi, inv_i, ii.size)
                    out[tag + "/gmu"] = np.array(dmu)
    np.savez_compressed(os.path.join(HERE, "svd_forward_grad.npz"), **out)


TRAJ_CASES = [
    # name, U, I, D, B, kwargs
    ("adam_tf1_mse", 40, 30, 15, 48, dict(optimizer="adam", adam_mode="tf1", loss="mse", lr=1e-3, reg=0.05)),
    ("adam_lazy_mse", 40, 30, 64, 48, dict(optimizer="adam", adam_mode="lazy", loss="mse", lr=1e-3, reg=0.05)),
    ("sgd_nll_fork", 40, 30, 20, 48, dict(optimizer="sgd", loss="nll", item_abs=True, reg_bias=True, lr=5e-3, reg=0.01)),
    ("adam_tf1_nll_abs", 25, 35, 128, 70, dict(optimizer="adam", adam_mode="tf1", loss="nll", item_abs=True, reg_bias=True, lr=1e-2, reg=0.01)),
    ("adam_lazy_frozen", 40, 30, 8, 48, dict(optimizer="adam", adam_mode="lazy", loss="mse", lr=1e-2, reg=0.05, frozen=(1 << so.MU) | (1 << so.BI) | (1 << so.QF))),
    ("sgd_mse", 33, 21, 5, 31, dict(optimizer="sgd", loss="mse", lr=1e-2, reg=0.02)),
]
NSTEPS = 5


def part2():
    rs = np.random.RandomState(20240602)
    out = {}
    for name, U, I, D, B, kw in TRAJ_CASES:
        kw = dict(kw)
        frozen = kw.pop("frozen", 0)
        t = rand_tables(rs, U, I, D)
        t32 = {k: v.astype(np.float32) for k, v in t.items()}
        orc = so.SvdOracle(U, I, D, dtype=np.float64, **kw)
        orc.set_tables(*(t32[k].astype(np.float64) for k in ("mu", "bu", "bi", "P", "Q")))
        orc.frozen = frozen
        for k, v in t32.items():
            out["%s/init/%s" % (name, k)] = v
        out[name + "/frozen"] = np.array(frozen)
        for s in range(NSTEPS):
            if s == 2:                                   # one batch with all-identical ids
                u = np.full(B, rs.randint(0, U), np.int32)
                i = np.full(B, rs.randint(0, I), np.int32)
            else:
                u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < 0.5).astype(np.float32) if kw["loss"] == "nll" else rs.randint(1, 6, B).astype(np.float32)
            lg, lossv, regv = orc.train_step(u, i, r)
            p = "%s/step%d/" % (name, s)
            out[p + "u"], out[p + "i"], out[p + "r"] = u, i, r
            out[p + "logits"], out[p + "loss"], out[p + "reg"] = lg, np.array(lossv), np.array(regv)
            for tid, tn in ((so.MU, "mu"), (so.BU, "bu"), (so.BI, "bi"), (so.PF, "P"), (so.QF, "Q")):
                out[p + tn] = np.array(orc.tables()[tid])
                if kw["optimizer"] == "adam":
                    out[p + tn + "_m"] = np.array(orc.slots[tid].m)
                    out[p + tn + "_v"] = np.array(orc.slots[tid].v)
    np.savez_compressed(os.path.join(HERE, "svd_trajectories.npz"), **out)


def part3():
    rs = np.random.RandomState(20240603)
    out = {}
    for n, B in ((5, 4), (40, 64), (200, 257), (6040, 10000), (1 << 20, 4096)):
        ids = dup_heavy_ids(rs, n, B)
        sk, pos, seg = so.sort_segments(ids)
        uq, inv = so.dedup(ids)
        key = "n%d_B%d" % (n, B)
        out[key + "/ids"], out[key + "/sorted_ids"], out[key + "/sorted_pos"] = ids, sk.astype(np.int32), pos
        out[key + "/seg_start"], out[key + "/unique_first_occurrence"] = seg, uq.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "svd_segments.npz"), **out)


def part4():
    """Real reference iterators (dataio.py:94-138), imported - not copied - from /root/reference."""
    sys.path.insert(0, "/root/reference")
    import dataio as ref_dataio          # numpy / pandas / yaml only
    out = {}
    for N, B, K in ((10, 4, 6), (1000, 64, 5), (900188, 1000, 3)):
        rs = np.random.RandomState(N)
        cols = [rs.randint(0, 6040, N).astype(np.int32), rs.randint(0, 3952, N).astype(np.int32),
                rs.randint(1, 6, N).astype(np.float32)]
        key = "shuffle_N%d_B%d" % (N, B)
        out[key + "/seed"] = np.array(13575)                      # svd_train_val.py:15
        out[key + "/colseed"] = np.array(N)
        np.random.seed(13575)
        it = ref_dataio.ShuffleIterator(cols, batch_size=B)
        out[key + "/len"] = np.array(len(it))
        for k in range(K):
            batch = next(it)
            for c, col in enumerate(batch):
                out["%s/batch%d/col%d" % (key, k, c)] = col
        np.random.seed(13575)
        out[key + "/ids"] = np.stack([np.random.randint(0, N, (B,)) for _ in range(K)])
    for N, B in ((10, 3), (10, -1), (1000, 64), (7, 7), (7, 10)):
        rs = np.random.RandomState(N + 1)
        cols = [rs.randint(0, 50, N).astype(np.int32), rs.randint(0, 40, N).astype(np.int32),
                rs.randint(1, 6, N).astype(np.float32)]
        key = "epoch_N%d_B%d" % (N, B)
        out[key + "/colseed"] = np.array(N + 1)
        it = ref_dataio.OneEpochIterator(cols, batch_size=B)
        for rep in range(2):                                       # it rewinds itself (dataio.py:133-135)
            batches = list(it)
            out["%s/rep%d/n" % (key, rep)] = np.array(len(batches))
            for k, batch in enumerate(batches):
                for c, col in enumerate(batch):
                    out["%s/rep%d/batch%d/col%d" % (key, rep, k, c)] = col
    np.savez_compressed(os.path.join(HERE, "iter_streams.npz"), **out)


def part5():
    rs = np.random.RandomState(20240605)
    out = {}
    for F, D, n, nnz, counts in ((1000, 8, 64, 8, False), (1000, 64, 64, 8, True), (300, 20, 33, 5, True)):
        V = rs.normal(0, 0.1, (F, D)).astype(np.float32)
        W = rs.normal(0, 0.1, F).astype(np.float32)
        mu = np.float32(rs.normal())
        lens = rs.randint(1, nnz + 1, n)
        lens[0] = 0                                                # an empty row
        indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        indices = np.concatenate([np.sort(rs.choice(F, l, replace=False)) for l in lens] + [np.zeros(0, int)]).astype(np.int32)
        data = (rs.randint(1, 5, indices.size) if counts else np.ones(indices.size)).astype(np.float32)
        key = "F%d_D%d_%s" % (F, D, "count" if counts else "binary")
        y = so.fm_forward(np.float64(mu), W.astype(np.float64), V.astype(np.float64), indptr, indices, data.astype(np.float64))
        for k, v in dict(V=V, W=W, mu=np.array(mu), indptr=indptr, indices=indices, data=data, y=y).items():
            out[key + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "fm_forward.npz"), **out)


def part6():
    """ALS trajectory from the REAL reference class (als3.py needs only numpy/scipy/sklearn)."""
    import contextlib
    import io
    sys.path.insert(0, "/root/reference")
    from als3 import MangakiALS3
    out = {}
    for name, (U, W, n, d, iters, lam) in {"small": (60, 45, 1500, 20, 3, 0.1), "d8": (120, 70, 4000, 8, 2, 0.05)}.items():
        rs = np.random.RandomState(len(name))
        X = np.stack([rs.randint(0, U, n), rs.randint(0, W, n)], 1)
        y = rs.randint(0, 6, n).astype(np.float64)                    # includes exact zeros (als3.py:29)
        Xt = np.stack([rs.randint(0, U, 300), rs.randint(0, W, 300)], 1)
        yt = rs.randint(1, 6, 300).astype(np.float64)
        als = MangakiALS3(nb_components=d, nb_iterations=iters, lambda_=lam)
        als.nb_users, als.nb_works = U, W                             # set by the caller, forward.py:32-33
        np.random.seed(7)
        with contextlib.redirect_stdout(io.StringIO()):
            als.fit(X, y, yt, Xt)
        for k, v in dict(X=X, y=y, Xt=Xt, yt=yt, U=als.U, V=als.V, W_user=als.W_user, W_work=als.W_work,
                         bias=np.array(als.bias), pred=als.predict(Xt), shape=np.array([U, W, d, iters]),
                         lam=np.array(lam), rmse=np.array(als.compute_rmse(yt, als.predict(Xt)))).items():
            out[name + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "als_trajectory.npz"), **out)


LEGEND_CASES = [dict(d=0, users=True, items=True), dict(d=5, users=True, items=True), dict(d=0, skills=True, attempts=True),
                dict(d=0, skills=True, wins=True, fails=True), dict(d=20, users=True, items=True, skills=True, wins=True, fails=True),
                dict(d=3, items=True, item_wins=True, item_fails=True, extra=True), dict(d=0), dict(d=7, skills=True, attempts=True),
                dict(d=2, users=False, items=True, wins=True)]


def part7():
    """CSV loaders pinned by the REAL reference readers (dataio.py:8-16,38-54): a small prepared-dataset folder
    (data written here - synthetic rows, not reference content) is read with the reference's own
    build_paths / read_process / get_data; the frames it returns are the fixture."""
    sys.path.insert(0, "/root/reference")
    import dataio as ref_dataio
    root = os.path.join(HERE, "csv_fixture")
    folder = os.path.join(root, "data", "tiny")
    os.makedirs(folder, exist_ok=True)
    rs = np.random.RandomState(99)
    for name, n in (("train", 23), ("val", 7), ("test", 5)):
        with open(os.path.join(folder, name + ".csv"), "w") as f:
            for _ in range(n):
                f.write("%d,%d,%d,%d,%d\n" % (rs.randint(0, 40), rs.randint(0, 30), rs.randint(0, 2), rs.randint(0, 6), rs.randint(0, 6)))
    with open(os.path.join(folder, "tabbed.tsv"), "w") as f:        # read_process's default separator is a tab
        for _ in range(4):
            f.write("%d\t%d\t%.1f\t0\t0\n" % (rs.randint(0, 40), rs.randint(0, 30), rs.randint(1, 6)))
    with open(os.path.join(folder, "config.yml"), "w") as f:
        f.write("USER_NUM: 40\nITEM_NUM: 30\nNB_CLASSES: 2\nBATCH_SIZE: 8\n")
    with open(os.path.join(folder, "all.csv"), "w") as f:           # the FM experiments' single-file layout (dataio.py:19-28,57-60)
        for _ in range(11):
            f.write("%d,%d,%d,%d,%d\n" % (rs.randint(0, 40), rs.randint(0, 30), rs.randint(0, 2), rs.randint(0, 6), rs.randint(0, 6)))
    out = {}
    cwd = os.getcwd()
    os.chdir(root)                                                  # the reference's paths are relative: data/<name>/...
    try:
        paths = ref_dataio.build_paths("tiny")
        out["paths"] = np.array([p.replace(os.sep, "/") for p in paths])
        frames = dict(zip(("train", "val", "test"), ref_dataio.get_data("tiny")))
        frames["tabbed"] = ref_dataio.read_process(os.path.join("data", "tiny", "tabbed.tsv"))
        out["new_paths"] = np.array([p.replace(os.sep, "/") for p in ref_dataio.build_new_paths("tiny")])
        frames["all"] = ref_dataio.get_new_data("tiny")
        import contextlib, io, json
        legends = []
        for args in LEGEND_CASES:                                   # dataio.py:63-87 (it prints; keep stdout clean)
            with contextlib.redirect_stdout(io.StringIO()):
                short, full, latex, active = ref_dataio.get_legend(dict(args))
            legends.append([short, full, latex, list(active)])
        out["legend_cases"] = np.array(json.dumps(LEGEND_CASES))
        out["legends"] = np.array(json.dumps(legends))
    finally:
        os.chdir(cwd)
    for name, df in frames.items():
        out[name + "/columns"] = np.array(list(df.columns))
        for c in df.columns:
            out["%s/%s" % (name, c)] = df[c].to_numpy()
            out["%s/%s/dtype" % (name, c)] = np.array(str(df[c].dtype))
    np.savez_compressed(os.path.join(HERE, "csv_frames.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "part7":
        part7()
        sys.exit(0)
    part1()
    part2()
    part3()
    part5()
    if os.path.isdir("/root/reference"):
        part4()
        part6()
        part7()
    else:
        print("note: /root/reference absent - iter_streams.npz not regenerated")
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
