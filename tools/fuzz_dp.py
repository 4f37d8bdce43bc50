#!/usr/bin/env python3
"""One-off fuzz (GPU box): the data-parallel building blocks (tfr_dp_local_grads with the look-ahead
hint, all-reduce, tfr_dp_apply) in a one-rank gloo world against the NumPy oracle, random shapes.
    python tools/fuzz_dp.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist
from tfrecomm_amd import dataparallel, _lib as L
from tests.util import RTOL, dup_heavy_ids, make_oracle, rand_tables, rel_err

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for n in range(n_cases):
    D = int(rs.choice([4, 16, 20, 64, 128, 256]))
    U = int(rs.choice([5, 300, 6040, 16384, 20000]))
    I = int(rs.choice([7, 200, 3952, 16384, 17000]))
    B = int(rs.choice([1, 64, 1025, 5000, 10000, 12289, 16384, 20000]))
    opt = ["adam", "sgd"][rs.randint(2)]
    kw = dict(loss=["mse", "nll"][rs.randint(2)], item_abs=bool(rs.randint(2)), reg_bias=bool(rs.randint(2)),
              optimizer=opt, adam_mode="tf1", lr=3e-3, reg=0.02)
    t = rand_tables(rs, U, I, D, scale=0.3 / np.sqrt(max(D, 16) / 16))
    N, steps = 20000, 3
    su, si = dup_heavy_ids(rs, U, N), dup_heavy_ids(rs, I, N)
    sr = (rs.rand(N) < 0.5).astype(np.float32) if kw["loss"] == "nll" else rs.randint(1, 6, N).astype(np.float32)
    ids = rs.randint(0, N, (steps, B))
    tag = "case %d U=%d I=%d D=%d B=%d %s" % (n, U, I, D, B, kw)
    try:
        ref = make_oracle(U, I, D, t, **kw)
        be = dataparallel.HipReplica(U, I, D, 0, **kw)
        be.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        be.model.upload_triples(su, si, sr)
        be.model.stage_ids(ids)
        base, _ = be.model.staged_ids_devptr()
        dp = dataparallel.DataParallelSvd(be)
        lerr = 0.0
        for s in range(steps):
            scal = dp.train_step(store_ids_ptr=base + s * B * 8, batch=B, next_ids_ptr=base + (s + 1) * B * 8 if s + 1 < steps else None)
            _, wloss, _ = ref.train_step(su[ids[s]], si[ids[s]], sr[ids[s]])
            lerr = max(lerr, abs(float(scal[0]) - wloss) / max(abs(wloss), 1e-6))
        be.sync()
        got = be.model.tables()
        run = 7.0 * B / max(1, min(U, I))
        tol = (2e-4 if opt == "adam" else 4 * RTOL) * max(1.0, np.sqrt(run / 64)) * 3
        errs = [rel_err(got[x], ref.tables()[x]) for x in (L.MU, L.BU, L.BI, L.P, L.Q)]
        clean = float(be.flat.abs().max()) == 0.0
        if max(errs) > tol or lerr > 1e-4 or not clean:
            bad += 1
            print("MISMATCH", tag, ["%.2e" % e for e in errs], "tol %.1e" % tol, "loss err %.1e" % lerr, "flat clean:", clean, flush=True)
        be.model.close()
    except Exception as e:                                 # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(e), flush=True)
print("dp fuzz done: %d cases, %d bad" % (n_cases, bad))
dist.destroy_process_group()
sys.exit(1 if bad else 0)
