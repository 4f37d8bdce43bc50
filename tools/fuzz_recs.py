#!/usr/bin/env python3
"""One-off fuzz (GPU box): random sequences of calls around the drawn-ids path - drawn steps of odd lengths, host-staged ids,
a new rating store of another size, forwards, table reads, generator re-seeds - once with the store records left beside the drawn
ids (default) and once without (TFR_RECS=0): losses and tables must hash identically.
    python tools/fuzz_recs.py [n_seeds] [first_seed]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
seed = int(sys.argv[1])
rs = np.random.RandomState(seed)
U, I = int(rs.randint(500, 7000)), int(rs.randint(300, 5000))
D = int(rs.choice([16, 64, 128, 20]))
B = int(rs.choice([1000, 4096, 10000, 2500]))
h = hashlib.sha256()
def store(n):
    return rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), rs.randint(1, 6, n).astype(np.float32)
with T.SvdModel(U, I, D, optimizer="adam", adam_mode=str(rs.choice(["tf1", "lazy"])), lr=1e-3, reg=0.05) as m:
    m.init_tables(seed=seed)
    N = int(rs.randint(20000, 400000))
    m.upload_triples(*store(N))
    np.random.seed(seed)
    m.rng_from_numpy()
    for step in range(int(rs.randint(6, 14))):
        op = rs.randint(0, 7)
        if op <= 2:
            h.update(m.train_steps_drawn(B, int(rs.randint(1, 40)), want_loss=True).tobytes())
        elif op == 3:
            k = int(rs.randint(1, 6))
            m.stage_ids(rs.randint(0, N, k * B).astype(np.int64))
            h.update(np.asarray(m.train_steps_staged(0, B, k, want_loss=True)).tobytes())
        elif op == 4:
            N = int(rs.randint(20000, 400000))
            m.upload_triples(*store(N))
        elif op == 5:
            u, i, _ = store(1000)
            h.update(m.forward(u, i).tobytes())
        else:
            np.random.seed(int(rs.randint(1 << 30)))
            m.rng_from_numpy()
    for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
        h.update(np.ascontiguousarray(m.get_table(tid)).tobytes())
print("HASH", h.hexdigest())
"""


def run(seed, recs):
    env = dict(os.environ, TFR_RECS=recs)
    p = subprocess.run([sys.executable, "-c", WORKER % ROOT, str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if p.returncode != 0:
        return "ERR " + p.stderr.decode()[-600:]
    return [l for l in p.stdout.decode().splitlines() if l.startswith("HASH")][0]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for seed in range(first, first + n):
        a, b = run(seed, "1"), run(seed, "0")
        ok = a == b and a.startswith("HASH")
        bad += not ok
        print("%s seed %d %s" % ("ok " if ok else "BAD", seed, a[:40] if ok else (a, b)), flush=True)
    print("fuzz_recs done: %d seeds, %d bad" % (n, bad))
    sys.exit(1 if bad else 0)
