#!/usr/bin/env python3
"""One-off soak (GPU box): two identical long runs of the headline workload through the multi-step
look-ahead path must end in bit-identical tables (no float atomics, fixed summation orders), and a third
run cut into calls of odd lengths must match too.
    python tools/soak_determinism.py [steps] [big]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tfrecomm_amd as T
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
big = len(sys.argv) > 2 and sys.argv[2] == "big"     # radix-sort path with the look-ahead stream, lazy Adam
if big:
    U, I, D, B = 300000, 40000, 64, 70000            # > 65536: the wide-tile radix pass
    train, _ = bench.synth_uniform(U, I, 3000000)
else:
    U, I, D, B = 6040, 3952, 64, 10000
    train, _ = bench.synth_movielens(U, I, 1000209)
np.random.seed(13575)
chunk = 500 if big else 2000
digests = []
for run in range(3):
    rs = np.random.RandomState(99)
    m = T.SvdModel(U, I, D, optimizer="adam", adam_mode="lazy" if big else "tf1", lr=1e-3, reg=0.05)
    m.init_tables(seed=13575)
    m.upload_triples(*train)
    done = 0
    while done < steps:
        n = min(chunk, steps - done)
        ids = rs.randint(0, len(train[0]), (n, B))
        m.stage_ids(ids)
        if run < 2:
            m.train_steps_staged(0, B, n)
        else:                                             # same batches, odd call lengths
            k = 0
            for ln in (1, 2, 3, 7, 31, 257):
                while k + ln <= n and ln != 257:
                    m.train_steps_staged(k, B, ln); k += ln
                    break
            while k < n:
                ln = min(257, n - k)
                m.train_steps_staged(k, B, ln); k += ln
        done += n
    t = m.tables()
    h = hashlib.sha256(b"".join(np.ascontiguousarray(t[k]).tobytes() for k in sorted(t))).hexdigest()
    finite = all(np.isfinite(t[k]).all() for k in t)
    print("run %d: %d steps, sha256 %s, finite %s, step %d" % (run, steps, h[:16], finite, m.step), flush=True)
    digests.append(h)
    m.close()
ok = digests[0] == digests[1] == digests[2]
print("deterministic:", ok)
sys.exit(0 if ok else 1)
