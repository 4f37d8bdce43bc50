"""Where a short call's time goes (the driver's `--steps 20 --warmup 5` form of the headline step): host timestamps around the
three statements of bench.py's timed region, repeated.  python tools/short_call.py [K] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from tfrecomm_amd.engine import SvdModel

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = bench.WORKLOADS["c2"]
U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
rs = np.random.RandomState(1)
n = 900000
m = SvdModel(U, I, D, device=0, optimizer="adam", adam_mode="tf1", lr=wl["lr"], reg=wl["reg"])
m.init_tables(seed=1)
m.upload_triples(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), rs.randint(1, 6, n).astype(np.float32))
np.random.seed(13575)
m.rng_from_numpy()
m.train_steps_drawn(B, 5); m.sync(); torch.cuda.synchronize()
rows = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_steps_drawn(B, K)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
    time.sleep(0.002)
a = np.array(rows)
med = np.median(a, axis=0)
print("TFR_SYNC_SPIN=%s K=%d  median us: enqueue %.1f  tfr_sync %.1f  torch.sync %.1f  total %.1f = %.2f us/step (min %.2f)" % (
    os.environ.get("TFR_SYNC_SPIN", "0"), K, med[0], med[1], med[2], med[3], med[3] / K, a[:, 3].min() / K))
