"""How long does the big-table step's k_apply_rows take when one item holds a fraction f of a 262144 batch?
   python tools/probes/hot_run.py        (MI355X)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
import tfrecomm_amd as T

U, I, D, B = 2_000_000, 200_000, 128, 262144
m = T.SvdModel(U, I, D, optimizer="adam", adam_mode="lazy", lr=1e-3, reg=0.05, device=0)
m.init_tables(seed=1)
rs = np.random.RandomState(3)
for f, nhot in ((0.0, 1), (0.001, 1), (0.01, 1), (0.05, 1), (0.1, 1), (0.3, 1), (0.1, 10), (0.3, 100), (0.5, 1000)):
    u = rs.randint(0, U, B).astype(np.int32)
    i = rs.randint(0, I, B).astype(np.int32)
    hot = rs.rand(B) < f
    i[hot] = rs.randint(0, nhot, hot.sum()).astype(np.int32) * 7 + 5
    r = rs.randint(1, 6, B).astype(np.float32)
    for k in range(2):
        m.train_step(u, i, r)
    m.profile(True)
    for k in range(5):
        m.train_step(u, i, r)
    p = m.profile_read()
    m.profile(False)
    print("hot fraction %.3f over %4d items (%6d entries, %5d pieces each):" % (f, nhot, hot.sum(), hot.sum() / nhot / 32),
          {k: round(v[0] * 1000 / max(1, v[1]), 1) for k, v in p.items() if v[1]}, flush=True)
