"""Per-user fine-tuning loop (adaptive_test.py:104-116): 300 steps on a 10-rating batch, user tables only -
one tfr_train_steps_repeat call against 300 tfr_train_step calls.   python tools/probes/finetune_rate.py   (MI355X)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
import tfrecomm_amd as T
from tfrecomm_amd import adaptive_test as AT

for (U, I, D, opt, mode) in ((6040, 3952, 20, "sgd", "tf1"), (6040, 3952, 20, "adam", "tf1"), (6040, 3952, 20, "adam", "lazy"), (6040, 3952, 20, "adam", "tf1"),
                             (1_000_000, 100_000, 64, "adam", "lazy")):
    m = T.SvdModel(U, I, D, loss="nll", optimizer=opt, adam_mode=mode, lr=5e-3, reg=0.0)
    m.init_tables(seed=1)
    m.set_frozen(AT.FROZEN_BUT_USER)
    rs = np.random.RandomState(0)
    B, n = 10, 300
    u = np.full(B, 17, np.int32); i = rs.randint(0, I, B).astype(np.int32); r = (rs.rand(B) < 0.5).astype(np.float32)
    m.train_steps_repeat(u, i, r, n); m.train_step(u, i, r)
    t0 = time.perf_counter()
    for k in range(5):
        m.train_steps_repeat(u, i, r, n)
    t_rep = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for k in range(n):
        m.train_step(u, i, r)
    t_one = time.perf_counter() - t0
    print("%d x %d dim %d %s/%s: %d steps on a %d-rating batch: one call %.2f ms (%.1f us/step), %d calls %.2f ms (%.1f us/step)" % (
        U, I, D, opt, mode, n, B, t_rep * 1e3, t_rep * 1e6 / n, n, t_one * 1e3, t_one * 1e6 / n), flush=True)
    m.close()
