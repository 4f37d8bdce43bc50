"""Timeline of a short (20-step) drawn call against a 20-step pre-staged call (run under rocprofv3 --kernel-trace)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
import tfrecomm_amd as T
import bench

U, I, D, B = 6040, 3952, 64, 10000
train, val = bench.synth_movielens(U, I, 1_000_209)
m = T.SvdModel(U, I, D, optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05, device=0)
m.init_tables(seed=13575)
m.upload_triples(*train)
N = len(train[0])
m.rng_seed(13575)
for rep in range(3):
    m.train_steps_drawn(B, 5); m.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_steps_drawn(B, 20); m.sync(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("drawn  20 steps: %.1f us/step" % ((t1 - t0) * 1e6 / 20), flush=True)
    time.sleep(0.01)
rs = np.random.RandomState(1)
ids = rs.randint(0, N, (25, B))
m.stage_ids(ids)
for rep in range(3):
    m.train_steps_staged(0, B, 5); m.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_steps_staged(5, B, 20); m.sync(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("staged 20 steps: %.1f us/step" % ((t1 - t0) * 1e6 / 20), flush=True)
    time.sleep(0.01)
