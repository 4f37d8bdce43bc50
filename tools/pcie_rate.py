#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-fed path (tfr_train_step from pageable NumPy arrays, logits
copied back, synchronous) on the headline workload - the number DESIGN.md quotes next to the
HBM-resident `value` of bench.py.  Not part of the product."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tfrecomm_amd as T
import bench

U, I, D, B = 6040, 3952, 64, 10000
train, _ = bench.synth_movielens(U, I, 1000209)
np.random.seed(13575)
ids = np.random.randint(0, len(train[0]), (300, B))
m = T.SvdModel(U, I, D, optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05)
m.init_tables(seed=13575)
batches = [(train[0][s], train[1][s], train[2][s]) for s in ids]
for b in batches[:50]:
    m.train_step(*b)
t0 = time.perf_counter()
for b in batches[50:]:
    m.train_step(*b)
el = time.perf_counter() - t0
t1 = time.perf_counter()
for b in batches[50:]:
    m.train_step(*b, want_logits=False)
el2 = time.perf_counter() - t1
print(json.dumps(dict(workload="c2 host-fed (H2D ids+rates, D2H logits, sync per step)", us_per_step=el / 250 * 1e6,
                      ratings_per_s=250 * B / el, us_per_step_no_logits=el2 / 250 * 1e6, ratings_per_s_no_logits=250 * B / el2)))
