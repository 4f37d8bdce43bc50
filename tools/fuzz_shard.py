#!/usr/bin/env python3
"""One-off fuzz (GPU box): the row-sharded step with the pipelined front end on 2 and 3 ranks sharing the box's GPU (gloo
stages the exchanges), random shapes incl. dims that take the unvectorised row layout - against the float64 oracle and bit for
bit against plain steps (tests/test_gpu_sharded.py::_pipelined_worker does the checking).
    python tools/fuzz_shard.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.multiprocessing as mp

from tests.test_gpu_sharded import _free_port, _pipelined_worker

def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    bad = 0
    for c in range(n_cases):
        world = int(rs.choice([2, 3]))
        D = int(rs.choice([1, 5, 6, 16, 20, 64, 100, 128]))
        U, I = int(rs.randint(world, 5000)), int(rs.randint(world, 3000))
        B = int(rs.choice([7, 64, 1000, 4097, 9000]))
        kw = [dict(optimizer="adam", adam_mode="lazy"), dict(optimizer="sgd", lr=1e-4), dict(optimizer="adam", adam_mode="tf1")][rs.randint(3)]
        tag = "case %d world=%d U=%d I=%d D=%d B=%d %s" % (c, world, U, I, D, B, kw)
        try:
            mp.spawn(_pipelined_worker, args=(world, _free_port(), kw, U, I, D, B, 4), nprocs=world, join=True)
            print("ok  ", tag, flush=True)
        except Exception as e:                                   # noqa: BLE001 - report and go on
            bad += 1
            print("BAD ", tag, "\n", str(e)[-1500:], flush=True)
    print("fuzz_shard done: %d cases, %d bad" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
