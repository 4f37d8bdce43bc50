#!/usr/bin/env python3
"""One-off soak (GPU box): the row-sharded step in a one-rank RCCL world - real asynchronous exchanges, the next batch's front end
really running on the side stream beside the step's kernels - with and without that pipelining: same table hashes after N steps.
(The two-rank tests stage their exchanges through gloo, which serialises the two streams.)
    python tools/soak_shard.py [steps]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from tfrecomm_amd import sharded, _lib as L
    U, I, D, B, N = 2_000_000, 200_000, 128, 131072, 4_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    su = torch.randint(0, U, (N,), dtype=torch.int32, device=dev, generator=g)
    si = torch.randint(0, I, (N,), dtype=torch.int32, device=dev, generator=g)
    sr = torch.randint(1, 6, (N,), device=dev, generator=g).to(torch.float32)
    ids = torch.randint(0, N, (steps + 1, B), dtype=torch.int64, device=dev, generator=g)
    digests = []
    for pipelined in (True, False, True):
        comm = sharded.Comm(side_group=dist.new_group())
        m = sharded.ShardedSvd(U, I, D, comm, lambda ur, ir, d: sharded.HipShard(ur, ir, d, 0, optimizer="adam", adam_mode="lazy"), device=dev)
        m.backend.model.init_tables(seed=3)
        torch.cuda.set_stream(m.backend.stream)
        m.backend.set_store(su, si, sr)
        for s in range(steps):
            m.train_step_local_ids(ids[s], ids[s + 1] if pipelined else None)
        m.backend.sync()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        t = m.local_tables()
        for k in (L.MU, L.BU, L.BI, L.P, L.Q):
            h.update(np.ascontiguousarray(t[k]).tobytes())
        digests.append(h.hexdigest()[:16])
        print("pipelined" if pipelined else "plain    ", steps, "steps, sha256", digests[-1], "finite", bool(np.isfinite(t[L.P]).all()), flush=True)
        m.backend.model.close()
    ok = len(set(digests)) == 1
    print("identical:", ok)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
