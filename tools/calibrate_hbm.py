#!/usr/bin/env python3
"""Same-box calibration of what the memory system delivers (not part of the product):
streaming copy, read-only reduction and a library row gather, next to the gather-dot forward."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(n)) * 1e-3, sum(ev[i].elapsed_time(ev[i + 1]) for i in range(n)) / n * 1e-3


out = {}
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()     # 1 GiB
y = torch.empty_like(x)
tmin, tavg = timeit(lambda: y.copy_(x))
out["copy_1GiB_read+write_GBps"] = [2 * x.numel() * 4 / tmin / 1e9, 2 * x.numel() * 4 / tavg / 1e9]
tmin, tavg = timeit(lambda: x.sum())
out["sum_1GiB_read_GBps"] = [x.numel() * 4 / tmin / 1e9, x.numel() * 4 / tavg / 1e9]
del y
U, D, B = 10_000_000, 128, 262144
P = torch.empty(U, D, dtype=torch.float32, device="cuda").normal_()
idx = torch.randint(0, U, (B,), device="cuda")
dst = torch.empty(B, D, dtype=torch.float32, device="cuda")
tmin, tavg = timeit(lambda: torch.index_select(P, 0, idx, out=dst))
out["index_select_512B_rows_read+write_GBps"] = [2 * B * D * 4 / tmin / 1e9, 2 * B * D * 4 / tavg / 1e9]
idx2 = torch.randint(0, U, (8 * B,), device="cuda")
dst2 = torch.empty(8 * B, D, dtype=torch.float32, device="cuda")
tmin, tavg = timeit(lambda: torch.index_select(P, 0, idx2, out=dst2))
out["index_select_512B_rows_8xB_read+write_GBps"] = [2 * 8 * B * D * 4 / tmin / 1e9, 2 * 8 * B * D * 4 / tavg / 1e9]
print(json.dumps(out))
