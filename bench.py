#!/usr/bin/env python3
"""bench.py - headline benchmark of the SVD minibatch training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one rank; without
them it starts its own N ranks as child processes BEFORE anything touches the GPU (no torch import, no library load in
the parent), relays rank 0's JSON line and exits with the children's return code.

A "step" = one pass of the reference's loop body (svd_train_val.py:66-72): next(iter_train) - the
np.random.randint id draw and the row gather - then forward, loss, backward, optimiser apply, on
synthetic ratings.  Default workload = BASELINE.json configs[1]: MovieLens-1M-shaped SVD, dim=64,
batch=10000, Adam (TF1 dense-moment semantics, i.e. exactly what tf.train.AdamOptimizer computes),
fp32.  The rating store is resident in HBM before the timed region; the id draw is INSIDE it (device
replica of NumPy's generator, same stream bit for bit).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  feeds               the same step with host-drawn ids (np.random.randint + upload inside the clock) and
                      with pre-staged ids, next to the device-drawn figure that is `value`
  roofline            dominant kernel of the timed workload, HIP-event timed on the model's stream
  val_rmse_converged  fixed 30-epoch leg (independent of --steps), next to the CPU oracle's on the same batches
  north_star_forward  the dim=128 gather-dot forward (BASELINE configs[2] shape), same fields
  cpu_baseline        oracle/svd_oracle.c (OpenMP restatement of the reference step) on all host cores
  c3_train_step       BASELINE configs[2] (10M x 1M rows, dim 128, batch 262144, lazy Adam): training step, per-kernel roofline
  c4_tables_one_gpu   BASELINE configs[3]'s tables (100M x 10M rows, 169 GB with Adam state) through the fused step on ONE GPU
  c5_fm_forward       BASELINE configs[4]: FM second-order forward, 1M features, dim 64
N>1 lines (data-parallel headline) add:
  sharded_c4          BASELINE configs[3] in its own form: tables row-sharded over the N GPUs, RCCL all-to-all row exchange
  single_gpu_reference  the N=1 step timed on rank 0 in the same run (what the N-GPU value scales from)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
PROFILE_ROUND = "r03"       # profiles/<round>_pmc_<workload>.csv: the committed counter summaries `traffic` is read from
STREAM_CEILING_GBS = 6290.0 # same guide: 6.29 TB/s measured for a float4 streaming copy (79 % of spec)

WORKLOADS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (plumbing)
    "c1": dict(name="MovieLens-1M-shaped SVD dim=15 batch=1000 Adam(tf1)", U=6040, I=3952, N=1000209,
               D=15, B=1000, adam_mode="tf1", lr=1e-3, reg=0.05),
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2": dict(name="MovieLens-1M-shaped SVD dim=64 batch=10000 Adam(tf1)", U=6040, I=3952, N=1000209,
               D=64, B=10000, adam_mode="tf1", lr=1e-3, reg=0.05),
    # BASELINE.json configs[2]: HBM-roofline run (1B-rating store scaled by --store-ratings)
    "c3": dict(name="synthetic 10M users x 1M items dim=128 batch=262144 Adam(lazy)", U=10_000_000,
               I=1_000_000, N=100_000_000, D=128, B=262144, adam_mode="lazy", lr=1e-3, reg=0.05),
    # BASELINE.json configs[3] tables (100M x 10M rows, 169 GB with Adam state) - on ONE GPU they still fit
    # its 288 GB; with --gpus N they are row-sharded
    "c4": dict(name="synthetic 100M users x 10M items dim=128 batch=262144 Adam(lazy)", U=100_000_000,
               I=10_000_000, N=100_000_000, D=128, B=262144, adam_mode="lazy", lr=1e-3, reg=0.05),
}


def synth_movielens(U, I, N, seed=13575):
    """ML-1M-shaped ratings from a rank-8 ground truth: r = clip(round(3.58 + b_u +
    b_i + <p,q> + eps), 1, 5); 90/10 train/val split."""
    rs = np.random.RandomState(seed)
    u = rs.randint(0, U, N).astype(np.int32)
    # popularity-skewed items, like real MovieLens
    w = 1.0 / np.arange(1, I + 1) ** 0.8
    i = rs.choice(I, N, p=w / w.sum()).astype(np.int32)
    pu, qi = rs.normal(0, 0.35, (U, 8)), rs.normal(0, 0.35, (I, 8))
    bu, bi = rs.normal(0, 0.35, U), rs.normal(0, 0.45, I)
    r = 3.58 + bu[u] + bi[i] + np.einsum("kd,kd->k", pu[u], qi[i]) + rs.normal(0, 0.85, N)
    r = np.clip(np.rint(r), 1, 5).astype(np.float32)
    cut = int(N * 0.9)
    perm = rs.permutation(N)
    tr, va = perm[:cut], perm[cut:]
    return (u[tr], i[tr], r[tr]), (u[va], i[va], r[va])


def synth_uniform(U, I, N, seed=13575, zipf=0.0):
    rs = np.random.RandomState(seed)
    u = rs.randint(0, U, N).astype(np.int32)
    if zipf > 0:        # popularity-skewed items (reported separately): rank k ~ k^-a, rank -> random row
        w = 1.0 / np.arange(1, I + 1, dtype=np.float64) ** zipf
        cdf = np.cumsum(w / w.sum())
        i = rs.permutation(I).astype(np.int32)[np.minimum(np.searchsorted(cdf, rs.random_sample(N)), I - 1)]
    else:
        i = rs.randint(0, I, N).astype(np.int32)
    r = rs.randint(1, 6, N).astype(np.float32)
    cut = N - min(N // 10, 1_000_000)
    return (u[:cut], i[:cut], r[:cut]), (u[cut:], i[cut:], r[cut:])


# ---- algorithmic bytes per launch: SURVEY 8(d)'s per-rating figures (every gathered row counted as if from HBM,
#      duplicates not discounted), split over the kernels of a step so that they add up to the step's figure ------
def algo_bytes(slot, B, D, U, I, adam_mode, small):
    if slot == "forward":                   # 2 rows + 2 biases + 2 ids + rating (+ g out): 8D+24
        return B * (8 * D + 24)
    if small:                               # tile path (ML-1M): k_tile_step = forward + gradient + tile-sort scratch
        if slot == "reduce_item":
            return B * (8 * D + 24 + 32)
        if slot == "apply":                 # TF1 Adam: w, m, v of EVERY row read and written, 6*4*(U+I)*(D+1) per step;
            dense = 24 * (U + I) * (D + 1)  # lazy / SGD touch <= 2B rows
            return dense if adam_mode == "tf1" else min(dense, B * 2 * 24 * (D + 1))
        return 0
    if adam_mode == "lazy":                 # 8(d): 56D+104 per rating = forward 8D+24, 2 x (24D+24) row updates, 32 sort scratch
        return {"reduce_item": B * (32 * D + 64), "reduce_user": B * (24 * D + 40), "sort": 0, "gather": 0,
                "apply": 0, "finalize": 0}.get(slot, 0)
    return {"reduce_item": B * 32, "apply": 24 * (U + I) * (D + 1)}.get(slot, 0)


def lane_group(D):
    """lanes a table row is spread over (csrc/svd_kernels.h geometry())"""
    lanes, g = (D // 4 if D % 4 == 0 else D), 4
    while g < lanes:
        g <<= 1
    return g


def profiled_traffic(fname, kernel_substr):
    """HBM-side bytes per launch from a committed rocprofv3 PMC summary under profiles/: TCC_EA0_RDREQ x 128 B
    (= FETCH_SIZE x 2 KiB, the gfx950 correction for 16-byte-per-lane reads, MI355X_MICROARCH.md "HBM") +
    WRITE_SIZE (KiB).  bench.py cannot collect counters itself; None when the summary or the kernel is absent."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    rd = wr = None
    for line in open(path):
        f = line.rstrip("\n").rsplit(",", 5)           # kernel names hold commas: split from the right
        if len(f) == 6 and f[0].strip('"').endswith(kernel_substr):
            if f[1] == "TCC_EA0_RDREQ_sum":
                rd = float(f[3]) * 128.0
            if f[1] == "WRITE_SIZE":
                wr = float(f[3]) * 1024.0
    return None if rd is None else rd + (wr or 0.0)


def host_cpus():
    """(cpus this process may run on, cgroup CPU quota or None): a GPU box exposes every host cpu (256) but gives the job a share"""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for qf, pf in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if pf is None:
                q, per = open(qf).read().split()[:2]
            else:
                q, per = open(qf).read().strip(), open(pf).read().strip()
            if q not in ("max", "-1"):
                quota = max(1, int(math.ceil(float(q) / float(per))))
                break
        except (OSError, ValueError):
            pass
    return ncpu, quota


def cpu_team_size():
    """OpenMP team for a CPU baseline: the cgroup quota when there is one, else at most 16 of the visible cpus"""
    ncpu, quota = host_cpus()
    return min(quota, ncpu) if quota else min(16, ncpu)


def cpu_convergence(wl, train, val, tabs0, steps):
    """oracle/svd_oracle.c (OpenMP, all host cores) from the same initial tables over the same id stream
    (np.random.seed(13575); one randint(0, N, (B,)) per step - dataio.py:115): its val RMSE, and the time of
    the training steps alone (the host-side gather of each batch is outside the clock)."""
    from oracle import c_oracle
    U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
    orc = c_oracle.COracle(U, I, D, adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"])
    tu, ti, tr = train
    # how many threads: a GPU box exposes every host cpu (256) but gives this job a share of them, and an OpenMP
    # team larger than the share crawls.  Take the cgroup quota if there is one; otherwise time each candidate
    # team size (after ~1 s of warm-up: the first second after a team-size change is far slower) and keep the fastest.
    ncpu, quota = host_cpus()
    cands = [min(quota, ncpu)] if quota else sorted({c for c in (4, 8, 16, 32, 64) if c <= ncpu} or {ncpu})
    rs = np.random.RandomState(0)
    best, tried = None, {}
    for c in cands:
        c_oracle.set_threads(c)
        orc.set_tables(*tabs0)
        t, t_start = [], time.perf_counter()
        while True:
            sel = rs.randint(0, len(tu), (B,))
            bu, bi, br = tu[sel], ti[sel], tr[sel]
            t0 = time.perf_counter()
            orc.train_step(bu, bi, br, want_logits=False)
            t.append(time.perf_counter() - t0)
            if len(cands) == 1 or (len(t) >= 6 and time.perf_counter() - t_start > 1.5) or len(t) >= 400:
                break
        tried[c] = min(t[-3:])
        if best is None or tried[c] < tried[best]:
            best = c
    c_oracle.set_threads(best)
    orc.close()
    orc = c_oracle.COracle(U, I, D, adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"])   # fresh optimiser state
    orc.set_tables(*tabs0)
    np.random.seed(13575)                                       # svd_train_val.py:15
    cpu_s, done = 0.0, 0
    for s in range(steps):
        sel = np.random.randint(0, len(tu), (B,))
        bu, bi, br = tu[sel], ti[sel], tr[sel]
        t0 = time.perf_counter()
        orc.train_step(bu, bi, br, want_logits=False)
        cpu_s += time.perf_counter() - t0
        done += 1
    vu, vi, vr = val
    rmse = float(np.sqrt(np.mean((orc.forward(vu, vi).astype(np.float64) - vr) ** 2)))
    orc.close()
    return rmse, dict(value=done * B / cpu_s, unit="ratings/s", cores=best, kind="port",
                      sample="%d steps of the same workload (same initial tables, same id stream) in %.1f s of step time; "
                             "oracle/svd_oracle.c, OpenMP restatement of the svd_train_val.py step (TensorFlow unavailable); "
                             "%d threads = the fastest team size on this host (ms per step by team size: %s; host exposes %d cpus)"
                             % (done, cpu_s, best, ", ".join("%d: %.2f" % (c, v * 1e3) for c, v in sorted(tried.items())), ncpu))


def north_star_forward(device, steps=100, warmup=10, U=10_000_000, I=1_000_000, D=128, B=262144, sequential=False, zipf=0.0):
    """BASELINE configs[2] shape, forward only: achieved algorithmic GB/s of the gather-dot."""
    import tfrecomm_amd as T
    import torch
    m = T.SvdModel(U, I, D, optimizer="sgd", device=device)
    m.init_tables(seed=7)
    g = torch.Generator(device="cuda")
    g.manual_seed(13575)
    nb = 8 if B <= 262144 else 2
    du = torch.randint(0, U, (nb, B), dtype=torch.int32, device="cuda", generator=g)
    di = torch.randint(0, I, (nb, B), dtype=torch.int32, device="cuda", generator=g)
    if zipf > 0:        # popularity-skewed items: rank k drawn with probability ~ k^-a (host, seeded), rank -> random row
        rs = np.random.RandomState(13575)
        w = 1.0 / np.arange(1, I + 1, dtype=np.float64) ** zipf
        cdf = np.cumsum(w / w.sum())
        ranks = np.searchsorted(cdf, rs.rand(nb * B)).clip(0, I - 1)
        perm = rs.permutation(I)
        di = torch.from_numpy(perm[ranks].astype(np.int32).reshape(nb, B)).to("cuda")
    if sequential:      # diagnostic: perfectly streaming rows
        du = (torch.arange(nb * B, device="cuda", dtype=torch.int64) % U).to(torch.int32).reshape(nb, B).contiguous()
        di = (torch.arange(nb * B, device="cuda", dtype=torch.int64) % I).to(torch.int32).reshape(nb, B).contiguous()
    out = torch.empty(B, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for s in range(warmup):
        m.forward_dev(du[s % nb].data_ptr(), di[s % nb].data_ptr(), B, out.data_ptr())
    m.sync()
    m.profile(True)
    t0 = time.perf_counter()
    for s in range(steps):
        m.forward_dev(du[s % nb].data_ptr(), di[s % nb].data_ptr(), B, out.data_ptr())
    m.sync()
    wall = time.perf_counter() - t0
    ms, n = m.profile_read()["forward"]
    m.profile(False)
    m.close()
    per_launch = B * (8 * D + 20)
    gbs = per_launch / (ms / n * 1e-3) / 1e9
    chk = float(out.double().sum().item())
    # same-device yardstick: the library's own float4 copy kernel (one element per thread, streaming loads and stores) over 1 GiB (read + write bytes); the guide's
    # figure for this kernel shape is 6.29 TB/s (STREAM_CEILING_GBS)
    copy_gbs, copy_mean = T.device_copy_rate(device, 1 << 30, 10)
    G = lane_group(D)
    pnt = U * D * 4 > (256 << 20)
    kern = "k_forward<%d, %d, 0, 4, %s>" % (G, 4 if D % 4 == 0 else 1, "true" if pnt else "false")
    tag = "8x_batch" if B == 8 * 262144 else "zipf" if zipf > 0 else "uniform"
    std = (U, I, D) == (10_000_000, 1_000_000, 128) and (B, zipf) in ((262144, 0.0), (262144, 1.05), (8 * 262144, 0.0)) and not sequential
    fname = PROFILE_ROUND + "_pmc_forward_%s.csv" % tag
    return dict(checksum=chk, kernel=kern,
                workload="%d users x %d items, dim=%d, batch=%d, %s ids" % (U, I, D, B, "Zipf(%.2f) item" % zipf if zipf > 0 else
                                                                          ("sequential" if sequential else "uniform")),
                bound="hbm (cache-assisted: hot item rows are L2 / Infinity-Cache hits, counter traffic is below the algorithmic bytes)" if zipf > 0 else "hbm",
                achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                traffic=profiled_traffic(fname, kern) if std else None,
                traffic_source="profiles/%s (rocprofv3 --pmc, separate passes over bench.py --only-north-star)" % fname,
                algorithmic_bytes_per_launch=per_launch, avg_launch_us=ms / n * 1e3,
                timing="HIP events around each launch on the model's stream, %d launches" % n,
                ratings_per_s=B * n / (ms * 1e-3), wall_ratings_per_s=B * steps / wall,
                stream_ceiling_GBps=STREAM_CEILING_GBS, frac_of_stream_ceiling=gbs / STREAM_CEILING_GBS,
                device_copy_GBps=copy_gbs, device_copy_mean_GBps=copy_mean, frac_of_device_copy=gbs / copy_gbs,
                device_copy_note="tfr_device_copy_rate: float4 copy kernel of this library (one 16-byte element per thread, streaming loads and stores: the fastest form on this part, tools/probes/copy_bw.hip), 1 GiB, best of 10 launches")


def fm_forward_bench(device, steps=50, warmup=5, F=1_000_000, D=64, n=1 << 20, nnz=8, no_cpu=False):
    """BASELINE configs[4]: FM pairwise-interaction forward, 1M sparse features, dim=64; synthetic rows of
    nnz=8 (1 "user" in [0,4e5), 1 "item" in [4e5,6e5), 6 count-valued features elsewhere)."""
    import tfrecomm_amd as T
    import torch
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev)
    g.manual_seed(13575)
    cols = [torch.randint(0, 400_000, (n, 1), device=dev, generator=g),
            torch.randint(400_000, 600_000, (n, 1), device=dev, generator=g),
            torch.randint(600_000, F, (n, nnz - 2), device=dev, generator=g)]
    indices = torch.cat(cols, 1).to(torch.int32).contiguous()
    data = torch.cat([torch.ones(n, 2, device=dev), torch.randint(1, 4, (n, nnz - 2), device=dev, generator=g).float()], 1).contiguous()
    indptr = (torch.arange(n + 1, device=dev, dtype=torch.int64) * nnz).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    m = T.FmModel(F, D, device=device, loss="nll", optimizer="sgd", lr=1e-3, reg=1e-4)
    m.init(seed=3, stddev=0.1)
    ms = []
    for s in range(warmup + steps):
        m.forward_dev(indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), n, out.data_ptr())
        t = m.sync()
        if s >= warmup:
            ms.append(t)
    # CPU baseline: the reference's own formula (forward.py:21-22) restated in C with OpenMP over rows (oracle/svd_oracle.c
    # fmo_forward, float64 accumulation like the reference's NumPy) on ALL the rows of the batch, the box's core share
    cpu = None
    if not no_cpu:
        from oracle import c_oracle
        ncores = cpu_team_size()
        c_oracle.set_threads(ncores)
        mu, Wv, Vv = m.get()
        h_ind, h_dat = indices.reshape(-1).cpu().numpy(), data.reshape(-1).cpu().numpy()
        h_ptr = np.arange(n + 1, dtype=np.int64) * nnz
        c_oracle.fm_forward(mu, Wv, Vv, h_ptr[: 4097], h_ind, h_dat)           # thread team warm-up
        t0 = time.perf_counter()
        y_cpu = c_oracle.fm_forward(mu, Wv, Vv, h_ptr, h_ind, h_dat)
        cpu_s = time.perf_counter() - t0
        got = out.double().cpu().numpy()
        err = float(np.abs(y_cpu - got).max() / max(1e-30, np.abs(y_cpu).max()))
        cpu = dict(value=n / cpu_s, unit="rows/s", cores=ncores, kind="port",
                   sample="all %d rows of the batch in %.2f s: forward.py:21-22's formula (general x^2 form), float64 accumulation, C + OpenMP over "
                          "rows (oracle/svd_oracle.c fmo_forward), %d threads; max scale-relative difference to the GPU's float32 output %.1e"
                          % (n, cpu_s, ncores, err))
    # training step on the same rows (SGD): forward+coefficients, radix sort of the non-zeros, segmented reduce
    yt = (torch.rand(n, device=dev, generator=g) < 0.5).float()
    tms = []
    for s in range(3 + min(steps, 20)):
        m.train_step_dev(indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), yt.data_ptr(), n, n * nnz)
        t = m.sync()
        if s >= 3:
            tms.append(t)
    train_ms = sum(tms) / len(tms)
    m.close()
    avg = sum(ms) / len(ms)
    per_launch = n * (nnz * (4 * D + 12) + 4)
    gbs = per_launch / (avg * 1e-3) / 1e9
    G = lane_group(D)
    kern = "k_fm_forward<%d, %d, false, %s>" % (G, 4 if D % 4 == 0 else 1, "true" if F * D * 4 >= 128 << 20 else "false")   # V streamed past the Infinity Cache
    # training step bytes per non-zero (SGD): forward V row + W + index + value (4D+12); backward: the row's factor sum s_r
    # (4D, gathered per non-zero), own V row read + written (8D), W read + written (8), coefficients / keys / positions of the
    # 3-pass radix sort (~16 B x 3 x 2) -> 16D + ~120
    train_bytes = n * nnz * (16 * D + 120)
    return dict(metric="FM second-order forward rows/sec (1M features, dim=64, nnz=8)", value=n / (avg * 1e-3), unit="rows/s",
                n_gpus=1, steps=steps, warmup=warmup, ms_per_step=avg, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f32", data="synthetic", config=dict(workload="c5: FM forward F=1M D=64 rows=2^20 nnz=8"),
                roofline=dict(kernel=kern, bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=gbs / HBM_PEAK_GBS, traffic=profiled_traffic(PROFILE_ROUND + "_pmc_c5.csv", kern),
                              traffic_source="profiles/%s_pmc_c5.csv (rocprofv3 --pmc, separate passes)" % PROFILE_ROUND,
                              algorithmic_bytes_per_launch=per_launch, avg_launch_us=avg * 1e3,
                              timing="HIP events around the launch (tfr_fm_sync), %d launches" % len(ms),
                              note="the 256 MB factor table is largely Infinity-Cache resident at this size"),
                cpu_baseline=cpu, checksum=float(out.double().sum().item()),
                train=dict(note="one SGD minibatch on the same 2^20 rows (8.4M non-zeros), whole step incl. the radix sort of the non-zeros; "
                                "own definition (the reference trains this model in the external libFM binary by MCMC)",
                           ms_per_step=train_ms, rows_per_s=n / (train_ms * 1e-3), algorithmic_bytes_per_step=train_bytes,
                           achieved_GBps=train_bytes / (train_ms * 1e-3) / 1e9, frac=train_bytes / (train_ms * 1e-3) / 1e9 / HBM_PEAK_GBS))


def als_roofline(n, U, W, d, ms_per_iter):
    """One ALS iteration = a user half-sweep and a work half-sweep (als3.py:67-108).  Per rating and half-sweep: the partner's
    factor row (8d bytes, float64), the rating (8) and two ids (16) are read, and the d x d normal equations take 2 d^2 flops for
    A += v v^T plus 2d for b += y v; per entity a Cholesky solve of ~d^3/3 + 2 d^2 flops and a d-row written."""
    flops = 2 * n * (2 * d * d + 2 * d) + (U + W) * (d ** 3 / 3.0 + 2 * d * d)
    nbytes = 2 * n * (8 * d + 24) + (U + W) * (8 * d + 8)
    sec = ms_per_iter * 1e-3
    return dict(kernel="k_als_fit / k_als_partial", bound="latency", achieved=flops / sec / 1e12, peak=78.6, unit="TFLOP/s",
                frac=flops / sec / 1e12 / 78.6, traffic=None, algorithmic_flops_per_iteration=flops, algorithmic_bytes_per_iteration=nbytes,
                achieved_GBps=nbytes / sec / 1e9, frac_of_hbm=nbytes / sec / 1e9 / HBM_PEAK_GBS,
                note="float64 like the reference; one 256-thread block per user / work builds its d x d normal equations from LDS tiles and "
                     "Cholesky-solves them: %.2f GFLOP and %.0f MB per iteration - far below both the f64 vector peak (78.6 TFLOP/s) and HBM, "
                     "the sweep is bound by per-entity latency (gather -> accumulate -> factorise chain) at this problem size"
                     % (flops / 1e9, nbytes / 1e6))


def als_bench(device, iters=5):
    """als3.py on MovieLens-1M-shaped ratings (6040 x 3952, 1M ratings, nb_components=20): seconds per
    ALS iteration (every user, then every work) on the GPU, next to the NumPy port on this host."""
    import tfrecomm_amd as T
    from oracle.als_oracle import AlsOracle
    U, W, d, lam = 6040, 3952, 20, 0.1
    (tu, ti, tr), (vu, vi, vr) = synth_movielens(U, W, 1000209)
    X, y = np.stack([tu, ti], 1).astype(np.int64), tr.astype(np.float64)
    Xt, yt = np.stack([vu, vi], 1).astype(np.int64), vr.astype(np.float64)
    als = T.MangakiALS3(nb_components=d, nb_iterations=iters, lambda_=lam, device=device, verbose=False)
    als.nb_users, als.nb_works = U, W
    np.random.seed(13575)
    t0 = time.perf_counter()
    als.fit(X, y, yt, Xt)
    wall = time.perf_counter() - t0
    rmse = als.compute_rmse(yt, als.predict(Xt))
    ms_per_iter = als.sweep_ms / iters
    als.close()
    o = AlsOracle(U, W, d, 1, lam)
    np.random.seed(13575)
    o.init_vars()
    o.load(X, y)
    t0 = time.perf_counter()
    o.sweep()
    cpu_s = time.perf_counter() - t0
    n = len(y)
    return dict(metric="ALS (als3.py) ratings/sec per iteration, MovieLens-1M-shaped, nb_components=20", value=2 * n / (ms_per_iter * 1e-3),
                unit="ratings/s", n_gpus=1, steps=iters, warmup=0, ms_per_step=ms_per_iter, higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f64", data="synthetic", config=dict(workload="als: 6040 x 3952, 900188 ratings, d=20, lambda=0.1"),
                val_rmse=rmse, fit_wall_s=wall,
                roofline=als_roofline(n, U, W, d, ms_per_iter),
                cpu_baseline=dict(value=2 * n / cpu_s, unit="ratings/s", cores=1, kind="port",
                                  sample="1 iteration of oracle/als_oracle.py (NumPy restatement of als3.py) in %.1f s" % cpu_s))


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of THIS file (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), nothing GPU-related imported here.  Rank 0's stdout carries the one
    JSON line and is relayed; the other ranks' stdout goes to stderr.  Returns the largest child return code; when a
    rank dies the others (which would wait for it in a collective) are terminated - by their own pids."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    print("bench.py: started %d ranks (pids %s), rendezvous 127.0.0.1:%d" % (n, [q.pid for q in procs], port), file=sys.stderr, flush=True)
    failed_at = None
    while any(q.poll() is None for q in procs[1:]):
        if failed_at is None and any(q.poll() not in (None, 0) for q in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 15:
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            break
        if procs[0].poll() is not None and failed_at is None and all(q.poll() is not None for q in procs):
            break
        time.sleep(0.2)
    out0 = procs[0].communicate()[0].decode("utf-8", "replace") if procs[0].stdout else ""
    for q in procs:
        try:
            q.wait(timeout=30)
        except subprocess.TimeoutExpired:
            q.kill()
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    for l in out0.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return max(abs(q.returncode or 0) for q in procs)


def device_store(U, I, N, dev, seed=13575):
    """uniform synthetic (user, item, rating) triples drawn on the device (int32, int32, float32 tensors)"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    su = torch.randint(0, U, (N,), dtype=torch.int32, device=dev, generator=g)
    si = torch.randint(0, I, (N,), dtype=torch.int32, device=dev, generator=g)
    sr = torch.randint(1, 6, (N,), device=dev, generator=g).to(torch.float32)
    return su, si, sr


def svd_single_gpu(key, wl, K, W, device, *, zipf=0.0, feeds=True, cpu=True, convergence=True, cpu_steps=None):
    """One SVD workload on one GPU: the timed region of the bench contract (W warm-up + exactly K steps of the whole
    loop body, device-drawn ids), the other feeds, per-kernel HIP-event timing with the roofline of every kernel, and the
    CPU baseline.  Returns the JSON object (the line itself for the headline workload, a nested object for the others)."""
    import torch
    import tfrecomm_amd as T
    U, I, D, B = wl["U"], wl["I"], wl["D"], wl["B"]
    dev = torch.device("cuda", device)
    small = B <= 16384 and max(U, I) <= 16384                    # the tile path (k_tile_step + k_dense_tiles)
    on_device = key in ("c3", "c4") and zipf == 0
    if on_device:       # big synthetic stores are drawn on the device (seconds of host randint + upload otherwise)
        su, si, sr = device_store(U, I, wl["N"], dev)
        cut = wl["N"] - min(wl["N"] // 10, 1_000_000)
        ntrain = cut
        val = tuple(x[cut:].cpu().numpy() for x in (su, si, sr))
        train = None
    else:
        gen = synth_movielens if key in ("c1", "c2") else synth_uniform
        train, val = gen(U, I, wl["N"], zipf=zipf) if zipf > 0 and gen is synth_uniform else gen(U, I, wl["N"])
        ntrain = len(train[0])

    def new_model():
        mm = T.SvdModel(U, I, D, optimizer="adam", adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"], device=device)
        mm.init_tables(seed=13575)
        if on_device:
            torch.cuda.synchronize()
            mm.set_triples_dev(su.data_ptr(), si.data_ptr(), sr.data_ptr(), ntrain)
        else:
            mm.upload_triples(*train)
        mm.upload_eval_triples(*val)
        return mm

    m = new_model()
    plan = m.kernel_plan(B)

    # ---- timed region: the whole reference step - next(iter_train) + sess.run(train_op) (svd_train_val.py:67-72) -
    #      inside the clock: the id draw (device replica of NumPy's generator, bit-identical stream), the gather from
    #      the HBM-resident rating store, forward, backward, optimiser.  W warm-up steps, then exactly K steps.
    np.random.seed(13575)
    m.rng_from_numpy()
    m.train_steps_drawn(B, W)
    m.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_steps_drawn(B, K)
    m.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / K * 1e3
    value = K * B / elapsed
    sse, _, nval = m.eval_resident()
    val_rmse_timed = math.sqrt(sse / nval)

    feeds_obj = None
    if feeds:
        # ---- the same step under the other two feeds, named (same model, training continues) -------------------
        m.rng_to_numpy()
        K2 = min(K, 300)
        ids = np.random.randint(0, ntrain, (K2, B))              # (a) ids drawn and uploaded BEFORE the clock starts
        m.stage_ids(ids)
        m.sync()
        t0 = time.perf_counter()
        m.train_steps_staged(0, B, K2)
        m.sync()
        staged_rate = K2 * B / (time.perf_counter() - t0)
        t0 = time.perf_counter()                                 # (b) the reference's own host draw inside the clock:
        for s in range(K2):                                      #     np.random.randint per step + an 8*B-byte upload
            m.train_step_ids(np.random.randint(0, ntrain, (B,))) #     (pinned ring, no host sync between steps)
        m.sync()
        host_rate = K2 * B / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for s in range(K2):
            np.random.randint(0, ntrain, (B,))
        host_draw_us = (time.perf_counter() - t0) / K2 * 1e6
        feeds_obj = dict(device_drawn_ids=dict(ratings_per_s=value, note="= value: MT19937 + masked rejection on the device (rng.hip), on a side stream ahead of the steps"),
                         host_drawn_ids=dict(ratings_per_s=host_rate, note="np.random.randint(0, N, (B,)) on the host per step (%.0f us each on this host) + "
                                             "async id upload; the host draw is the bound" % host_draw_us),
                         prestaged_ids=dict(ratings_per_s=staged_rate, note="ids drawn and uploaded before the clock (round-1 headline definition)"))

    # ---- per-kernel HIP-event timing (measured intervals, calibrated empty-pair overhead subtracted) -------------
    kp = min(K, 300)
    np.random.seed(13575)
    m.rng_from_numpy()
    m.profile(True)
    m.train_steps_drawn(B, kp)
    prof = m.profile_read()
    m.profile(False)
    pmc_file = "%s_pmc_%s.csv" % (PROFILE_ROUND, key)
    kern = {}
    for slot, (tot_ms, n) in prof.items():
        if n:
            ab = algo_bytes(slot, B, D, U, I, wl["adam_mode"], small)
            per_step_us = tot_ms / kp * 1e3
            name = plan.get(slot, {"draw": "k_mt_draw, or k_mt_blocks + k_mt_count + k_mt_emit from 32768 ids per draw (side stream)"}.get(slot, slot))
            gb = ab / (per_step_us * 1e-6) / 1e9 if ab and per_step_us > 0 else None
            kern[slot] = dict(kernel=name, launches_per_step=n / kp, us_per_step=per_step_us, algorithmic_bytes_per_step=ab,
                              achieved_GBps=gb, frac=(gb / HBM_PEAK_GBS if gb else None),
                              traffic=profiled_traffic(pmc_file, name) if slot in plan else None)
    on_path = {k: v for k, v in kern.items() if k != "draw"}
    dom = max(on_path, key=lambda k: on_path[k]["us_per_step"])
    d = on_path[dom]
    gbs = d["achieved_GBps"] or 0.0
    step_bytes = sum(v["algorithmic_bytes_per_step"] or 0 for v in on_path.values())
    roofline = dict(kernel=d["kernel"], slot=dom,
                    bound="latency" if small else "hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                    traffic=d["traffic"], traffic_source="profiles/%s (rocprofv3 --pmc, separate passes)" % pmc_file,
                    algorithmic_bytes_per_launch=d["algorithmic_bytes_per_step"] / max(d["launches_per_step"], 1e-9),
                    avg_launch_us=d["us_per_step"] / max(d["launches_per_step"], 1e-9),
                    whole_step=dict(algorithmic_bytes=step_bytes, achieved_GBps=step_bytes / (ms_per_step * 1e-3) / 1e9,
                                    frac=step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS),
                    timing="HIP events on the model's stream around every launch of %d steps; measured intervals, not normalised "
                           "(their sum is below ms_per_step: launch gaps and the un-amortised ramp of a short timed region are not kernel time)" % kp,
                    kernels=kern,
                    note=("the two dependent launches of this step work on 2.6 MB of tables + Adam state that stay in L2 / Infinity Cache; "
                          "each is a chain of 2-3 dependent memory round trips, so the step is latency-bound and its HBM fraction is small by "
                          "construction - the HBM-bound measurement of this repo is north_star_forward") if small else
                         ("algorithmic bytes per SURVEY 8(d): 56D+104 per rating for the lazy-Adam step, split item side (forward inside) 32D+64 / "
                          "user side 24D+40; the pre-update item-row copy and the partner re-read are implementation traffic, not counted"))
    tabs_now = None
    if cpu and not (key in ("c1", "c2") and convergence) and key != "c4":
        t5 = m.tables()                                          # any valid tables do for a throughput baseline
        tabs_now = tuple(np.array(t5[k]) for k in (T._lib.MU, T._lib.BU, T._lib.BI, T._lib.P, T._lib.Q))
    m.close()

    metric = "training ratings/sec + val RMSE, MovieLens-1M SVD dim=64 @1 GPU" if key == "c2" \
        else "training ratings/sec, %s" % wl["name"]
    out = dict(metric=metric, value=value,
               unit="ratings/s", n_gpus=1, steps=K, warmup=W, ms_per_step=ms_per_step, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload="%s: %s" % (key, wl["name"]) + (" [item ids ~ Zipf(%.2f)]" % zipf if zipf > 0 else ""), users=U, items=I, dim=D, batch=B, train_ratings=ntrain,
                           optimizer="adam", adam_mode=wl["adam_mode"], lr=wl["lr"], reg=wl["reg"],
                           id_stream="np.random.seed(13575); randint(0, N, (B,)) per step, drawn inside the timed loop",
                           parallelism="single GPU"),
               val_rmse_after_timed_steps=val_rmse_timed, roofline=roofline)
    if feeds_obj:
        out["feeds"] = feeds_obj

    # ---- fixed-length convergence leg, independent of --steps (the "+ val RMSE" half of the metric): 30 epochs of 90
    #      steps from the same initial tables on the GPU and on the CPU oracle, same id stream -----------------------
    if key in ("c1", "c2") and convergence:
        conv_steps = 30 * (ntrain // B if ntrain // B < 90 else 90)
        m2 = new_model()
        t5 = m2.tables()
        tabs0 = tuple(np.array(t5[k]) for k in (T._lib.MU, T._lib.BU, T._lib.BI, T._lib.P, T._lib.Q))
        np.random.seed(13575)
        m2.rng_from_numpy()
        t0 = time.perf_counter()
        m2.train_steps_drawn(B, conv_steps)
        m2.sync()
        gpu_s = time.perf_counter() - t0
        sse, _, nval = m2.eval_resident()
        m2.close()
        out["val_rmse_converged"] = math.sqrt(sse / nval)
        out["convergence"] = dict(steps=conv_steps, gpu_seconds=gpu_s, gpu_val_rmse=out["val_rmse_converged"],
                                  note="noise floor of the synthetic ratings ~0.9 (sigma 0.85 + rounding); README.md:47-57 reports ~0.91 on real ML-1M")
        if cpu:
            cpu_rmse, cpu_obj = cpu_convergence(wl, train, val, tabs0, conv_steps)
            out["convergence"]["cpu_oracle_val_rmse"] = cpu_rmse
            out["convergence"]["rel_diff"] = abs(cpu_rmse - out["val_rmse_converged"]) / cpu_rmse
            out["cpu_baseline"] = cpu_obj
    elif cpu and key == "c4":
        out["cpu_baseline"] = None       # filled by the caller from c3's figure (same per-step work: 2 x B touched rows)
    elif cpu:
        if on_device:
            train = tuple(x[:ntrain].cpu().numpy() for x in (su, si, sr))
        _, out["cpu_baseline"] = cpu_convergence(wl, train, val, tabs_now, cpu_steps or (3 if B >= 100000 else 200))
    return out


def multi_gpu(args, rank, local_rank, world, force_dp, saved_stdout):
    """One rank of the N>1 line.  Headline workload (c2: tables every GPU can hold) -> data parallel; tables at the scale
    sharding is meant for (c3 / c4) -> row-sharded.  The default line carries BOTH: the data-parallel headline and, as
    `sharded_c4`, BASELINE configs[3] row-sharded over the same GPUs (the north star's 1 -> 8 curve), plus the single-GPU
    step and the CPU baseline measured on rank 0 in this run."""
    import torch
    import torch.distributed as dist
    wl = dict(WORKLOADS[args.workload])
    if args.adam_mode:
        wl["adam_mode"] = args.adam_mode
    if args.store_ratings:
        wl["N"] = args.store_ratings
    U, I, D = wl["U"], wl["I"], wl["D"]
    K, W = args.steps, args.warmup
    small = (U + I) * (D + 1) * 4 <= 256 << 20 and wl["adam_mode"] == "tf1"
    from tfrecomm_amd import sharded
    extra = {}
    if small:
        from tfrecomm_amd import dataparallel
        gen = synth_movielens if args.workload in ("c1", "c2") else synth_uniform
        train, val = gen(U, I, wl["N"])
        res = dataparallel.bench_entry(wl, K, W, rank, local_rank, world, train, val, workload_key=args.workload)
        if not args.no_sharded_c4:
            c4 = dict(WORKLOADS["c4"])
            if args.c4_scale > 1:                              # rehearsals on one shared GPU: smaller tables, same code path
                c4["U"], c4["I"] = c4["U"] // args.c4_scale, c4["I"] // args.c4_scale
                c4["name"] += " [tables / %d]" % args.c4_scale
            sh = sharded.bench_entry(c4, min(K, 40), min(W, 8), rank, local_rank, world, workload_key="c4")
            extra["sharded_c4"] = dict(metric=sh["metric"], value=sh["value"], unit=sh["unit"], ratings_per_s=sh["value"],
                                       ms_per_step=sh["ms_per_step"], steps=sh["steps"], warmup=sh["warmup"], scaling="weak",
                                       config=sh["config"], roofline=sh["roofline"], phases_us=sh["roofline"].get("phases_us"))
    else:
        res = sharded.bench_entry(wl, K, W, rank, local_rank, world, workload_key=args.workload)
    # rank 0, in the same run, while the other ranks wait at the barrier below: the single-GPU step the N-GPU value scales
    # from and the CPU baseline (same workload, same box)
    if rank == 0 and not args.no_single_gpu_reference:
        def ref_of(one, note):
            return dict(value=one["value"], unit="ratings/s", ms_per_step=one["ms_per_step"], steps=one["steps"], warmup=one["warmup"], note=note)
        one = svd_single_gpu(args.workload, wl, K, W, local_rank, feeds=False, cpu=not args.no_cpu_baseline, convergence=False,
                             cpu_steps=900 if small else None)
        res["single_gpu_reference"] = ref_of(one, "the N=1 step (train_steps_drawn, device-drawn ids) on rank 0's GPU after the N-rank timed region")
        res["cpu_baseline"] = one.get("cpu_baseline")
        if "sharded_c4" in extra:
            c4 = dict(WORKLOADS["c4"])
            c3 = dict(WORKLOADS["c3"])
            if args.c4_scale > 1:
                for w in (c3, c4):
                    w["U"], w["I"] = w["U"] // args.c4_scale, w["I"] // args.c4_scale
            one4 = svd_single_gpu("c4", c4, 40, 8, local_rank, feeds=False, cpu=False, convergence=False)
            extra["sharded_c4"]["single_gpu_reference"] = ref_of(one4, "config 4's tables (all rows) through the fused step on rank 0's GPU alone, same run")
            if not args.no_cpu_baseline:
                one3 = svd_single_gpu("c3", c3, 10, 4, local_rank, feeds=False, cpu=True, convergence=False)
                cb = one3.get("cpu_baseline")
                if cb:
                    extra["sharded_c4"]["cpu_baseline"] = dict(cb, sample="config 3's tables (a lazy-Adam step touches 2 x 262144 rows whatever the table size; "
                                                               "config 4's 56 GB of tables are not rebuilt on the host); " + cb["sample"])
    res.update(extra)
    res["launcher"] = os.environ.get("TFR_BENCH_LAUNCHER", "torch.distributed.run or bench.py's own ranks")
    dist.barrier()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)                              # stdout is stdout again: the one line
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(res), flush=True)
    sys.stdout.flush()
    os.dup2(2, 1)                                         # (and whatever the teardown prints goes to stderr)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--warmup", type=int, default=90)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["c5", "als"])
    ap.add_argument("--adam-mode", default=None, choices=["tf1", "lazy"])
    ap.add_argument("--store-ratings", type=int, default=None, help="override the size of the rating store")
    ap.add_argument("--dim", type=int, default=None, help="override the workload's dim (not the BASELINE configuration: for kernel studies)")
    ap.add_argument("--zipf", type=float, default=0.0, help="c3/c4 training store: item ids ~ Zipf(a) instead of uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-convergence", action="store_true", help="skip the fixed 30-epoch val-RMSE leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the c3 / c4 / c5 objects of the default line")
    ap.add_argument("--no-sharded-c4", action="store_true", help="N>1: skip the row-sharded config-4 object")
    ap.add_argument("--no-single-gpu-reference", action="store_true", help="N>1: skip rank 0's single-GPU and CPU legs")
    ap.add_argument("--c4-scale", type=int, default=1, help="N>1 rehearsals: divide config 4's table rows by this")
    ap.add_argument("--only-north-star", action="store_true", help="just the dim=128 forward roofline run")
    ap.add_argument("--ns-users", type=int, default=10_000_000)
    ap.add_argument("--ns-items", type=int, default=1_000_000)
    ap.add_argument("--ns-batch", type=int, default=262144)
    ap.add_argument("--ns-dim", type=int, default=128)
    ap.add_argument("--ns-sequential", action="store_true")
    ap.add_argument("--ns-zipf", type=float, default=0.0, help="item ids ~ Zipf(a) instead of uniform (SURVEY 8d)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["TFR_BENCH_LAUNCHER"] = "bench.py's own ranks (subprocess per GPU)"
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import tfrecomm_amd as T
    if not torch.cuda.is_available():
        raise SystemExit("bench.py rank %d/%d needs an MI355X: no HIP device visible (there is no CPU path to time)" % (rank, world))
    if os.environ.get("TFR_SHARE_GPU"):          # rehearsal on a 1-GPU box: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dp = world == 1 and bool(os.environ.get("TFR_FORCE_DP"))     # 1-rank rehearsal of the N>1 step
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("TFR_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        # RCCL prints a version banner on stdout when a communicator comes up (the first one here, the side communicator of the
        # row-sharded step later): stdout is pointed at stderr until the ONE JSON line is due (multi_gpu restores it)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            t = torch.zeros(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(t)
            dist.all_to_all_single(torch.empty_like(t.repeat(world)), t.repeat(world))
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend)
        print("bench.py rank %d/%d: communicator up (backend %s, world size %d, device %d)"
              % (rank, world, backend, dist.get_world_size(), local_rank), file=sys.stderr, flush=True)

    if (world > 1 or force_dp) and (args.workload in ("als", "c5") or args.only_north_star):
        sys.stdout.flush()                               # single-GPU workloads: nothing below brings up another communicator
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        if rank != 0:
            return
    if args.workload == "als":
        print(json.dumps(als_bench(local_rank)), flush=True)
        return
    if args.workload == "c5":
        print(json.dumps(fm_forward_bench(local_rank, steps=min(args.steps, 100), warmup=min(args.warmup, 10), no_cpu=args.no_cpu_baseline)), flush=True)
        return
    if args.only_north_star:
        print(json.dumps(north_star_forward(local_rank, steps=args.steps, warmup=args.warmup, U=args.ns_users,
                                            I=args.ns_items, B=args.ns_batch, D=args.ns_dim, sequential=args.ns_sequential, zipf=args.ns_zipf)), flush=True)
        return
    if world > 1 or force_dp:
        multi_gpu(args, rank, local_rank, world, force_dp, saved_stdout)
        return

    wl = dict(WORKLOADS[args.workload])
    if args.adam_mode:
        wl["adam_mode"] = args.adam_mode
    if args.store_ratings:
        wl["N"] = args.store_ratings
    if args.dim:
        wl["D"] = args.dim
        wl["name"] = wl["name"] + " [dim overridden to %d]" % args.dim
    t_start = time.perf_counter()
    out = svd_single_gpu(args.workload, wl, args.steps, args.warmup, local_rank, zipf=args.zipf, cpu=not args.no_cpu_baseline,
                         convergence=not args.no_convergence)
    if not args.no_north_star:
        out["north_star_forward"] = north_star_forward(local_rank)                      # uniform ids: worst case for caches
        out["north_star_forward_zipf"] = north_star_forward(local_rank, zipf=1.05)      # reported separately
        # the same kernel on eight batches per launch (whole-set inference shape: tfr_forward_resident / eval)
        out["north_star_forward_8x_batch"] = north_star_forward(local_rank, steps=30, warmup=4, B=8 * 262144)
    if args.workload == "c2" and not args.no_configs:
        # BASELINE configs[2..4] in the same driver-run line: fixed short legs (40 steps after 8 warm-up), independent of --steps
        c3 = svd_single_gpu("c3", dict(WORKLOADS["c3"]), 40, 8, local_rank, feeds=False, cpu=not args.no_cpu_baseline, convergence=False)
        out["c3_train_step"] = c3
        c4 = svd_single_gpu("c4", dict(WORKLOADS["c4"]), 40, 8, local_rank, feeds=False, cpu=not args.no_cpu_baseline, convergence=False)
        if c3.get("cpu_baseline"):
            c4["cpu_baseline"] = dict(c3["cpu_baseline"], sample="config 3's figure from this run (10M x 1M rows): a lazy-Adam step touches 2 x 262144 rows "
                                      "whatever the table size, and config 4's 56 GB of fp32 tables + Adam state are not rebuilt on the host; " + c3["cpu_baseline"]["sample"])
        out["c4_tables_one_gpu"] = c4
        out["c5_fm_forward"] = fm_forward_bench(local_rank, steps=30, warmup=5, no_cpu=args.no_cpu_baseline)
    out["bench_wall_s"] = time.perf_counter() - t_start
    out["reference_readme"] = dict(note="README.md:63 batch=10000: 1.1 s/epoch ~ 8.2e5 ratings/s (derived, dim and "
                                        "hardware unstated) - context only, not this metric", ratings_per_s=8.2e5)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
