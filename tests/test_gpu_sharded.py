"""The row-sharded step with the REAL kernels: two processes (gloo, exchange staged through host
memory) sharing the box's single GPU, against the float64 oracle and the single-GPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import RTOL, dup_heavy_ids, make_oracle, rand_tables, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kw, U, I, D, B, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfrecomm_amd import sharded, _lib as L
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        rs = np.random.RandomState(11)
        t = rand_tables(rs, U, I, D)
        ref = make_oracle(U, I, D, t, **kw)
        comm = sharded.Comm()
        m = sharded.ShardedSvd(U, I, D, comm, lambda ur, ir, d: sharded.HipShard(ur, ir, d, 0, **kw), device=dev)
        m.set_tables_from_global(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        # a rating store every rank holds a copy of: the last step is a pre-split one (each rank brings its own rows of it)
        Ns = 2 * B + 3
        su, si = dup_heavy_ids(rs, U, Ns), dup_heavy_ids(rs, I, Ns)
        sr = (rs.rand(Ns) < 0.5).astype(np.float32) if kw.get("loss") == "nll" else rs.randint(1, 6, Ns).astype(np.float32)
        keep = [torch.from_numpy(x).to(dev) for x in (su, si, sr)]
        m.backend.set_store(*keep)
        rs_own = np.random.RandomState(500 + rank)
        for s in range(steps):
            if s == steps - 1 and steps > 1:
                my = rs_own.randint(0, Ns, max(1, B // world))
                allids = [None] * world
                dist.all_gather_object(allids, my)
                union = np.concatenate(allids)
                u, i, r = su[union], si[union], sr[union]
                logits, mine, scal = m.train_step_local_ids(torch.from_numpy(my).to(dev))
                torch.cuda.synchronize()
                wl, wloss, wreg = ref.train_step(u, i, r)
                tol = RTOL * (s + 1)
                sc = scal.cpu().numpy()
                assert abs(sc[0] - wloss) <= tol * abs(wloss) and abs(sc[1] - wreg) <= tol * abs(wreg)
                n = int(m.backend.routed()["counts"][0].item())
                counts = [None] * world
                dist.all_gather_object(counts, n)
                assert sum(counts) == union.size
                continue
            u, i = dup_heavy_ids(rs, U, B), dup_heavy_ids(rs, I, B)
            r = (rs.rand(B) < 0.5).astype(np.float32) if kw.get("loss") == "nll" else rs.randint(1, 6, B).astype(np.float32)
            logits, mine, scal = m.train_step(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev),
                                              torch.from_numpy(r).to(dev))
            torch.cuda.synchronize()
            wl, wloss, wreg = ref.train_step(u, i, r)
            tol = RTOL * (s + 1)
            n = int(m.backend.routed()["counts"][0].item())
            own = mine.cpu().numpy()[:n]
            assert np.array_equal(own, np.flatnonzero(u // m.per_u == rank))
            assert rel_err(logits.cpu().numpy()[:n], wl[own]) <= tol, "logits step %d" % s
            sc = scal.cpu().numpy()
            assert abs(sc[0] - wloss) <= tol * abs(wloss) and abs(sc[1] - wreg) <= tol * abs(wreg)
        m.backend.sync()
        got = m.gather_global_tables()
        for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
            assert rel_err(got[tid], ref.tables()[tid]) <= RTOL * steps, "table %d" % tid
        assert m.backend.model.step == steps
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(optimizer="adam", adam_mode="tf1"), dict(optimizer="adam", adam_mode="lazy"),
                                dict(optimizer="sgd", loss="nll", item_abs=True, reg_bias=True, lr=5e-3, reg=0.01)])
def test_two_rank_sharded_step_matches_oracle(kw):
    mp.spawn(_worker, args=(2, _free_port(), kw, 500, 301, 64, 2000, 4), nprocs=2, join=True)


def test_two_rank_d128_long_runs():
    mp.spawn(_worker, args=(2, _free_port(), dict(optimizer="adam", adam_mode="lazy"), 40, 30, 128, 3000, 2),
             nprocs=2, join=True)


def test_device_routing_is_bit_exact_against_numpy():
    """csrc/shard.hip (owner filter by stable compaction, distinct items by radix sort, slots grouped by owner) against
    the NumPy statement in tests/fake_shard_backend.py: every output array, every rank of several worlds, duplicate-heavy
    and uniform ids, the exact-capacity and the slack-capacity regime, rows that do not divide by the world."""
    from tfrecomm_amd import sharded
    from tests.fake_shard_backend import OracleShard
    dev = torch.device("cuda", 0)
    cases = [(50, 31, 8, 120, 3, True), (5000, 3001, 16, 4097, 2, True), (100000, 70001, 8, 70000, 4, False),
             (1000003, 100003, 4, 200000, 8, False), (7, 5, 4, 64, 8, True)]
    for U, I, D, Bg, world, dup in cases:
        rs = np.random.RandomState(U)
        u = dup_heavy_ids(rs, U, Bg) if dup else rs.randint(0, U, Bg).astype(np.int32)
        i = dup_heavy_ids(rs, I, Bg) if dup else rs.randint(0, I, Bg).astype(np.int32)
        r = rs.randint(1, 6, Bg).astype(np.float32)
        tu, ti, tr = torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(r)
        for rank in sorted({0, world - 1, world // 2}):
            lo_u, hi_u = sharded.shard_range(U, world, rank)
            lo_i, hi_i = sharded.shard_range(I, world, rank)

            class _C(object):                            # capacities() needs only rank / world
                pass
            c = _C(); c.rank, c.world = rank, world
            hip = sharded.HipShard(hi_u - lo_u, hi_i - lo_i, D, 0)
            sh = sharded.ShardedSvd(U, I, D, c, lambda a, b, d: hip, device=dev)
            sample_cap, slot_cap = sh.capacities(Bg)
            ora = OracleShard(hi_u - lo_u, hi_i - lo_i, D)
            want_req = ora.route(tu, ti, tr, rank, world, U, I, sample_cap, slot_cap)
            got_req = hip.route(tu.to(dev), ti.to(dev), tr.to(dev), rank, world, U, I, sample_cap, slot_cap)
            hip.sync()
            assert np.array_equal(got_req.cpu().numpy(), want_req.numpy()), (U, I, world, rank)
            g, w = hip.routed(), ora.routed()
            for k in ("counts", "mine", "u_local", "slot"):
                assert np.array_equal(g[k].cpu().numpy(), w[k].numpy()), (k, U, I, world, rank)
            # the same global batch as rows of a rating store the rank holds (tfr_shard_route_ids): same routing, bit for bit
            Ns = Bg + 1000
            ids = rs.permutation(Ns)[:Bg].astype(np.int64)
            su = rs.randint(0, U, Ns).astype(np.int32); si = rs.randint(0, I, Ns).astype(np.int32)
            sr = rs.randint(1, 6, Ns).astype(np.float32)
            su[ids], si[ids], sr[ids] = u, i, r
            keep = [torch.from_numpy(x).to(dev) for x in (su, si, sr)]
            hip.set_store(*keep)
            got2 = hip.route_ids(torch.from_numpy(ids).to(dev), rank, world, U, I, sample_cap, slot_cap)
            hip.sync()
            assert np.array_equal(got2.cpu().numpy(), want_req.numpy()), ("ids", U, I, world, rank)
            g = hip.routed()
            for k in ("counts", "mine", "u_local", "slot"):
                assert np.array_equal(g[k].cpu().numpy(), w[k].numpy()), ("ids", k, U, I, world, rank)
            # pre-split batches: every sender's own rows grouped by owner (tfr_shard_bucket_ids), then the routing of what this
            # rank would receive from them (tfr_shard_route_recs) - both bit for bit against NumPy
            b_loc = max(1, Bg // world)
            pair_cap = sh.pair_capacity(b_loc)
            ora.set_store(torch.from_numpy(su), torch.from_numpy(si), torch.from_numpy(sr))
            recv_parts = []
            for sender in range(world):
                ids_s = np.random.RandomState(sender * 7 + U).randint(0, Ns, b_loc).astype(np.int64)
                want_send = ora.bucket_ids(torch.from_numpy(ids_s), world, U, pair_cap)
                if sender in (0, world - 1):               # the device kernels for two of the senders
                    got_send = hip.bucket_ids(torch.from_numpy(ids_s).to(dev), world, U, pair_cap)
                    hip.sync()
                    assert np.array_equal(got_send.cpu().numpy(), want_send.numpy()), ("bucket", U, I, world, sender)
                recv_parts.append(want_send.numpy()[rank * pair_cap:(rank + 1) * pair_cap])
            recv = np.ascontiguousarray(np.concatenate(recv_parts))
            sc2, sl2 = sh.capacities(b_loc * world)
            sc2 = min(sc2, world * pair_cap)
            want_req = ora.route_recs(torch.from_numpy(recv), rank, world, U, I, sc2, sl2)
            got_req = hip.route_recs(torch.from_numpy(recv).to(dev), rank, world, U, I, sc2, sl2)
            hip.sync()
            assert np.array_equal(got_req.cpu().numpy(), want_req.numpy()), ("recs", U, I, world, rank)
            g, w = hip.routed(), ora.routed()
            for k in ("counts", "mine", "u_local", "slot"):
                assert np.array_equal(g[k].cpu().numpy(), w[k].numpy()), ("recs", k, U, I, world, rank)
            hip.model.close()


def test_capacity_overflow_voids_the_step_loudly():
    from tfrecomm_amd import sharded, _lib as L
    import tfrecomm_amd as T
    dev = torch.device("cuda", 0)
    hip = sharded.HipShard(100, 100, 8, 0)
    rs = np.random.RandomState(0)
    u = torch.from_numpy(rs.randint(0, 100, 500).astype(np.int32)).to(dev)
    i = torch.from_numpy(rs.randint(0, 100, 500).astype(np.int32)).to(dev)
    r = torch.ones(500, device=dev)
    hip.route(u, i, r, 0, 1, 100, 100, 100, 100)         # 500 local samples, room for 100
    with pytest.raises(T.TfrError) as e:
        hip.sync()
    assert e.value.code == L.ERR_OOB and "capacit" in str(e.value)
    hip.route(u, i, r, 0, 1, 100, 100, 500, 10)          # ~100 distinct items, 10 slots
    with pytest.raises(T.TfrError):
        hip.sync()
    hip.route(u, i, r, 0, 1, 100, 100, 500, 100)         # fits
    hip.sync()
    hip.model.close()


def _pipelined_worker(rank, world, port, kw, U, I, D, B, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfrecomm_amd import sharded, _lib as L
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        rs = np.random.RandomState(21)
        t = rand_tables(rs, U, I, D)
        Ns = 4 * B + 7
        su, si = dup_heavy_ids(rs, U, Ns), dup_heavy_ids(rs, I, Ns)
        sr = rs.randint(1, 6, Ns).astype(np.float32)
        keep = [torch.from_numpy(x).to(dev) for x in (su, si, sr)]
        rs_own = np.random.RandomState(700 + rank)
        batches = [rs_own.randint(0, Ns, B) for _ in range(steps)]
        d_batches = [torch.from_numpy(b).to(dev) for b in batches]
        tables = {}
        for mode in ("pipelined", "plain"):
            ref = make_oracle(U, I, D, t, **kw)
            comm = sharded.Comm()
            m = sharded.ShardedSvd(U, I, D, comm, lambda ur, ir, d: sharded.HipShard(ur, ir, d, 0, **kw), device=dev)
            m.set_tables_from_global(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
            m.backend.set_store(*keep)
            for s in range(steps):
                nxt = d_batches[s + 1] if (mode == "pipelined" and s + 1 < steps and s != 2) else None   # step 3 starts without a prepared batch
                logits, mine, scal = m.train_step_local_ids(d_batches[s], nxt)
                torch.cuda.synchronize()
                allids = [None] * world
                dist.all_gather_object(allids, batches[s])
                union = np.concatenate(allids)
                wl, wloss, wreg = ref.train_step(su[union], si[union], sr[union])
                sc = scal.cpu().numpy()
                tol = RTOL * (s + 1)
                assert abs(sc[0] - wloss) <= tol * abs(wloss) and abs(sc[1] - wreg) <= tol * abs(wreg), (mode, s)
            m.backend.sync()
            got = m.gather_global_tables()
            for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
                assert rel_err(got[tid], ref.tables()[tid]) <= (2e-4 if kw.get("optimizer") == "adam" else RTOL * steps), "table %d (%s)" % (tid, mode)
            tables[mode] = {k: np.array(v) for k, v in m.local_tables().items()}
            m.backend.model.close()
        for k in tables["plain"]:                            # the front end reads no table: the same numbers, bit for bit
            assert np.array_equal(tables["plain"][k], tables["pipelined"][k]), "table %d differs between the two forms" % k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(optimizer="adam", adam_mode="lazy"), dict(optimizer="sgd", lr=2e-4)])
def test_two_rank_pipelined_front_end_equals_plain_steps(kw):
    """train_step_local_ids(ids, next_ids): the next batch's bucket / exchanges / routing / sorts run on a side stream into the
    model's second routed-batch set while the step's own kernels run; the next call starts at the row gather.  Five steps on two
    ranks (real kernels; gloo stages the exchanges through the host) against the float64 oracle, and bit for bit against the same
    steps without the pipelining."""
    mp.spawn(_pipelined_worker, args=(2, _free_port(), kw, 3000, 900, 64, 6000, 5), nprocs=2, join=True)


def _one_rank_overflows(rank, world, port, D):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfrecomm_amd import sharded, _lib as L
        import tfrecomm_amd as T
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        U, I, B = 2000, 300, 40000
        rs = np.random.RandomState(5)
        t = rand_tables(rs, U, I, D)
        kw = dict(optimizer="adam", adam_mode="lazy")
        comm = sharded.Comm()
        m = sharded.ShardedSvd(U, I, D, comm, lambda ur, ir, d: sharded.HipShard(ur, ir, d, 0, **kw), device=dev)
        m.set_tables_from_global(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        # store rows [0, B): every user in rank 1's block (a hot user block); rows [B, 2B): uniform users
        su = np.concatenate([rs.randint(U // 2, U, B), rs.randint(0, U, B)]).astype(np.int32)
        si = rs.randint(0, I, 2 * B).astype(np.int32)
        sr = rs.randint(1, 6, 2 * B).astype(np.float32)
        keep = [torch.from_numpy(x).to(dev) for x in (su, si, sr)]
        m.backend.set_store(*keep)
        assert m.pair_capacity(B) < B                            # slack-sized: rank 0's 40000 records for rank 1 do not fit
        before = {k: np.array(v) for k, v in m.local_tables().items()}
        # rank 0 brings only hot-block rows -> its bucket for owner 1 overflows; rank 1's own batch is harmless
        my = np.arange(B, dtype=np.int64) if rank == 0 else B + np.arange(B, dtype=np.int64)
        ids = torch.from_numpy(my).to(dev)
        m.train_step_local_ids(ids)
        m.train_step_local_ids(ids)                              # the flag is sticky until a sync: the next step is void as well
        with pytest.raises(T.TfrError) as e:
            m.backend.sync()
        assert e.value.code == L.ERR_OOB
        assert ("capacit" in str(e.value)) if rank == 0 else ("another rank" in str(e.value)), str(e.value)
        after = m.local_tables()
        for k in before:                                         # void on EVERY rank: no table moved anywhere
            assert np.array_equal(before[k], np.asarray(after[k])), "rank %d table %d moved in a void step" % (rank, k)
        # and the ranks are still in step with each other: a batch that fits goes through on both
        ok_ids = torch.from_numpy(B + rs.randint(0, B, B).astype(np.int64)).to(dev)
        m.train_step_local_ids(ok_ids)
        m.backend.sync()
        moved = m.local_tables()
        assert not np.array_equal(before[L.P], np.asarray(moved[L.P]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [64, 6])
def test_overflow_on_one_rank_voids_the_step_on_every_rank(D):
    """a capacity overflow on ONE rank (a hot user block) must not let the peers apply that rank's stale gradient
    rows: the error flag rides with the packed rows of the row exchange, every rank skips the same step and raises at its
    next sync.  D=6 takes the unvectorised row layout (stride D + 2)."""
    mp.spawn(_one_rank_overflows, args=(2, _free_port(), D), nprocs=2, join=True)


# ------------------------------------------------------------------ data parallel (replicated tables)
def _dp_worker(rank, world, port, kw, U, I, D, B, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfrecomm_amd import dataparallel, _lib as L
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        rs = np.random.RandomState(13)
        t = rand_tables(rs, U, I, D)
        ref = make_oracle(U, I, D, t, **kw)
        be = dataparallel.HipReplica(U, I, D, 0, **kw)
        be.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        dp = dataparallel.DataParallelSvd(be)
        for s in range(steps):
            u, i = dup_heavy_ids(rs, U, world * B), dup_heavy_ids(rs, I, world * B)
            r = (rs.rand(world * B) < 0.5).astype(np.float32) if kw.get("loss") == "nll" else rs.randint(1, 6, world * B).astype(np.float32)
            sl = slice(rank * B, (rank + 1) * B)
            scal = dp.train_step(torch.from_numpy(u[sl]).to(dev), torch.from_numpy(i[sl]).to(dev), torch.from_numpy(r[sl]).to(dev))
            _, wloss, wreg = ref.train_step(u, i, r)                    # ONE step on the global batch
            sc = scal.cpu().numpy()
            tol = RTOL * (s + 1)
            assert abs(sc[0] - wloss) <= tol * abs(wloss) and abs(sc[1] - wreg) <= tol * abs(wreg)
        be.sync()
        got = be.model.tables()
        for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
            assert rel_err(got[tid], ref.tables()[tid]) <= RTOL * steps, "table %d" % tid
        # replicas are bit-identical
        mine = np.concatenate([got[tid].reshape(-1) for tid in (L.MU, L.BU, L.BI, L.P, L.Q)])
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        assert all(np.array_equal(parts[0], p) for p in parts)
        assert be.model.step == steps and float(be.flat.abs().max()) == 0.0   # buffer left clean
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(optimizer="adam", adam_mode="tf1"),
                                dict(optimizer="sgd", loss="nll", item_abs=True, reg_bias=True, lr=5e-3, reg=0.01)])
def test_two_rank_data_parallel_equals_one_global_step(kw):
    mp.spawn(_dp_worker, args=(2, _free_port(), kw, 300, 200, 64, 1500, 4), nprocs=2, join=True)


def test_two_rank_data_parallel_large_table_route():
    # > 16384 rows: the global sort + dense_rows route instead of the tile-local one
    mp.spawn(_dp_worker, args=(2, _free_port(), dict(optimizer="adam", adam_mode="tf1"), 20000, 17000, 32, 3000, 3),
             nprocs=2, join=True)


def _dp_resident_worker(rank, world, port, U, I, D, B, steps):
    """Replicas fed from the resident store with the look-ahead hint (tfr_dp_hint_next): the tile sort of
    step s+1 rides in step s's launch.  Must equal one oracle step on the global batch, every step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfrecomm_amd import dataparallel, _lib as L
        torch.cuda.set_device(0)
        kw = dict(optimizer="adam", adam_mode="tf1")
        rs = np.random.RandomState(21)
        t = rand_tables(rs, U, I, D)
        N = 5000
        su, si = dup_heavy_ids(rs, U, N), dup_heavy_ids(rs, I, N)
        sr = rs.randint(1, 6, N).astype(np.float32)
        ids = rs.randint(0, N, (steps, world * B))
        ref = make_oracle(U, I, D, t, **kw)
        be = dataparallel.HipReplica(U, I, D, 0, **kw)
        be.model.set_tables(t["mu"], t["bu"], t["bi"], t["P"], t["Q"])
        be.model.upload_triples(su, si, sr)
        be.model.stage_ids(np.ascontiguousarray(ids[:, rank * B:(rank + 1) * B]))
        base, _ = be.model.staged_ids_devptr()
        dp = dataparallel.DataParallelSvd(be)
        for s in range(steps):
            scal = dp.train_step(store_ids_ptr=base + s * B * 8, batch=B,
                                 next_ids_ptr=base + (s + 1) * B * 8 if s + 1 < steps else None)
            _, wloss, wreg = ref.train_step(su[ids[s]], si[ids[s]], sr[ids[s]])
            sc = scal.cpu().numpy()
            tol = RTOL * (s + 1)
            assert abs(sc[0] - wloss) <= tol * abs(wloss) and abs(sc[1] - wreg) <= tol * abs(wreg), s
        be.sync()
        got = be.model.tables()
        for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
            assert rel_err(got[tid], ref.tables()[tid]) <= RTOL * steps, "table %d" % tid
        assert be.model.step == steps and float(be.flat.abs().max()) == 0.0
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_resident_lookahead():
    mp.spawn(_dp_resident_worker, args=(2, _free_port(), 300, 200, 64, 1500, 5), nprocs=2, join=True)


def test_data_parallel_rejects_lazy_adam():
    import tfrecomm_amd as T
    with T.SvdModel(10, 10, 8, optimizer="adam", adam_mode="lazy") as m:
        with pytest.raises(T.TfrError):
            m.dp_apply(1)


def _big_world1(rank, port, U=2_000_000, I=200_000):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import tfrecomm_amd as T
        from tfrecomm_amd import sharded, _lib as L
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        D, B = 128, 262144                                      # the slack-capacity regime (batch > 65536), radix paths
        kw = dict(optimizer="adam", adam_mode="lazy", lr=2e-3, reg=0.03)
        rs = np.random.RandomState(23)
        batches = [(rs.randint(0, U, B).astype(np.int32), rs.randint(0, I, B).astype(np.int32),
                    rs.randint(1, 6, B).astype(np.float32)) for _ in range(3)]
        comm = sharded.Comm()
        m = sharded.ShardedSvd(U, I, D, comm, lambda ur, ir, d: sharded.HipShard(ur, ir, d, 0, **kw), device=dev)
        sample_cap, slot_cap = m.capacities(B)
        assert sample_cap == B and slot_cap == min(I, B)        # world 1: everything is local, nothing may overflow
        m.backend.model.init_tables(seed=9)
        with T.SvdModel(U, I, D, **kw) as ref:
            ref.init_tables(seed=9)                             # counter-based initialiser: the same tables
            for s, (u, i, r) in enumerate(batches):
                logits, mine, scal = m.train_step(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(r).to(dev))
                torch.cuda.synchronize()
                wl, wloss, wreg = ref.train_step(u, i, r)
                n = int(m.backend.routed()["counts"][0].item())
                assert n == B and np.array_equal(mine.cpu().numpy(), np.arange(B))
                assert rel_err(logits.cpu().numpy(), wl) <= RTOL * (s + 1)
                sc = scal.cpu().numpy()
                assert abs(sc[0] - wloss) <= 2 * RTOL * abs(wloss) and abs(sc[1] - wreg) <= 2 * RTOL * abs(wreg)
            m.backend.sync()
            probe_u, probe_i = batches[0][0][:50000], batches[1][1][:50000]
            a = m.backend.model.forward(probe_u, probe_i)
            assert rel_err(a, ref.forward(probe_u, probe_i)) <= 4 * RTOL        # the touched rows moved the same way
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("U,I", [(2_000_000, 200_000), (12_500_000, 1_250_000)])
def test_one_rank_sharded_step_at_scale_equals_the_fused_single_gpu_step(U, I):
    """The row-sharded step (device routing in its slack-capacity regime, packed exchange buffers, dynamic counts) on a
    262144-rating batch, world 1: same logits, loss, regulariser and updated rows as the fused single-GPU step of the same
    model.  2M x 200k rows, and BASELINE config 4's per-rank shard at 8 GPUs (12.5M x 1.25M rows, batch 262144 per GPU)."""
    mp.spawn(_big_world1, args=(_free_port(), U, I), nprocs=1, join=True)
