"""The minibatch id draw on the device (rng.hip, include/tfrecomm.h "id draw"): NumPy's legacy
``np.random.randint(0, N, (B,))`` (dataio.py:115, seeded at svd_train_val.py:15) replayed bit for bit.

Index work is a bit-exact target: checked against NumPy itself, against the id streams the REAL
reference dataio.py produced (tests/golden/iter_streams.npz) and, end to end, by training on
device-drawn ids vs the same steps with host-drawn ids.  Nothing here reads /root/reference."""
import hashlib

import numpy as np
import pytest

import tfrecomm_amd as T
from tfrecomm_amd import _lib as L

pytestmark = pytest.mark.gpu


def _digest(m):
    h = hashlib.sha256()
    for k, v in sorted(m.tables().items()):
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


@pytest.fixture
def model():
    with T.SvdModel(50, 40, 8, device=0) as m:
        yield m


def test_draw_matches_numpy_and_leaves_numpy_state(model):
    m = model
    np.random.seed(13575)                                        # svd_train_val.py:15
    m.rng_seed(13575)
    # ranges around powers of two (no rejection / ~50 % rejection), tiny, ML-1M's 900188, 90M; counts that
    # end inside, exactly at and just past a 624-word block
    cases = [(900188, 10000), (7, 5), (1 << 20, 3000), ((1 << 20) + 1, 3000), (90_000_000, 5000), (2, 1), (2, 623), (2, 1),
             (2, 624), (2, 625), (1, 17), (3, 0), ((1 << 32) - 1, 2000), (1 << 32, 1500), (1000209, 262144),
             # the wide form (k_mt_blocks / k_mt_count / k_mt_emit, from 32768 ids): at its threshold, without rejection, at ~50 %
             # rejection, the C3 draw, tiny ranges, and more ids than one wide pass holds (2^22)
             (900188, 32767), (900188, 32768), (1 << 20, 65536), ((1 << 20) + 1, 100000), (90_000_000, 262144), (2, 40000),
             (3, 50001), (70_000_000, 5_000_000), (1 << 32, 33000), ((1 << 31) + 1, 70000)]
    for high, count in cases:
        want = np.random.randint(0, high, (count,))
        got = m.draw_ids(high, count)
        assert got.dtype == want.dtype == np.int64
        assert np.array_equal(got, want), (high, count)
        key, pos = m.rng_get_state()
        st = np.random.get_state()
        assert pos == st[2] and np.array_equal(key, st[1]), (high, count)
    # hand-over both ways: host draws, device continues, host continues
    np.random.randint(0, 77, 1000)
    m.rng_from_numpy()
    a = m.draw_ids(12345, 4321)
    assert np.array_equal(a, np.random.randint(0, 12345, 4321))
    np.random.seed(1)
    m.rng_to_numpy()                                             # back to where the device is
    assert np.array_equal(np.random.randint(0, 999, 100), m.draw_ids(999, 100))


def test_draw_into_device_memory_and_join(model):
    """tfr_draw_ids_dev / tfr_join_draws (the data-parallel loop's feed): asynchronous draws into caller-owned device
    buffers, the same stream and final state as NumPy; a buffer is only overwritten after the work queued before the call."""
    import torch
    m = model
    np.random.seed(99)
    m.rng_seed(99)
    bufs = [torch.empty(70000, dtype=torch.int64, device="cuda") for _ in range(3)]
    want = []
    for k, (high, count) in enumerate(((900188, 20000), (900188, 70000), (50, 1), (1 << 20, 40000), (1, 5))):
        want.append(np.random.randint(0, high, (count,)))
        m.draw_ids_dev(high, count, bufs[k % 3].data_ptr())
        if k % 3 == 2 or k == 4:
            m.join_draws()
            m.sync()
            torch.cuda.synchronize()
            for q in range(k - (k % 3), k + 1):
                assert np.array_equal(bufs[q % 3][: want[q].size].cpu().numpy(), want[q]), q
    key, pos = m.rng_get_state()
    st = np.random.get_state()
    assert pos == st[2] and np.array_equal(key, st[1])
    with pytest.raises(T.TfrError):
        m.draw_ids_dev(0, 5, bufs[0].data_ptr())


def test_join_one_draw_of_several_in_flight(model):
    """tfr_join_draw(ordinal): with several draws in flight a caller waits for the one whose buffer it reads next (issued draws
    are counted from 1 and finish in issue order) - more draws than the event ring holds, joined one by one and out of step."""
    import torch
    m = model
    np.random.seed(5)
    m.rng_seed(5)
    store = np.random.RandomState(1)
    U, I = m.user_num, m.item_num
    n = 5000
    m.upload_triples(store.randint(0, U, n).astype(np.int32), store.randint(0, I, n).astype(np.int32), np.ones(n, np.float32))
    bufs = [torch.empty(3000, dtype=torch.int64, device="cuda") for _ in range(12)]
    want = []
    for k in range(12):                                      # twelve draws issued back to back: the ring holds eight events
        want.append(np.random.randint(0, n, (3000,)))
        m.draw_ids_dev(n, 3000, bufs[k].data_ptr())
    for k in (2, 0, 11, 7):
        m.join_draw(k + 1)
        m.sync()                                             # the model's stream has waited for draw k (and all before it)
        assert np.array_equal(bufs[k].cpu().numpy(), want[k]), k
    with pytest.raises(T.TfrError):
        m.join_draw(0)
    m.join_draws()
    m.sync()
    for k in range(12):
        assert np.array_equal(bufs[k].cpu().numpy(), want[k]), k


def test_wide_draw_that_comes_up_short_is_finished_by_the_sequential_kernel():
    """TFR_RNG_WIDE_TRIM makes the wide form generate too few blocks: the k_mt_draw launch behind it draws the rest from
    the state the last block left - same ids, same final state (a fresh process: the switch is read once)."""
    import os, subprocess, sys
    code = (
        "import numpy as np, torch\n"
        "torch.cuda.init(); torch.zeros(1, device='cuda')\n"
        "import tfrecomm_amd as T\n"
        "m = T.SvdModel(50, 40, 8, device=0)\n"
        "np.random.seed(7); m.rng_seed(7)\n"
        "for high, count in ((900188, 50000), ((1 << 20) + 1, 262144), (90_000_000, 40000)):\n"
        "    want = np.random.randint(0, high, (count,)); got = m.draw_ids(high, count)\n"
        "    assert np.array_equal(got, want), (high, count)\n"
        "    key, pos = m.rng_get_state(); st = np.random.get_state()\n"
        "    assert pos == st[2] and np.array_equal(key, st[1]), (high, count)\n"
        "print('short draws ok')\n")
    for trim in ("60", "97"):
        env = dict(os.environ, TFR_RNG_WIDE_TRIM=trim)
        out = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "short draws ok" in out.stdout, out.stderr[-2000:]


def test_draw_matches_the_reference_iterators_own_streams(model, golden):
    """tests/golden/iter_streams.npz holds the ids the real /root/reference/dataio.py ShuffleIterator drew."""
    g = golden("iter_streams.npz")
    keys = sorted({k.split("/")[0] for k in g.files if k.startswith("shuffle_")})
    assert len(keys) == 3
    for key in keys:
        N, B = int(key.split("_")[1][1:]), int(key.split("_")[2][1:])
        ids = g[key + "/ids"]
        model.rng_seed(int(g[key + "/seed"]))
        for k in range(ids.shape[0]):
            assert np.array_equal(model.draw_ids(N, B), ids[k]), (key, k)


def test_errors(model):
    with pytest.raises(T.TfrError) as e:
        model.draw_ids(10, 5)                                    # no state yet
    assert e.value.code == L.ERR_STATE
    model.rng_seed(1)
    for bad in (0, -3, (1 << 32) + 1):
        with pytest.raises(T.TfrError) as e:
            model.draw_ids(bad, 5)
        assert e.value.code == L.ERR_ARG
    with pytest.raises(T.TfrError) as e:
        model.train_steps_drawn(16, 2)                           # no resident store
    assert e.value.code == L.ERR_STATE


@pytest.mark.parametrize("U,I,D,B,N,steps,mode", [
    (6040, 3952, 64, 10000, 900188, 7, "tf1"),                   # the headline shape (k_tile_step + look-ahead sort)
    (300, 200, 20, 1000, 5000, 5, "tf1"),
    (300, 200, 20, 700, 5000, 4, "lazy"),
    (40000, 30000, 32, 4096, 200000, 5, "lazy"),                 # big-table path: gather + radix sort on the side stream
    (40000, 30000, 32, 4096, 200000, 3, "tf1"),
    (120, 90, 8, 300, 1, 3, "tf1"),                              # a one-rating store: every id is 0, no draw consumed
])
def test_drawn_steps_equal_host_drawn_steps(U, I, D, B, N, steps, mode):
    rs = np.random.RandomState(3)
    su, si = rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32)
    sr = rs.randint(1, 6, N).astype(np.float32)
    digests, losses = [], []
    for device_draw in (False, True):
        with T.SvdModel(U, I, D, adam_mode=mode, device=0) as m:
            m.init_tables(seed=5)
            m.upload_triples(su, si, sr)
            np.random.seed(13575)
            if device_draw:
                m.rng_from_numpy()
                loss = np.concatenate([m.train_steps_drawn(B, 2, want_loss=True), m.train_steps_drawn(B, steps - 2, want_loss=True)])
                m.rng_to_numpy()
            else:
                ids = np.random.randint(0, N, (steps, B))        # dataio.py:115, one draw per step
                loss = m.train_steps_resident(ids, B)
            tail = np.random.randint(0, 1 << 30, 8)              # the host stream continues identically
            digests.append((_digest(m), tail.tolist()))
            losses.append(loss)
    assert digests[0] == digests[1]
    assert np.array_equal(losses[0], losses[1])


def test_host_ids_async_steps_equal_resident_steps():
    U, I, D, B, N, steps = 6040, 3952, 64, 10000, 900188, 6
    rs = np.random.RandomState(4)
    su, si = rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32)
    sr = rs.randint(1, 6, N).astype(np.float32)
    ids = rs.randint(0, N, (steps, B))
    out = []
    for ring in (False, True):
        with T.SvdModel(U, I, D, device=0) as m:
            m.init_tables(seed=5)
            m.upload_triples(su, si, sr)
            if ring:
                for s in range(steps):
                    m.train_step_ids(ids[s])                     # no host sync between steps
                m.sync()
            else:
                m.train_steps_resident(ids, B)
            out.append(_digest(m))
    assert out[0] == out[1]
    with T.SvdModel(U, I, D, device=0) as m:
        m.init_tables(seed=5)
        m.upload_triples(su, si, sr)
        m.train_step_ids(np.array([0, N], dtype=np.int64))       # out-of-range store index: reported at the next sync
        with pytest.raises(IndexError):
            m.sync()


def test_run_ahead_between_calls_is_invisible():
    """After a drawn call the generator keeps going into an alternate buffer for the NEXT call; a call with another
    batch size or length, a host-visible draw, a state read or a new store in between must all see the stream exactly
    where the consumed ids end."""
    U, I, D, N = 300, 200, 16, 20000
    rs = np.random.RandomState(8)
    su, si = rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32)
    sr = rs.randint(1, 6, N).astype(np.float32)
    N2 = 7777
    plan = [("steps", 1000, 3), ("steps", 1000, 5), ("steps", 1000, 1), ("steps", 700, 4), ("draw", 12345, 50), ("steps", 700, 9),
            ("state",), ("steps", 700, 2), ("store", N2), ("steps", 256, 40), ("steps", 256, 40), ("steps", 256, 300),
            # calls of >= 8 steps end by sorting the next call's first batch (drawn ahead) inside their last launch: the same
            # call again takes that sort; a host-fed step, another batch size or a short call in between must not
            ("steps", 1000, 20), ("steps", 1000, 20), ("fed", 500), ("steps", 1000, 20), ("steps", 1000, 8), ("steps", 1000, 8),
            ("steps", 900, 8), ("steps", 1000, 3), ("steps", 1000, 12), ("state",), ("steps", 1000, 12)]
    out = []
    for device_draw in (False, True):
        with T.SvdModel(U, I, D, adam_mode="lazy", device=0) as m:
            m.init_tables(seed=2)
            m.upload_triples(su, si, sr)
            n_store = N
            np.random.seed(99)
            if device_draw:
                m.rng_from_numpy()
            log = []
            for op in plan:
                if op[0] == "steps":
                    _, B, k = op
                    if device_draw:
                        log.append(m.train_steps_drawn(B, k, want_loss=True).tobytes())
                    else:
                        log.append(m.train_steps_resident(np.random.randint(0, n_store, (k, B)), B).tobytes())
                elif op[0] == "fed":
                    k0 = len(log) * 37 % (n_store - op[1])
                    lg, loss, _ = m.train_step(su[k0:k0 + op[1]], si[k0:k0 + op[1]], sr[k0:k0 + op[1]])
                    log.append(lg.tobytes())
                elif op[0] == "draw":
                    log.append((m.draw_ids(op[1], op[2]) if device_draw else np.random.randint(0, op[1], op[2])).tobytes())
                elif op[0] == "state":
                    if device_draw:
                        key, pos = m.rng_get_state()
                    else:
                        st = np.random.get_state()
                        key, pos = st[1], st[2]
                    log.append((key.tobytes(), int(pos)))
                else:
                    n_store = op[1]
                    m.upload_triples(su[:n_store], si[:n_store], sr[:n_store])
            if device_draw:
                m.rng_to_numpy()
            log.append(np.random.randint(0, 1 << 30, 4).tobytes())
            out.append((_digest(m), log))
    assert out[0][0] == out[1][0]
    assert out[0][1] == out[1][1]


_RECS_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import tfrecomm_amd as T
from tfrecomm_amd import _lib as L
U, I, D, B = 6040, 3952, 64, 10000
rs = np.random.RandomState(5)
with T.SvdModel(U, I, D, optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05) as m:
    m.init_tables(seed=7)
    N = 300000
    m.upload_triples(rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32), rs.randint(1, 6, N).astype(np.float32))
    np.random.seed(3)
    m.rng_from_numpy()
    h = hashlib.sha256()
    for k in (1, 7, 20, 3, 33):                      # calls of odd lengths: ids drawn ahead between calls, buffers swapped
        h.update(m.train_steps_drawn(B, k, want_loss=True).tobytes())
    ids = np.random.randint(0, N, (4 * B,)).astype(np.int64)      # host-staged ids in the same buffer, then drawn again
    m.stage_ids(ids)
    m.train_steps_staged(0, B, 4)
    h.update(m.train_steps_drawn(B, 5, want_loss=True).tobytes())
    for tid in (L.MU, L.BU, L.BI, L.P, L.Q):
        h.update(np.ascontiguousarray(m.get_table(tid)).tobytes())
    print("HASH", h.hexdigest())
"""


def test_records_beside_the_drawn_ids_are_invisible():
    """TFR_RECS=0: the small-table step's sorts read ids -> store; default: the store records the drawing stream left beside
    the ids.  Same losses and tables, bit for bit, over calls of odd lengths (ids drawn ahead, buffers swapped), host-staged
    ids in between (no records) and drawn ids again."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for flag in ("1", "0"):
        env = dict(os.environ, TFR_RECS=flag)
        p = subprocess.run([sys.executable, "-c", _RECS_SCRIPT % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out.append([l for l in p.stdout.decode().splitlines() if l.startswith("HASH")][0])
    assert out[0] == out[1]
