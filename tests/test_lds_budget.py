"""Static LDS-budget guard (no GPU needed): every (dim, batch, table size) the dispatcher can route to a kernel
with a shape-dependent LDS request must stay within a gfx950 CU's 160 KB.  tfr_lds_bytes computes static +
dynamic bytes on the host exactly as the launchers do (csrc/svd_kernels.h tile_step_static_lds /
tile_step_dyn_lds; the kernels static_assert those formulas against their own __shared__ declarations).
Background: two GPU faults in round 1 came from k_tile_step asking for more than a CU has."""
import ctypes as C

import pytest

from tfrecomm_amd import _lib as L

LDS_PER_CU = 160 * 1024
K_TILE_STEP, K_SEG_REDUCE_FWD, K_FRONT, K_MT_DRAW, K_SEG_REDUCE = 0, 1, 2, 3, 4


def lds(kernel, dim, batch, users, items):
    st, dy = C.c_int64(-1), C.c_int64(-1)
    rc = L.load().tfr_lds_bytes(kernel, dim, batch, users, items, C.byref(st), C.byref(dy))
    return rc, st.value, dy.value


def supported_dims():
    return [d for d in range(1, 257) if (d % 4 == 0) or d <= 64]


def test_every_selectable_tile_step_shape_fits_a_cu():
    worst, n = 0, 0
    for dim in supported_dims():
        for ntiles in range(1, 17):
            for bits in range(1, 15):                           # 2 ... 16384 bins (rows fit LDS bins)
                rows = 1 << bits
                for users, items in ((rows, 1), (1, rows), (rows, rows)):
                    rc, st, dy = lds(K_TILE_STEP, dim, ntiles * 1024, users, items)
                    if rc != 0:                                 # the dispatcher refuses the shape for this kernel
                        continue
                    n += 1
                    assert st > 0 and dy >= 4 * rows
                    assert st + dy <= LDS_PER_CU, (dim, ntiles, rows, st, dy)
                    worst = max(worst, st + dy)
    assert n > 10000                                            # the enumeration really ran
    assert worst > 64 * 1024                                    # ... and includes the 16384-bin shapes


def test_other_kernels_fit_a_cu():
    for dim in supported_dims():
        for kern in (K_SEG_REDUCE_FWD, K_SEG_REDUCE, K_MT_DRAW):
            rc, st, dy = lds(kern, dim, 262144, 10_000_000, 1_000_000)
            assert rc == 0 and 0 < st + dy <= LDS_PER_CU // 2   # two blocks per CU must remain possible
        for bits in range(1, 15):
            rc, st, dy = lds(K_FRONT, dim, 300000, 1 << bits, 7)
            if rc == 0:
                assert st + dy <= LDS_PER_CU


def test_bad_arguments_are_refused():
    assert lds(K_TILE_STEP, 257, 1024, 10, 10)[0] == L.ERR_ARG          # unsupported dim
    assert lds(K_TILE_STEP, 64, 17 * 1024, 10, 10)[0] == L.ERR_ARG      # 17 tiles: not the tile path
    assert lds(K_TILE_STEP, 64, 1024, 20000, 10)[0] == L.ERR_ARG        # rows beyond the LDS bins: radix path
    assert lds(9, 64, 1024, 10, 10)[0] == L.ERR_ARG
