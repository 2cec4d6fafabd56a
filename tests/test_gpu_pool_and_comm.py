"""The column pool behind per-object hosts (lk_pool_*: slabs, owner tags, generations, rank-independent slab geometry) and the native RCCL
communicator with a single rank (lk_comm_init_rank: the sum is the identity, so results are bit-identical to the run without one)."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, _pool_fns, _arnoldi_h
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


def test_native_rccl_single_rank_is_bit_identical():
    """lk_comm_init_rank with a 1-rank communicator: every sweep's scalars go through ncclAllReduce on the engine's
    stream; the sum over one rank is the identity, so H and the basis must be bit-identical to the run without it."""
    plain = lk.Context(device=0)
    H0, X0 = _arnoldi_h(plain)
    plain.close()
    c = lk.Context(device=0)
    uid = lk.Context.comm_unique_id()
    assert len(uid) == _capi.LK_COMM_ID_BYTES and any(uid)
    c.init_native_comm(1, 0, uid)
    H1, X1 = _arnoldi_h(c)
    with pytest.raises(_capi.LightKrylovHipError, match="already has a communicator"):
        c.init_native_comm(1, 0, uid)
    c.destroy_native_comm()
    H2, _ = _arnoldi_h(c)                      # and back
    c.close()
    assert H1.tobytes() == H0.tobytes() and X1.tobytes() == X0.tobytes() and H2.tobytes() == H0.tobytes()


def test_column_pool_contract(ctx):
    lib = _capi.load()
    st = (C.c_int64 * 4)()

    def stats():
        _capi.check(lib.lk_pool_stats(ctx._h, st))
        return tuple(st)

    def acquire(dtype, n, tag):
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(lib.lk_pool_acquire(ctx._h, dtype, n, C.c_uint64(tag), C.byref(slab), C.byref(col)))
        return slab.value, col.value

    def owner(slab, col):
        t = C.c_uint64()
        _capi.check(lib.lk_pool_owner(ctx._h, C.c_void_p(slab), col, C.byref(t)))
        return t.value

    _capi.check(lib.lk_pool_release_all(ctx._h))
    ctx.set_tuning("pool_slab_cols", 8)
    base = stats()
    n = 1000
    cols = [acquire(_capi.LK_F64, n, 0x1000 + 64 * i) for i in range(10)]      # 10 objects: 8 + 2 over two slabs
    assert [c for _s, c in cols[:8]] == list(range(8)) and len({s for s, _c in cols[:8]}) == 1   # consecutive, one slab
    assert cols[8][0] != cols[0][0] and cols[8][1] == 0
    assert stats()[0] - base[0] == 2 and stats()[2] == 10
    assert acquire(_capi.LK_F64, n, 0x1000 + 64 * 3) == cols[3]                 # same address again: same column
    assert owner(*cols[3]) == 0x1000 + 64 * 3 and owner(cols[0][0], 77) == 0 and owner(0xdead0, 0) == 0
    _capi.check(lib.lk_pool_release(ctx._h, C.c_void_p(cols[5][0]), cols[5][1]))
    _capi.check(lib.lk_pool_release(ctx._h, C.c_void_p(cols[2][0]), cols[2][1]))
    assert owner(*cols[2]) == 0
    assert acquire(_capi.LK_F64, n, 0x9000) == cols[2]                          # lowest released column first
    assert acquire(_capi.LK_F64, n, 0x9040) == cols[5]
    zslab, zcol = acquire(_capi.LK_C128, n, 0x1000)                             # other kind at a known address: new slab,
    assert zslab not in {s for s, _c in cols} and owner(*cols[0]) == 0          # and the old column is given back
    # the columns are real device vectors
    B = lk.krylov_basis_gpu(n, 8, np.float64, ctx, _handle=C.c_void_p(cols[1][0]))
    B._owner = B                                                                # not ours to destroy
    v = lk.dense_vector_gpu(_basis=B, _col=cols[1][1])
    v.rand(True, seed=3)
    assert abs(v.norm() - 1.0) < 1e-14
    B._h = C.c_void_p()
    _capi.check(lib.lk_pool_release_all(ctx._h))
    assert stats()[0] == 0 and stats()[2] == 0
    ctx.set_tuning("pool_slab_cols", 160)


def test_pool_generation_counter_exposes_stale_bit_copies():
    """A column's generation goes up every time the pool hands it out: first use, re-use by the same owner tag (an object
    re-created at a dead one's address), re-use after a release.  A handle that remembers the generation it was bound at --
    the Fortran plugin's does -- can tell that its column now belongs to something else."""
    c = lk.Context(device=0)
    lib, acquire, info = _pool_fns(c)
    c.set_tuning("pool_slab_cols", 8)
    a = acquire(_capi.LK_F64, 1000, 0x1000)
    ga = info(*a)[1]
    assert info(*a) == (0x1000, ga) and ga >= 1
    b = acquire(_capi.LK_F64, 1000, 0x2000)
    gb = info(*b)[1]
    assert info(*b)[0] == 0x2000 and gb > ga                                          # ONE counter per context: never the same value twice
    assert acquire(_capi.LK_F64, 1000, 0x1000) == a and info(*a)[1] > gb             # same address again: same column, a later generation
    _capi.check(lib.lk_pool_release(c._h, C.c_void_p(b[0]), b[1]))
    assert info(*b) == (0, gb)
    g3 = info(*a)[1]
    assert acquire(_capi.LK_F64, 1000, 0x3000) == b and info(*b)[0] == 0x3000 and info(*b)[1] > g3   # released column handed to another owner
    assert info(a[0], 7) == (0, 0) and info(0xdead0, 0) == (0, 0)                     # never carved / not a slab: safe to ask
    # a stale handle from before lk_pool_release_all must not match whatever a NEW slab hands out, even when that slab lands on
    # the freed one's heap address and the column index is the same: generations do not restart
    seen = {info(*a)[1], info(*b)[1]}
    _capi.check(lib.lk_pool_release_all(c._h))
    a2 = acquire(_capi.LK_F64, 1000, 0x1000)
    assert info(*a2)[1] not in seen and info(*a2)[1] > max(seen)
    _capi.check(lib.lk_pool_release_all(c._h))
    c.close()


def test_pool_slab_geometry_is_rank_independent_on_sharded_contexts():
    """On a single-rank context a slab shrinks to a quarter of the free memory; on a row-sharded one it must not (free memory
    differs between ranks; unequal slabs would desynchronise the lazy path's batched all-reduces): there a slab has exactly
    pool_slab_cols columns, or the acquisition fails."""
    import torch
    free_b, _total = torch.cuda.mem_get_info(0)
    want = 2048
    n = int(0.3 * free_b / (8.0 * want))                    # `want` columns = 0.3 of the free memory: more than a quarter, and it fits
    lib = _capi.load()

    def slab_cols(ctx):
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(lib.lk_pool_acquire(ctx._h, _capi.LK_F64, n, C.c_uint64(0x1000), C.byref(slab), C.byref(col)))
        nc = C.c_int()
        _capi.check(lib.lk_basis_info(slab, None, None, C.byref(nc), None, None))
        _capi.check(lib.lk_pool_release_all(ctx._h))
        return nc.value

    single = lk.Context(device=0)
    single.set_tuning("pool_slab_cols", want)
    got1 = slab_cols(single)
    single.close()
    assert got1 < want                                      # single rank: memory-derived
    cb = _capi.ALLREDUCE_FN(lambda _u, _p, _n, _s: 0)       # a 2-rank context (the reduction itself is not exercised here)
    sharded = lk.Context(device=0)
    _capi.check(lib.lk_set_allreduce(sharded._h, cb, None, 2, 0))
    sharded.set_tuning("pool_slab_cols", want)
    got2 = slab_cols(sharded)
    assert got2 == want                                     # sharded: exactly the configured geometry
    # ... and an impossible geometry fails loudly instead of shrinking
    sharded.set_tuning("pool_slab_cols", 4096)
    big = int(free_b / (8.0 * 4096) * 1.5)
    slab, col = C.c_void_p(), C.c_int()
    rc = lib.lk_pool_acquire(sharded._h, _capi.LK_F64, big, C.c_uint64(0x2000), C.byref(slab), C.byref(col))
    assert rc != 0 and b"pool_slab_cols" in lib.lk_last_error()
    _capi.check(lib.lk_set_allreduce(sharded._h, _capi.ALLREDUCE_FN(), None, 1, 0))
    sharded.close()
