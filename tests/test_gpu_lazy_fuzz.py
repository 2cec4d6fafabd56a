"""Randomised differential test of the lazy per-object path: the same sequence of type-bound-procedure calls on an EAGER
context and on a LAZY one (virtual temporaries, pending updates, memoised dots and norms, fused sweeps) must return the
same scalars and leave the same vectors.  The generator mixes the reference's own patterns (zero + a run of axpbys from
consecutive columns + y%sub(proj) + norm / dots) with calls that cut into them at every point: reads and writes of the
temporary, writes into the columns it is defined from, aliasing consumers, overwrites, copies, uploads, user-kernel
accesses (lk_vec_device_ptr)."""
import ctypes as C

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi

pytestmark = pytest.mark.gpu
NV = 12            # vectors 0..NV-1 are the columns of one panel, NV and NV+1 are stand-alone vectors


def _vec(B, S, i):
    return B[i] if i < NV else S[i - NV]


def _run(seed, dtype, lazy):
    rng = np.random.default_rng(seed)
    n = 1501
    cplx = np.dtype(dtype).kind == "c"
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1 if lazy else 0)
    B = lk.krylov_basis_gpu(n, NV, dtype, c)
    S = [lk.dense_vector_gpu(n, dtype, c) for _ in range(2)]
    for i in range(NV + 2):
        _vec(B, S, i).rand(False, seed=100 + i)
    out = []

    def scalar():
        v = float(rng.uniform(-1.5, 1.5))
        return complex(v, float(rng.uniform(-1.5, 1.5))) if cplx else v

    def projection():                       # linear_combination into T from a run of consecutive panel columns, then y%sub(T)
        T = int(rng.integers(0, NV + 2))
        j0 = int(rng.integers(0, NV - 1))
        cnt = int(rng.integers(1, NV - j0 + 1))
        cols = [j for j in range(j0, j0 + cnt) if j != T]
        if rng.random() < 0.85:
            _vec(B, S, T).zero()
        for j in cols:
            _vec(B, S, T).axpby(scalar(), B[j], 1.0)
        return T, cols

    for _step in range(120):
        r = rng.random()
        if r < 0.30:
            T, cols = projection()
            y = int(rng.integers(0, NV + 2))
            if y != T:
                _vec(B, S, y).axpby(-1.0 if rng.random() < 0.7 else scalar(), _vec(B, S, T), 1.0)
                k = rng.random()
                if k < 0.5:
                    out.append(_vec(B, S, y).norm())
                    for j in cols[: int(rng.integers(0, len(cols) + 1))]:
                        out.append(B[j].dot(_vec(B, S, y)))
                elif k < 0.7:
                    out.append(_vec(B, S, y).dot(_vec(B, S, y)))
        elif r < 0.36:                          # gram_matrix's loop over a sub-range (self fixed, vec running), cut by a write
            i0 = int(rng.integers(0, NV - 3))
            i1 = int(rng.integers(i0 + 2, NV))
            cut = int(rng.integers(0, 40)) if rng.random() < 0.5 else -1
            cnt = 0
            for i in range(i0, i1 + 1):
                for j in range(i, i1 + 1):
                    out.append(B[i].dot(B[j]))
                    cnt += 1
                    if cnt == cut:
                        B[int(rng.integers(0, NV))].scal(scalar())
        elif r < 0.40:
            a, b = rng.integers(0, NV + 2, 2)
            out.append(_vec(B, S, int(a)).dot(_vec(B, S, int(b))))
        elif r < 0.47:
            out.append(_vec(B, S, int(rng.integers(0, NV + 2))).norm())
        elif r < 0.55:
            _vec(B, S, int(rng.integers(0, NV + 2))).scal(scalar())
        elif r < 0.65:
            a, b = rng.integers(0, NV + 2, 2)
            if a != b:
                beta = [0.0, 1.0, scalar()][int(rng.integers(0, 3))]
                _vec(B, S, int(b)).axpby(scalar(), _vec(B, S, int(a)), beta)
        elif r < 0.72:
            a, b = rng.integers(0, NV + 2, 2)
            if a != b:
                lk.copy(_vec(B, S, int(b)), _vec(B, S, int(a)))
        elif r < 0.77:
            _vec(B, S, int(rng.integers(0, NV + 2))).zero()
        elif r < 0.81:
            _vec(B, S, int(rng.integers(0, NV + 2))).rand(False, seed=int(rng.integers(1, 1000)))
        elif r < 0.86:
            v = _vec(B, S, int(rng.integers(0, NV + 2)))
            out.append(float(np.abs(v.to_array()).sum()))                                   # single-vector download
        elif r < 0.90:
            v = _vec(B, S, int(rng.integers(0, NV + 2)))
            v.basis.upload(np.full((n, 1), scalar(), dtype=dtype), v.col)                   # single-vector upload
        elif r < 0.95:
            v = _vec(B, S, int(rng.integers(0, NV + 2)))                                    # a user's kernel on the vector
            acc = ["r", "w", "rw"][int(rng.integers(0, 3))]
            with c.torch_stream():                                                          # ordered with the engine's stream
                t = v.as_torch(acc)
                if acc == "r":
                    out.append(float(t.abs().sum().item()))
                elif acc == "w":
                    t.fill_(0.25)
                else:
                    t.mul_(0.5)
        else:
            k = int(rng.integers(1, NV))
            M = lk.innerprod(B[:k], B[k])                                                   # a panel-level call flushes everything
            out.extend(np.atleast_1d(M).tolist())
    final = [B.download()] + [s.to_array() for s in S]
    stats = c.lazy_fusion_stats() + c.lazy_stats()
    del B, S
    c.close()
    return out, final, stats


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_lazy_context_is_observationally_equal_to_an_eager_one(dtype):
    fused = virtual = 0
    for seed in range(24):
        oe, fe, _ = _run(seed, dtype, lazy=False)
        ol, fl, st = _run(seed, dtype, lazy=True)
        assert len(oe) == len(ol)
        scale = max(1.0, max((abs(v) for v in oe), default=1.0))
        for i, (a, b) in enumerate(zip(oe, ol)):
            assert abs(a - b) <= 1e-12 * scale, f"seed {seed}: scalar #{i}: eager {a}, lazy {b}"
        for a, b in zip(fe, fl):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max()), f"seed {seed}: final vectors differ"
        fused += st[0]
        virtual += st[2]
    assert fused > 50 and virtual > 20          # the sequences really exercised the fused sweep and unwritten temporaries
