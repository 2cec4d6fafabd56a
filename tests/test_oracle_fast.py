"""The multi-threaded oracle (oracle/lk_oracle_fast.inc) against the single-threaded restatement:
SEQUENTIAL mode must be BIT-IDENTICAL (it is the same schedule with independent sums run as tasks and
element-wise work split by rows); COMPENSATED mode must agree with an exact (fraction-free) dot."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np
import pytest

from oracle import oracle as ora


def _rand(n, k, dtype, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, k))
    if np.dtype(dtype).kind == "c":
        X = X + 1j * rng.standard_normal((n, k))
    return np.asfortranarray(X.astype(dtype))


def _bits_equal(a, b):
    return a.tobytes(order="A") == b.tobytes(order="A")


@pytest.fixture(autouse=True)
def _one_thread_after():
    yield
    ora.set_threads(1)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (1023, 7), (1024, 1), (10007, 20), (70001, 33)])
@pytest.mark.parametrize("threads", [1, 4])
def test_fast_dgs_is_bit_identical(dtype, n, k, threads):
    X = _rand(n, k, dtype, 1)
    X[:, k // 2] = 0                      # a zero coefficient exercises axpy's quick return
    y0 = _rand(n, 1, dtype, 2)[:, 0]
    y_ref, y_fast = y0.copy(), y0.copy()
    h_ref, i_ref = ora.double_gram_schmidt_step(y_ref, X)
    ora.set_threads(threads)
    h_fast, i_fast = ora.double_gram_schmidt_step(y_fast, X, fast=True)
    assert i_ref == i_fast
    assert _bits_equal(h_ref, h_fast)
    assert _bits_equal(y_ref, y_fast)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_fast_arnoldi_is_bit_identical(dtype):
    n, m = 30011, 24
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype) if dtype == np.float64 else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    X1 = np.zeros((n, m + 1), dtype=dtype, order="F")
    ora.fill_counter(X1[:, 0], 7)
    ora.scal(X1[:, 0], 1.0 / ora.norm(X1[:, 0]))
    X2 = X1.copy(order="F")
    H1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    H2 = H1.copy(order="F")
    assert ora.arnoldi(ora.DiagOp(d), X1, H1) == 0
    ora.set_threads(3)
    assert ora.arnoldi(ora.DiagOp(d), X2, H2, fast=True) == 0
    assert _bits_equal(H1, H2)
    assert _bits_equal(X1, X2)


def test_diaglin_operator_is_one_fma():
    n, row0 = 1001, 12345
    x = _rand(n, 1, np.float64, 3)[:, 0]
    y = np.empty(n)
    d0, dstep = 1.0, 1.0 / 1e8
    ora.DiagLinOp(d0, dstep, row0).matvec(x, y)
    for i in (0, 1, 500, n - 1):
        exact = Fraction(dstep) * (row0 + i) + Fraction(d0)
        d = float(exact)                                   # correctly rounded = fma
        assert y[i] == d * x[i]


def _exact_dot(x, y):
    return sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(x, y))


def test_compensated_dot_is_twice_working_precision():
    rng = np.random.default_rng(5)
    n = 200_001
    # ill-conditioned: large cancelling terms + small signal
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
    y = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
    exact = _exact_dot(x[:20001], y[:20001])
    got = ora.dot_mode(x[:20001].copy(), y[:20001].copy(), ora.COMPENSATED)
    scale = float(sum(abs(Fraction(float(a)) * Fraction(float(b))) for a, b in zip(x[:20001], y[:20001])))
    assert abs(Fraction(float(got)) - exact) <= Fraction(1.2e-16) * abs(exact) + Fraction(scale) * Fraction(1e-30)
    # thread-count independence (fixed chunking)
    a1 = ora.dot_mode(x, y, ora.COMPENSATED)
    ora.set_threads(4)
    a4 = ora.dot_mode(x, y, ora.COMPENSATED)
    assert a1 == a4
    # complex kind: conj on the first argument
    xz = (x[:5001] + 1j * y[:5001]).astype(np.complex128)
    yz = (y[:5001] - 0.5j * x[:5001]).astype(np.complex128)
    gz = ora.dot_mode(xz, yz, ora.COMPENSATED)
    re = _exact_dot(xz.real, yz.real) + _exact_dot(xz.imag, yz.imag)
    im = _exact_dot(xz.real, yz.imag) - _exact_dot(xz.imag, yz.real)
    mag = math.hypot(float(re), float(im))
    assert abs(float(Fraction(float(gz.real)) - re)) <= 4e-16 * mag + 1e-300
    assert abs(float(Fraction(float(gz.imag)) - im)) <= 4e-16 * mag + 1e-300


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_compensated_arnoldi_close_to_sequential(dtype):
    n, m = 20011, 16
    X1 = np.zeros((n, m + 1), dtype=dtype, order="F")
    ora.fill_counter(X1[:, 0], 7)
    ora.scal(X1[:, 0], 1.0 / ora.norm(X1[:, 0]))
    X2 = X1.copy(order="F")
    H1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    H2 = H1.copy(order="F")
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype)
    ora.arnoldi(ora.DiagOp(d), X1, H1)
    ora.set_threads(2)
    ora.arnoldi(ora.DiagOp(d), X2, H2, mode=ora.COMPENSATED)
    err = max(np.abs(H1[:, j] - H2[:, j]).max() / np.abs(H1[:, j]).max() for j in range(m))
    assert err < 1e-13


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_restarted_eigs_on_several_threads_is_bit_identical(dtype):
    """oracle eigs(fast=True): Arnoldi steps through the multi-threaded evaluation, the columns of the restart's basis update
    X <- X Z(:, :n) (BaseKrylov.fypp:816-824) and of the eigenvector reconstruction on a thread pool -- the checker of the
    restarted-eigs test at n = 10^6 -- must reproduce the one-thread restatement bit for bit; `stop_after_cycles` cuts the loop
    after a fixed number of cycles (the reference loops until convergence)."""
    n, nev, kdim = 1500, 3, 12
    rng = np.random.default_rng(7)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    x0 = rng.standard_normal(n)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * rng.standard_normal((n, n)) / np.sqrt(n)
        x0 = x0 + 1j * rng.standard_normal(n)
    A = np.asfortranarray(A.astype(dtype))
    ref = ora.eigs(ora.DenseOp(A), x0.astype(dtype), nev, kdim, 1e-30, stop_after_cycles=3)
    assert kdim + 2 <= ref[3] < 3 * kdim                                # one full cycle, then two from where the restarts left off
    ora.set_threads(3)
    try:
        got = ora.eigs(ora.DenseOp(A), x0.astype(dtype), nev, kdim, 1e-30, stop_after_cycles=3, fast=True)
    finally:
        ora.set_threads(1)
    assert got[3] == ref[3]
    for a, b in zip(got[:3], ref[:3]):
        assert np.array_equal(a, b)
