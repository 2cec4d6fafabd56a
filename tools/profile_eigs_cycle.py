"""Where the host time of one configs[3] `eigs` cycle goes (Ginzburg-Landau stepper, n = 1e6 complex(dp), kdim = 128, nev = 8,
one Krylov-Schur cycle + restart + eigenvectors): cProfile of the calling thread over a few steady-state calls, top entries by
cumulative time.   python tools/profile_eigs_cycle.py [reps]"""
import cProfile, os, pstats, sys, time, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = lk.Context(device=0)
n, kdim, nev = 1_000_000, 128, 8
A = lk.ginzburg_landau_linop_gpu(n, ctx, tau=0.01, nsub=1)
x0 = lk.dense_vector_gpu(n, np.complex128, ctx)
x0.rand(True, seed=13)
V = lk.krylov_basis_gpu(n, nev, np.complex128, ctx)
for _ in range(2):
    lk.eigs(A, V, x0=x0, kdim=kdim, tolerance=1e-10, max_restarts=0)
ctx.sync()
ts = []
pr = cProfile.Profile()
for _ in range(reps):
    t0 = time.perf_counter()
    pr.enable()
    lk.eigs(A, V, x0=x0, kdim=kdim, tolerance=1e-10, max_restarts=0)
    ctx.sync()
    pr.disable()
    ts.append(time.perf_counter() - t0)
print("ms per eigs call:", [round(1e3 * t, 1) for t in ts])
# timeline of one more call: the marks lightkrylov_amd.solvers leaves when asked (ms since the call began)
from lightkrylov_amd import solvers as _sv
_sv._eigs_trace = []
t0 = time.perf_counter()
lk.eigs(A, V, x0=x0, kdim=kdim, tolerance=1e-10, max_restarts=0)
ctx.sync()
t1 = time.perf_counter()
print("timeline of one cycle (ms since the call):")
for label, tm in _sv._eigs_trace:
    print(f"  {1e3 * (tm - t0):8.2f}  {label}")
print(f"  {1e3 * (t1 - t0):8.2f}  eigs returned")
_sv._eigs_trace = None
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
print(out.getvalue())
