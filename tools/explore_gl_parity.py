"""Exploration (round 3): how far do the engine's and the oracle's Arnoldi factorisations of the Ginzburg-Landau stepper differ,
column by column, and how much of that is conditioning?  python tools/explore_gl_parity.py n kdim [threads]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
from oracle import oracle as ora

n, m = int(float(sys.argv[1])), int(sys.argv[2])
threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(64, ora.max_threads())
tau, nsub = 0.01, 1
ctx = lk.Context(device=0)
A = lk.ginzburg_landau_linop_gpu(n, ctx, tau=tau, nsub=nsub)
p = A.params
Ao = ora.GLOp(n, p["dx"], tau, nsub, p["nu"], p["gamma"], p["mu_c"], p["mu2"])
x0 = np.empty(n, dtype=np.complex128); ora.fill_counter(x0, 13); x0 /= np.linalg.norm(x0)


def engine(x):
    X = lk.krylov_basis_gpu(n, m + 1, np.complex128, ctx); X.upload(x.reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    assert lk.arnoldi(A, X, H) == 0
    return H


def oracle(op):
    X = np.zeros((n, m + 1), dtype=np.complex128, order="F"); X[:, 0] = x0
    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    ora.set_threads(threads)
    t0 = time.time()
    assert ora.arnoldi(op, X, H, fast=True) == 0
    ora.set_threads(1)
    return H, time.time() - t0


vx, vy = lk.dense_vector_gpu(n, np.complex128, ctx), lk.dense_vector_gpu(n, np.complex128, ctx)
def eng_op(x):
    vx.basis.upload(np.ascontiguousarray(x).reshape(-1, 1), vx.col)
    A.apply_matvec(vx, vy)
    return vy.to_array()

He = engine(x0)
rng = np.random.default_rng(0)
xp = x0 * (1.0 + 1e-14 * rng.standard_normal(n)); xp /= np.linalg.norm(xp)
Hp = engine(xp)
Ho, t1 = oracle(Ao)
Ho2, t2 = oracle(ora.PyOp(eng_op, np.complex128))
col = lambda Ha, Hb: np.array([np.abs(Ha[:, j] - Hb[:, j]).max() / np.abs(Hb[:, j]).max() for j in range(m)])
d_own, d_same, d_pert = col(He, Ho), col(He, Ho2), col(Hp, He)
sub = np.abs(np.diag(He, -1)); cn = np.array([np.linalg.norm(He[:j + 2, j]) for j in range(m)])
rz = lambda H: np.linalg.eigvals(H[:m, :m])
def ritz_diff(Ha, Hb, top=8):
    a, b = rz(Ha), rz(Hb)
    a = a[np.argsort(-np.abs(a))][:top]
    return np.array([np.abs(b - z).min() / abs(z) for z in a])
from scipy.linalg import eig as seig
w, vl, vr = seig(He[:m, :m], left=True, right=True)
kap = 1.0 / np.abs(np.sum(vl.conj() * vr, axis=0))
order = np.argsort(-np.abs(w))
print(json.dumps({"n": n, "m": m, "oracle_seconds": [t1, t2],
                  "H_engine_vs_oracle_own_operator": {"max": d_own.max(), "first8": d_own[:8].tolist(), "last8": d_own[-8:].tolist()},
                  "H_engine_vs_oracle_same_operator_values": {"max": d_same.max(), "first8": d_same[:8].tolist(), "last8": d_same[-8:].tolist()},
                  "H_engine_perturbed_1e-14_vs_engine": {"max": d_pert.max(), "first8": d_pert[:8].tolist(), "last8": d_pert[-8:].tolist()},
                  "kappa_step=|H(:,j)|/|H(j+1,j)|": {"max": (cn / sub).max(), "prod_log10": float(np.log10(cn / sub).sum())},
                  "ritz_top8_rel_diff_own": ritz_diff(He, Ho).tolist(), "ritz_top8_rel_diff_same": ritz_diff(He, Ho2).tolist(),
                  "ritz_top8_rel_diff_pert": ritz_diff(He, Hp).tolist(),
                  "ritz_top8_eigen_condition": kap[order][:8].tolist(), "H_norm": float(np.linalg.norm(He[:m, :m], 2))}))
