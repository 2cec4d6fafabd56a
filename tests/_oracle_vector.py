"""TEST-ONLY abstract_vector implementation backed by the CPU oracle's arithmetic.

It exists so the host logic of lightkrylov_amd (the generic abstract_vector path of arnoldi /
gmres / eigs / krylov_schur, the row partition, the all-reduce plumbing) can be exercised
without a GPU, optionally row-sharded over a torch.distributed (gloo) group: each rank holds
a contiguous row block and `dot` finishes with a sum all-reduce -- exactly what a user's
distributed vector type does in the reference (paper/paper.md:35,97,101).
This file lives under tests/ and is never imported by the product.
"""
import numpy as np

from lightkrylov_amd.vectors import abstract_vector
from lightkrylov_amd.linops import abstract_linop
from oracle import oracle as ora


class oracle_vector(abstract_vector):
    def __init__(self, data: np.ndarray, group=None, row0: int = 0, seed: int = 0):
        self.data = np.ascontiguousarray(data)
        self.dtype = self.data.dtype
        self.group, self.row0, self.seed = group, row0, seed

    def zeros_like(self):
        return oracle_vector(np.zeros_like(self.data), self.group, self.row0, self.seed)

    def zero(self):
        self.data[:] = 0

    def rand(self, ifnorm=False, seed=None):
        ora.fill_counter(self.data, self.seed if seed is None else seed, self.row0)
        if ifnorm:
            self.scal(1.0 / self.norm())

    def scal(self, alpha):
        ora.scal(self.data, alpha)

    def axpby(self, alpha, vec, beta):
        if beta == 0:
            self.data[:] = 0               # true axpby
        ora.axpby(alpha, vec.data, beta, self.data)

    def dot(self, vec):
        val = ora.dot(self.data, vec.data)
        if self.group is not None:
            import torch
            import torch.distributed as dist
            t = torch.tensor([np.real(val), np.imag(val)], dtype=torch.float64)
            dist.all_reduce(t, group=self.group)
            val = complex(t[0].item(), t[1].item()) if self.dtype.kind == "c" else t[0].item()
        return val

    def get_size(self):
        return self.data.size


class oracle_diag_linop(abstract_linop):
    def __init__(self, d):
        super().__init__()
        self.d = np.ascontiguousarray(d)

    def matvec(self, vi, vo):
        vo.data[:] = self.d * vi.data

    def rmatvec(self, vi, vo):
        vo.data[:] = np.conj(self.d) * vi.data


class oracle_dense_linop(abstract_linop):
    def __init__(self, A):
        super().__init__()
        self.op = ora.DenseOp(A)
        self.A = self.op.A

    def matvec(self, vi, vo):
        self.op.matvec(vi.data, vo.data)

    def rmatvec(self, vi, vo):
        vo.data[:] = self.A.conj().T @ vi.data


class oracle_lap5_linop(abstract_linop):
    def __init__(self, N):
        super().__init__()
        self.op = ora.Lap5Op(N)

    def matvec(self, vi, vo):
        self.op.matvec(vi.data, vo.data)

    rmatvec = matvec
