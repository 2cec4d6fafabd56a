"""lightkrylov_amd -- MI355X-native engine for LightKrylov's Krylov inner loop.

Scope: the abstract_vector primitives (axpby, dot, norm, scal, copy) and the
double_gram_schmidt_step orthogonalisation that arnoldi / lanczos / gmres / eigs spend their
time in, as hand-written HIP kernels for gfx950 behind a C ABI (include/lightkrylov_hip.h).
This package is the host-side mirror of the reference's interface for that path: same names,
argument meaning and error behaviour as LightKrylov's Fortran modules.

Importing the package is cheap and does not touch the GPU; the first object that needs the
engine loads ``liblightkrylov_hip.so`` and raises if it has not been built or no HIP device
exists.  There is no CPU fallback.
"""
from .constants import atol_dp, rtol_dp  # noqa: F401
from .context import Context, default_context, row_partition  # noqa: F401
from .vectors import (Gram, abstract_vector, axpby_basis, copy, dense_vector_gpu, innerprod,  # noqa: F401
                      krylov_basis_gpu, linear_combination, rand_basis, verify_vector_axioms, zero_basis)
from .linops import (Id, abstract_linop, adjoint_linop, axpby_linop, scaled_linop, csr_linop_gpu, dense_linop_gpu, diag_linop_gpu, ginzburg_landau_linop_gpu,
                     grid_partition, laplacian2d_linop_gpu)  # noqa: F401
from .krylov import (arnoldi, bidiagonalization, double_gram_schmidt_step, is_orthonormal, krylov_schur, lanczos,  # noqa: F401
                     orthogonalize_against_basis, qr)
from .solvers import (apply_givens_rotation, cg, cg_dp_metadata, cg_dp_opts, eig, eighs, eigs, gmres, gmres_dp_metadata,  # noqa: F401
                      gmres_dp_opts, svds)

from .outputs import save_eigenspectrum, write_results  # noqa: F401

__version__ = "0.1.0"
