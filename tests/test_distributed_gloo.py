"""N > 1 path on CPU: two processes (gloo), each owning a contiguous row block; the only cross-rank
step is the sum all-reduce inside `dot` -- the same place the HIP engine calls its RCCL callback.
The row-sharded Arnoldi / GMRES must reproduce the single-process result (H to 1e-13 normwise: the
partial sums are combined in a different order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, m, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lightkrylov_amd as lk
        from oracle import oracle as ora
        from tests._oracle_vector import oracle_diag_linop, oracle_vector
        row0, nl = lk.row_partition(n, world, rank)
        d = 1.0 + (row0 + np.arange(nl)) / n
        X = [oracle_vector(np.zeros(nl), dist.group.WORLD, row0) for _ in range(m + 1)]
        X[0].rand(True, seed=7)                                  # same global vector for every partition
        H = np.zeros((m + 1, m), order="F")
        info = lk.arnoldi(oracle_diag_linop(d), X, H)
        # sharded GMRES on the same operator
        b = oracle_vector(np.zeros(nl), dist.group.WORLD, row0); b.rand(False, seed=11)
        x = b.zeros_like()
        meta = lk.gmres_dp_metadata()
        ginfo = lk.gmres(oracle_diag_linop(d), b, x, rtol=1e-10, options=lk.gmres_dp_opts(kdim=20, maxiter=3), meta=meta)
        if rank == 0:
            np.savez(out, H=H, info=info, ginfo=ginfo, res=np.array(meta.res))
        np.save(f"{out}.x{rank}.npy", x.data)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sharded_arnoldi_and_gmres_world2(tmp_path):
    sys.path.insert(0, ROOT)
    import lightkrylov_amd as lk
    from oracle import oracle as ora
    from tests._oracle_vector import oracle_diag_linop, oracle_vector
    n, m, world = 20_001, 12, 2
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, m, out), nprocs=world, join=True)
    got = np.load(out)

    d = 1.0 + np.arange(n) / n
    X = [oracle_vector(np.zeros(n)) for _ in range(m + 1)]
    X[0].rand(True, seed=7)
    H = np.zeros((m + 1, m), order="F")
    info = lk.arnoldi(oracle_diag_linop(d), X, H)
    assert int(got["info"]) == info == 0
    for j in range(m):
        assert np.abs(got["H"][:, j] - H[:, j]).max() <= 1e-13 * np.abs(H[:, j]).max()

    b = oracle_vector(np.zeros(n)); b.rand(False, seed=11)
    x = b.zeros_like()
    meta = lk.gmres_dp_metadata()
    ginfo = lk.gmres(oracle_diag_linop(d), b, x, rtol=1e-10, options=lk.gmres_dp_opts(kdim=20, maxiter=3), meta=meta)
    assert int(got["ginfo"]) == ginfo
    np.testing.assert_allclose(got["res"], meta.res, rtol=1e-9, atol=1e-13 * meta.res[0])   # history, relative to |r0|
    xs = np.concatenate([np.load(f"{out}.x{r}.npy") for r in range(world)])
    np.testing.assert_allclose(xs, x.data, rtol=1e-9, atol=1e-13 * np.abs(x.data).max())
