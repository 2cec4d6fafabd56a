"""The N > 1 entry on a real GPU (BASELINE configs[4]).
* RCCL plumbing with ONE rank: bench.py under torch.distributed.run goes through init_process_group("nccl") and the library's own
  communicator (lk_comm_init_rank: ncclAllReduce issued by liblightkrylov_hip on its stream), or -- LK_NATIVE_RCCL=0 -- through the
  ctypes callback; with one rank the sum is the identity, so the factorisation must be bit-identical to the run without a group.
* Row sharding with 2 / 3 / 4 / 8 REAL processes on this one GPU (gloo standing in for RCCL, which refuses two ranks on one device;
  the engine's hook stages the scalars through host memory): every operator (diagonal, dense, CSR, stencil), the plain `--gpus P`
  self-launch, and configs[4] itself at full size (n = 1e8, m = 128, 8 ranks) against the oracle fixture.
* What must never happen on the first real 8-GPU lease: a silent hang.  A rank that cannot enter the native communicator, a communicator
  that fails inside the collective, a rank that stops moving: each ends the whole job with a non-zero status, the phase every rank was
  in and (watchdog) the stacks -- one attempt per launch, no retry branch.
(The N > 1 host logic is also covered on CPU by tests/test_distributed_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(cmd, timeout=420, **extra_env):
    """One attempt, no retry: bench.py's own watchdog (LK_BENCH_WATCHDOG) turns a stalled rank into a non-zero exit with every
    rank's phase markers and stacks on stderr, which a failure here prints."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    env.setdefault("LK_BENCH_WATCHDOG", "150")
    env.setdefault("LK_TRACE_COLLECTIVES", "1")             # a stalled job shows every rank's last collective (sequence number, count)
    if extra_env.get("LK_DIST_BACKEND") == "gloo":
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")          # the box's hostname does not resolve: keep gloo's full mesh on loopback
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def _run(cmd, **extra_env):
    out = _launch(cmd, **extra_env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-8000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_single_rank_rccl_path_is_bit_identical():
    args = ["--gpus", "1", "--rows", "2000001", "--kdim", "24", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    plain = _run([sys.executable, "bench.py"] + args)
    dist = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                 "127.0.0.1", "--master-port", str(_free_port()), "bench.py"] + args)
    assert plain["config"]["all_reduce"].startswith("none") and dist["config"]["all_reduce"].startswith("RCCL native")
    assert dist["config"]["H_fro"] == plain["config"]["H_fro"]
    assert dist["config"]["H_last_subdiag"] == plain["config"]["H_last_subdiag"]
    assert dist["config"]["info"] == plain["config"]["info"] == 0
    assert dist["n_gpus"] == 1 and dist["roofline"]["launches"] == plain["roofline"]["launches"]
    # the older route (torch.distributed.all_reduce through the ctypes callback) stays selectable and identical
    cb = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), "bench.py"] + args, LK_NATIVE_RCCL="0")
    assert cb["config"]["all_reduce"] == "RCCL via torch.distributed callback"
    assert cb["config"]["H_fro"] == plain["config"]["H_fro"] and cb["config"]["H_last_subdiag"] == plain["config"]["H_last_subdiag"]


def test_two_processes_sharing_one_gpu_reproduce_the_single_process_factorisation():
    """True multi-process row sharding (torch.distributed.run, 2 ranks, both on device 0) with the gloo backend
    standing in for RCCL: every sweep's partial h / norm is all-reduced between the processes.  The
    factorisation must match the single-process one to rounding (partial sums are combined in another order)."""
    args = ["--rows", "3000001", "--kdim", "20", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    plain = _run([sys.executable, "bench.py", "--gpus", "1"] + args)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + args,
               LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
    # ... and through the PLAIN entry: `python bench.py --gpus 2` starts the launcher itself as a child process
    plain2 = _run([sys.executable, "bench.py", "--gpus", "2"] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
    assert plain2["n_gpus"] == 2 and plain2["config"]["H_fro"] == two["config"]["H_fro"]
    assert plain2["config"]["nccl_algo"] == "Ring"                      # the reduction algorithm is pinned
    assert two["n_gpus"] == 2 and two["config"]["all_reduce"].startswith("gloo")
    assert two["config"]["n_local"] == 1500000 and two["config"]["info"] == 0
    assert abs(two["config"]["H_fro"] - plain["config"]["H_fro"]) <= 1e-13 * plain["config"]["H_fro"]
    assert abs(two["config"]["H_last_subdiag"] - plain["config"]["H_last_subdiag"]) <= 1e-12 * plain["config"]["H_last_subdiag"]


@pytest.mark.parametrize("operator,rows", [("dense", "2051"), ("csr", "3721"), ("lap5", "3721")])
def test_two_processes_on_the_row_sharded_dense_and_csr_operators(operator, rows):
    """bench.py --operator dense / csr / lap5 with two PROCESSES (gloo standing in for RCCL, both ranks on one GPU): a row block
    of A per rank and x all-gathered per matvec (dense, CSR); whole grid lines per rank and one line exchanged with the
    neighbour (stencil).  Same factorisation as the single process to rounding."""
    args = ["--operator", operator, "--rows", rows, "--kdim", "12", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        h1, h2 = os.path.join(tmp, "h1.npy"), os.path.join(tmp, "h2.npy")
        one = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-h", h1] + args)
        two = _run([sys.executable, "bench.py", "--gpus", "2", "--dump-h", h2] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
        H1, H2 = np.load(h1), np.load(h2)
    assert two["n_gpus"] == 2 and two["config"]["operator"] == operator and two["config"]["info"] == one["config"]["info"] == 0
    # every column of H normwise at 1e-12 (north_star), the whole matrix -- not two scalars of it
    assert H1.shape == H2.shape
    for j in range(H1.shape[1]):
        assert np.abs(H2[:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max(), (operator, j)
    assert abs(two["config"]["H_fro"] - one["config"]["H_fro"]) <= 1e-12 * one["config"]["H_fro"]
    assert one["roofline"]["matvec"]["launches"] == 12
    # the line times the operator's exchange too (round 5): one all-gather per application for the row-sharded dense / CSR operators,
    # one neighbour exchange for the stencil; none of either on a single rank
    kind = "halo" if operator == "lap5" else "allgather"
    other = "allgather" if operator == "lap5" else "halo"
    assert two["comm"][kind]["launches"] == 12 and two["comm"][kind]["avg_us"] > 0 and two["comm"][other]["launches"] == 0
    assert one["comm"][kind]["launches"] == 0 and two["comm"]["allreduce"]["launches"] >= 3 * 12


@pytest.mark.parametrize("P", [2, 3, 4, 8])
def test_processes_sharing_one_gpu_shard_the_metric_workload(P):
    """BASELINE configs[4]'s partitioning (row blocks of n / P contiguous rows, the last one ragged, reductions all-reduced) with
    P = 2, 3, 4, 8 real processes on one GPU (gloo standing in for RCCL) at a reduced n, through the plain `--gpus P` entry: same H as
    the single process to rounding.  (The full-size 8-process run of the same command is recorded under profiles/.)"""
    args = ["--rows", str(1000000 * P + 2), "--kdim", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + args)
    import time
    for rep in range(int(os.environ.get("LK_TEST_REPEAT", "1"))):          # (the stress run of a round: LK_TEST_REPEAT=20)
        t0 = time.time()
        many = _run([sys.executable, "bench.py", "--gpus", str(P)] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
        if os.environ.get("LK_STRESS_REPORT"):
            with open(os.environ["LK_STRESS_REPORT"], "a") as f:
                f.write(json.dumps({"P": P, "rep": rep, "seconds": round(time.time() - t0, 1), "rc": 0, "H_fro": many["config"]["H_fro"]}) + "\n")
        assert many["n_gpus"] == many["gpus_requested"] == P and many["config"]["n_local"] == 1000000 and many["config"]["info"] == 0
        assert abs(many["config"]["H_fro"] - one["config"]["H_fro"]) <= 1e-13 * one["config"]["H_fro"]
        assert abs(many["config"]["H_last_subdiag"] - one["config"]["H_last_subdiag"]) <= 1e-12 * one["config"]["H_last_subdiag"]


def test_a_rank_that_cannot_enter_the_native_communicator_ends_the_whole_job():
    """ncclCommInitRank is a collective: a rank that fails BEFORE entering it cannot be waited for (the others sit in the
    bootstrap).  bench.py's answer: that rank says so and exits non-zero at once, the launcher terminates the rest -- rc != 0
    within seconds on every rank, never a deadlock with the ranks on different reduction routes."""
    import time
    args = ["--gpus", "2", "--rows", "200001", "--kdim", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    t0 = time.time()
    out = _launch([sys.executable, "bench.py"] + args, timeout=300, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0",
                  LK_NATIVE_RCCL="force", LK_BENCH_TEST_HOOKS="1", LK_TEST_FAIL_COMM_RANK="1")
    took = time.time() - t0
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout[-2000:] + out.stderr[-6000:]
    assert "FATAL after phase 'native RCCL communicator" in out.stderr and "simulated failure of lk_comm_init_rank on this rank only" in out.stderr
    assert took < 120, f"{took:.0f} s"            # interpreter start-up + torch import of three processes; the failure itself is immediate


def test_a_communicator_that_fails_inside_the_collective_fails_on_every_rank_alike():
    """ncclCommInitRank itself failing (here naturally: RCCL refuses two ranks on one GPU, on BOTH ranks): the outcome is MIN-reduced over
    the launcher's group after the call, every rank raises the same error and leaves with status 3 -- nobody falls back alone onto
    another reduction route."""
    import time
    args = ["--gpus", "2", "--rows", "200001", "--kdim", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    t0 = time.time()
    out = _launch([sys.executable, "bench.py"] + args, timeout=300, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0", LK_NATIVE_RCCL="force",
                  LK_BENCH_WATCHDOG="60")
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout[-2000:] + out.stderr[-6000:]
    import re
    assert "native RCCL communicator failed on at least one rank" in out.stderr
    assert len(re.findall(r"exitcode\s*: 3", out.stderr)) >= 2, out.stderr[-4000:]                  # both ranks, the same verdict
    assert time.time() - t0 < 120


def test_a_rank_that_stops_moving_is_reported_with_its_stack_and_ends_the_job():
    """The watchdog of bench.py: one rank hangs before its first engine call; after LK_BENCH_WATCHDOG seconds every thread's stack of
    that rank is on stderr with the phase it was in, the process leaves with status 1 and the launcher ends the job."""
    args = ["--gpus", "2", "--rows", "200001", "--kdim", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    out = _launch([sys.executable, "bench.py"] + args, timeout=300, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0",
                  LK_BENCH_TEST_HOOKS="1", LK_TEST_HANG_RANK="0", LK_BENCH_WATCHDOG="20")
    assert out.returncode != 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "Timeout (0:00:20)!" in out.stderr and "phase: creating the engine context" in out.stderr
    # (the dumped stacks name bench.py's lines; both ranks' watchdogs fire within the same second here and their dumps can interleave
    # character by character on the shared stderr, so only the file name is looked for)
    assert "bench.py" in out.stderr.split("Timeout (0:00:20)!", 1)[1]


def test_config5_at_full_size_as_eight_processes_against_the_oracle_fixture(tmp_path):
    """BASELINE configs[4] -- arnoldi, n = 10^8 real(dp), m = 128, row-sharded 8 ways, reductions all-reduced -- at FULL size as eight
    real processes (12.5 million rows each; all on this one GPU, gloo standing in for RCCL) through the plain `--gpus 8` entry: EVERY
    column of H against the committed fixture of the reference's arithmetic (tests/golden/arnoldi_diaglin_n100000000_m128_rdp.npz, the
    same fixture the single-GPU test uses) normwise within 1e-12, and against the twice-working-precision fixture within 1e-13.  (The
    timing of this run means nothing: eight processes time-share one GPU.)"""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "arnoldi_diaglin_n100000000_m128_rdp.npz"))
    hpath = str(tmp_path / "H8.npy")
    out = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--dump-h", hpath],
               LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
    assert out["n_gpus"] == 8 and out["config"]["n_local"] == 12_500_000 and out["config"]["m"] == 128 and out["config"]["info"] == 0
    H = np.load(hpath)
    colerr = lambda A, B: max(np.abs(A[:, j] - B[:, j]).max() / np.abs(B[:, j]).max() for j in range(B.shape[1]))   # noqa: E731
    e_seq, e_comp = colerr(H, z["H_seq"]), colerr(H, z["H_comp"])
    print(f"n = 1e8, m = 128 on 8 ranks: |dH| vs the reference's arithmetic {e_seq:.2e}, vs compensated dots {e_comp:.2e}")
    assert e_seq <= 1e-12 and e_comp <= 1e-13
    # ... and the bench line checks itself against the same fixture, at any number of ranks
    par = out["config"]["parity"]
    assert par["ok"] is True and par["max_normwise_column_error_vs_reference_arithmetic"] == pytest.approx(e_seq)
    # ... and EXPLAINS itself (round 5): collectives timed on the engine's stream, every rank's sweep figures, the stored PMC traffic of
    # this shard size, the scaling model's prediction, where rank 0's step time went
    ar = out["comm"]["allreduce"]
    assert 3 * 128 <= ar["launches"] <= 3 * 128 + 2 and ar["avg_us"] > 0 and 0 < out["comm"]["frac_of_step_time"] < 1     # three per Arnoldi step (+ x0's norm)
    assert ar["bytes_per_launch"] <= 8 * 129 and ar["ms_total_by_rank"]["max"] >= ar["ms_total_by_rank"]["min"] > 0
    assert out["comm"]["halo"]["launches"] == out["comm"]["allgather"]["launches"] == 0
    pr = out["roofline"]["per_rank"]
    assert len(pr["avg_launch_ms"]["by_rank"]) == 8 and pr["launches_by_rank"] == [3 * 128] * 8
    assert pr["avg_launch_ms"]["min"] <= pr["avg_launch_ms"]["mean"] <= pr["avg_launch_ms"]["max"] == out["roofline"]["avg_launch_ms"]
    assert out["roofline"]["frac"] == pytest.approx(pr["GBps"]["min"] / 8000.0)
    assert out["roofline"]["traffic"] is not None and 0.98 <= out["roofline"]["traffic_over_algorithmic"] <= 1.02, out["roofline"]["traffic_source"]
    assert out["predicted_it_s"] > 250 and out["predicted"]["value_over_predicted"] > 0
    att = out["step_time_attribution_rank0"]
    assert abs(sum(att.values()) - 1.0) < 1e-9 and att["comm"] == out["comm"]["frac_of_step_time"]


def test_a_wrong_reduction_ends_the_run_nonzero_with_its_line_printed():
    """The line polices itself: one rank's partial sums scaled by 1 + 1e-6 before every all-reduce (test hook, callback route) -- the
    factorisation still runs to the end and the line is printed, with config.parity.ok = false against the committed fixture, and EVERY
    rank leaves with a non-zero status.  The same command without the fault: ok = true, status 0."""
    args = ["--gpus", "2", "--rows", "10000000", "--kdim", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    good = _run([sys.executable, "bench.py"] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0")
    assert good["config"]["parity"]["ok"] is True and 3 * 64 <= good["comm"]["allreduce"]["launches"] <= 3 * 64 + 2
    # the hook variable alone (no opt-in) changes nothing
    leaked = _run([sys.executable, "bench.py"] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0", LK_TEST_SCALE_PARTIALS_RANK="1")
    assert leaked["config"]["parity"]["ok"] is True and leaked["config"]["H_fro"] == good["config"]["H_fro"]
    out = _launch([sys.executable, "bench.py"] + args, LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0", LK_BENCH_TEST_HOOKS="1",
                  LK_TEST_SCALE_PARTIALS_RANK="1")
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode != 0 and lines, out.stdout[-2000:] + out.stderr[-6000:]
    bad = json.loads(lines[-1])
    assert bad["config"]["parity"]["ok"] is False and bad["config"]["parity"]["max_normwise_column_error_vs_reference_arithmetic"] > 1e-9
    assert "config.parity.ok is false" in out.stderr
    import re
    assert len(re.findall(r"exitcode\s*: 4", out.stderr)) >= 1 or "exit 4" in out.stderr
