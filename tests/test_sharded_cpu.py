"""world_size-2 gloo tests of the N>1 path on CPU: bench.py launching its own workers, its rendezvous plumbing, and the
PRODUCT under row sharding -- the device loop's host driver (FBSolver) over a sharded NumPy stand-in for the device
context, and the generic host loop with sharded closures (one all-reduce of the A^T partial sums per iteration,
all-reduced ||r||^2 for the line search)."""
import json
import os
import socket
import subprocess
import sys
import warnings

import numpy as np
import pytest

from oracle import fasta_np as fo
from oracle import problems as pr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("path", ["driver", "generic"])
@pytest.mark.parametrize("mode", ["adaptive", "accelerated"])
def test_two_rank_row_sharding_matches_single_rank(tmp_path, mode, path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), mode, path]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]

    np.random.seed(5)
    P = pr.sparse_least_squares(M=96, N=160, K=6)
    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True,
                adaptive=(mode != "accelerated"), accelerate=(mode == "accelerated"))
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank takes the same branches and holds the same replicated iterate
    for key in ("residuals", "stepsizes", "objectives", "solution", "iteration_count", "backtracks"):
        assert np.array_equal(r0[key], r1[key]), key
    assert int(r0["iteration_count"]) == want.iteration_count and int(r0["backtracks"]) == want.backtracks
    k = want.iteration_count
    np.testing.assert_allclose(r0["residuals"][:k], want.residuals[:k], rtol=1e-8)
    np.testing.assert_allclose(r0["objectives"][:k + 1], want.objectives[:k + 1], rtol=1e-10)
    np.testing.assert_allclose(r0["iterates"][:k + 1], want.iterates[:k + 1], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("mode", ["adaptive", "accelerated"])
def test_one_rank_timing_out_makes_both_ranks_drop_the_one_pass_kernel_together(tmp_path, mode):
    """the row-sharded one-pass step all-reduces its timeout word with g1, so a hand-off timeout on ONE rank must make
    EVERY rank fall back to K-fwd / K-adj in the same launch -- otherwise their collective sequences diverge and the job hangs.
    Rank 1's 4th one-pass launch is made to time out; both ranks must report it at launch 3, finish, and match the oracle."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), mode, "timeout"]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    np.random.seed(5)
    P = pr.sparse_least_squares(M=96, N=160, K=6)
    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True,
                adaptive=(mode != "accelerated"), accelerate=(mode == "accelerated"))
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for key in ("residuals", "stepsizes", "objectives", "solution", "iteration_count", "backtracks",
                "timeout_raised_at", "fused_steps", "one_pass_launches", "two_launch_fwd"):
        assert np.array_equal(r0[key], r1[key]), key
    assert int(r0["timeout_raised_at"]) == 3 and int(r0["one_pass_launches"]) == 4 and int(r0["fused_steps"]) == 3
    assert int(r0["iteration_count"]) == want.iteration_count and int(r0["backtracks"]) == want.backtracks
    k = want.iteration_count
    np.testing.assert_allclose(r0["residuals"][:k], want.residuals[:k], rtol=1e-8)
    np.testing.assert_allclose(r0["iterates"][:k + 1], want.iterates[:k + 1], rtol=1e-8, atol=1e-12)


def test_bench_starts_its_own_workers_from_a_bare_shell():
    """`python bench.py --gpus 2` with no torch.distributed.run environment: the script must launch its two workers itself
    (as a child process) and relay rank 0's single JSON line on stdout.  --plumbing-only keeps the GPU out of it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout, and it is JSON
    out = json.loads(lines[0])
    assert out["ranks"] == 2 and out["n_gpus"] == 2 and out["rows_per_rank"] == 32768
    assert out["max_elapsed_s"] >= 0.02                 # the slower rank's time (max over ranks)


def _bare_env(**extra):
    drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FASTA_BENCH_RDV", "HSA_ENABLE_IPC_MODE_LEGACY")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(extra)
    return env


def test_bench_socket_launcher_needs_no_torch():
    """`--launcher socket`: bench.py's own spawner + TCP rendezvous (what `--gpus N` falls back to when torch cannot be imported):
    same single JSON line, max over ranks, and the IPC mode the pool needs exported to every worker."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--plumbing-only", "--launcher", "socket"],
                         capture_output=True, text=True, timeout=120, env=_bare_env(), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["ranks"] == 3 and out["rendezvous"] == "SocketGroup" and out["max_elapsed_s"] >= 0.03
    assert out["hsa_enable_ipc_mode_legacy"] == "0"


def test_bench_exports_the_ipc_mode_when_somebody_else_launches_it():
    """The driver starts `torch.distributed.run ... bench.py --gpus N` itself: the workers must still get
    HSA_ENABLE_IPC_MODE_LEGACY=0 (set at the top of main(), before anything loads HIP), not only under bench.py's own launcher."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_bare_env(OMP_NUM_THREADS="1"), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert out["hsa_enable_ipc_mode_legacy"] == "0" and out["rendezvous"] == "Group" and out["ranks"] == 2


@pytest.mark.parametrize("launcher", ["socket", "torch"])
def test_a_rank_that_dies_ends_the_job_instead_of_hanging_it(launcher):
    """Rank 1 exits mid-job (after the first barrier).  The job must END with a non-zero status within the rendezvous timeout --
    the launcher takes the surviving rank down (or its next barrier fails) -- and print no result line."""
    import time
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--launcher", launcher,
                          "--die-at-rank", "1", "--rdv-timeout", "20"],
                         capture_output=True, text=True, timeout=240, env=_bare_env(), cwd=ROOT)
    assert res.returncode != 0
    assert time.time() - t0 < 120
    assert not [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr


def test_row_partition_of_bench_covers_matrix_once():
    m_total = 65536
    for world in (1, 2, 4, 8):
        rows = [(r * (m_total // world), m_total // world) for r in range(world)]
        assert sum(c for _, c in rows) == m_total
        assert [s for s, _ in rows] == list(range(0, m_total, m_total // world))


def test_socket_rendezvous_ignores_strangers_and_never_unpickles(tmp_path):
    """rank 0 of bench.py's TCP rendezvous used to unpickle whatever the first connections sent.  The frames are now fixed
    binary (kind, length, raw bytes / one float64) behind a per-job token: a stranger that connects first and sends a pickle that would
    create a file is dropped -- nothing it sent is evaluated -- and the job completes with its own ranks."""
    import pickle
    import socket
    import threading
    import time
    sys.path.insert(0, ROOT)
    import bench
    port = _free_port()
    marker = tmp_path / "owned"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))
    payload = pickle.dumps(Evil())

    def stranger():
        deadline = time.time() + 20
        while time.time() < deadline:
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=5)
                break
            except OSError:
                time.sleep(0.02)
        else:
            return
        try:
            c.sendall(len(payload).to_bytes(8, "little") + payload)      # the round-3 wire format
            time.sleep(0.5)
        finally:
            c.close()
    results = {}

    def rank(r):
        os.environ["FASTA_BENCH_TOKEN"] = "feedc0de" * 4
        if r:
            time.sleep(1.0)                                               # the stranger gets there first
        g = bench.SocketGroup(r, 2, f"127.0.0.1:{port}", 30.0)
        g.barrier()
        results[r] = (g.broadcast_bytes(b"uid" * 40 if r == 0 else None), g.max(float(r + 1)))
        g.close()
    threads = [threading.Thread(target=f, args=a) for f, a in ((rank, (0,)), (stranger, ()), (rank, (1,)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert results == {0: (b"uid" * 40, 2.0), 1: (b"uid" * 40, 2.0)}
    assert not marker.exists()
    with pytest.raises(ValueError):                                       # and a malformed frame is an error, not an object
        a, b = socket.socketpair()
        a.sendall(b"P" + (5).to_bytes(8, "little") + b"hello")
        bench.SocketGroup._recv(b)


def test_a_rank_stuck_for_good_is_ended_by_the_job_timeout():
    """A rank that never comes back (as when stuck inside a GPU collective, which no rendezvous timeout covers) must not keep
    `bench.py --gpus N` alive forever: the spawner's wall-clock limit ends every rank and the status is non-zero."""
    import time
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--launcher", "socket",
                          "--hang-at-rank", "1", "--rdv-timeout", "300", "--job-timeout", "6"],
                         capture_output=True, text=True, timeout=120, env=_bare_env(), cwd=ROOT)
    assert res.returncode != 0 and "--job-timeout" in res.stderr
    assert time.time() - t0 < 60
    assert not [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]


def test_preflight_without_a_gpu_fails_at_its_first_step_with_one_line():
    """fasta_python_amd/preflight.py on a GPU-less box: ONE line on stderr naming the step, exit status 3 -- never a traceback, and
    never a fallback to something that "passes"."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "fasta_python_amd.preflight", "2", "--inproc"], capture_output=True, text=True, timeout=120, cwd=root)
    lines = [ln for ln in res.stderr.splitlines() if ln.strip()]
    assert res.returncode == 3, res.stderr[-2000:]
    assert len(lines) == 1 and lines[0].startswith("fasta preflight FAILED at step 'device count'"), res.stderr[-2000:]
